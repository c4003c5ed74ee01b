"""Makes tests/golden/reference_screenshot_metal_256x192.png from the reference's own Metal screenshot
(/root/reference/img/screenshot_metal.png, README.md:30-36): the only OUTPUT of the reference that exists.
Runs only in the authoring container (the reference tree does not travel).  The fixture is data: the
window's content area (title bar and the 1-pixel border row cropped), box-filtered to 256x192.
"""
import os
import sys

import numpy as np
from PIL import Image

SRC = "/root/reference/img/screenshot_metal.png"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_screenshot_metal_256x192.png")

if __name__ == "__main__":
    if not os.path.exists(SRC):
        sys.exit("reference tree not present")
    im = Image.open(SRC).convert("RGB")
    assert im.size == (2042, 1588)
    content = im.crop((0, 55, 2042, 1588))  # rows 0..53 = title bar, row 54 = black separator
    small = content.resize((256, 192), Image.BOX)
    small.save(DST, optimize=True)
    print("wrote", DST, os.path.getsize(DST), "bytes; mean", np.asarray(small).mean(axis=(0, 1)))
