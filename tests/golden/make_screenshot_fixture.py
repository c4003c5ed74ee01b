"""Makes the screenshot fixtures under tests/golden/ from the reference's own screenshots (README.md:30-36): the only
OUTPUTS of the reference that exist.  Runs only in the authoring container (the reference tree does not travel).
The fixtures are data (pixels), not source:

  reference_screenshot_metal_256x192.png    content area of img/screenshot_metal.png box-filtered to 256x192 (colour /
                                            brightness pin, tests/test_screenshot_pin.py::test_oracle_looks_like...)
  reference_screenshot_metal_1021x766.png   the same content area at the window's logical resolution: the 2042x1532
                                            retina pixels box-filtered 2 -> 1 (the app window is 1024x768, main.cpp:22;
                                            the screenshot lacks 6 x 4 retina pixels of it)
  reference_screenshot_d3d12_1024x768.png   client area of img/screenshot_d3d12.png, native resolution, exact crop
                                            (1-pixel window border, 31-row title bar)
"""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/img"
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference tree not present")
    im = Image.open(os.path.join(REF, "screenshot_metal.png")).convert("RGB")
    assert im.size == (2042, 1588)
    content = im.crop((0, 55, 2042, 1588))  # rows 0..53 = title bar, row 54 = black separator
    small = content.resize((256, 192), Image.BOX)
    small.save(os.path.join(HERE, "reference_screenshot_metal_256x192.png"), optimize=True)
    half = im.crop((0, 55, 2042, 1587)).resize((1021, 766), Image.BOX)
    half.save(os.path.join(HERE, "reference_screenshot_metal_1021x766.png"), optimize=True)
    d3d = Image.open(os.path.join(REF, "screenshot_d3d12.png")).convert("RGB")
    assert d3d.size == (1026, 800)
    a = np.asarray(d3d)
    assert (a[31:799, 1] == 0).all() and (a[30, 1:1025] == 255).all() and (a[799, 1] != 0).any()   # the client area is rows 31..798, cols 1..1024
    d3d.crop((1, 31, 1025, 799)).save(os.path.join(HERE, "reference_screenshot_d3d12_1024x768.png"), optimize=True)
    for f in sorted(os.listdir(HERE)):
        if f.startswith("reference_screenshot"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
