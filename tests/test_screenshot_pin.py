"""Weak parity pin against the ONLY reference output that exists: the Metal screenshot in the reference's
README (img/screenshot_metal.png, committed as a 256x192 data fixture by tests/golden/make_screenshot_fixture.py).
The oracle renders the reference app's default view (1024x768, main.cpp:22; 3 bounces, MetalRenderer.mm:426),
applies the reference's ACES + sRGB (PostProcessing.metal:44-57) and must look like the screenshot:
same layout, same colours, same brightness.  Loose by construction (unknown frame count, display colour
profile, +-3 pixel crop), so this pins camera, scene, light and tone curve -- not arithmetic."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_looks_like_the_reference_screenshot(O, cornell):
    ref = np.asarray(Image.open(os.path.join(HERE, "golden", "reference_screenshot_metal_256x192.png")).convert("RGB")).astype(np.float64) / 255.0
    acc, _ = O.render(cornell, 1024, 768, 12, 3)
    img = O.postprocess(acc, flip_y=True)[..., :3]
    small = np.asarray(Image.fromarray(img).resize((256, 192), Image.BOX)).astype(np.float64) / 255.0

    # blur both to 64x48 cells: insensitive to the few-pixel crop uncertainty and to residual noise
    def cells(a):
        return a.reshape(48, 4, 64, 4, 3).mean((1, 3))
    a, b = cells(small), cells(ref)
    assert np.abs(a - b).mean() < 0.05                 # mean absolute error in sRGB units
    for ch in range(3):
        assert np.corrcoef(a[..., ch].ravel(), b[..., ch].ravel())[0, 1] > 0.93
    # box opening: black side bars of the same width (box half-width / (aspect * 2.38 * tan 22.5 deg) = 0.38)
    def bar_width(x):
        col = x.mean((0, 2))
        return int(np.argmax(col > 0.02)), int(np.argmax(col[::-1] > 0.02))
    (l0, r0), (l1, r1) = bar_width(small), bar_width(ref)
    assert abs(l0 - l1) <= 2 and abs(r0 - r1) <= 2 and 28 <= l0 <= 33
    # red wall left, green wall right, light on the ceiling (top of the picture)
    assert small[60:130, 36:56, 0].mean() > 2 * small[60:130, 36:56, 1].mean()
    assert small[60:130, 200:220, 1].mean() > 1.5 * small[60:130, 200:220, 0].mean()
    ly, lx = np.unravel_index(np.argmax(small.sum(-1)), small.shape[:2])
    ry, rx = np.unravel_index(np.argmax(ref.sum(-1)), ref.shape[:2])
    assert ly < 50 and ry < 50 and abs(lx - 128) < 24 and abs(rx - 128) < 24
