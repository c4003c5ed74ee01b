import sys; sys.path.insert(0, ".")  # run from the repo root: python tests/dev/<script>.py
import numpy as np
from oracle import pyoracle as O
from toyraygun_amd import capi
from tests.util import make_ctx
from tests.test_gpu_parity import _rays, _adversarial_rays
s = O.OracleScene.cornell_box()
c = make_ctx(O, s, 64, 64)
rays = np.concatenate([_rays(O, 60000, 21), _adversarial_rays(O, s)])
ref = O.intersect_nearest(s, rays)
for strict in (1, 0):
    c.set_option(capi.OPT_STRICT, strict)
    got = c.trace(rays)
    bad = np.where((got.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4)).any(1))[0]
    print("strict", strict, "mismatch", len(bad), "of", len(rays))
    for i in bad[:12]:
        print(i, rays[i], "gpu", got[i], "cpu", ref[i])
