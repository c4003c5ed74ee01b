"""Debug aid: the shipped build's hit records on the Cornell box traversed from HBM against the oracle; prints the rays whose weights differ."""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from toyraygun_amd import capi
from oracle import pyoracle as O
from util import make_ctx
import test_gpu_parity as T
cornell = O.OracleScene.cornell_box()
c = make_ctx(O, cornell, 64, 64)
c.set_option(capi.OPT_FORCE_GLOBAL, 1)
rays = np.concatenate([T._rays(O, 60000, 21), T._adversarial_rays(O, cornell)])
ref = O.intersect_nearest(cornell, rays)
c.set_option(capi.OPT_STRICT, 0)
fast = c.trace(rays)
same = (fast["primitiveIndex"] == ref["primitiveIndex"]) & (ref["primitiveIndex"] >= 0)
bad = same & (np.abs(fast["coordinates"] - ref["coordinates"]).max(1) > 2e-5)
print("bad", int(bad.sum()), "of", int(same.sum()))
for i in np.nonzero(bad)[0][:12]:
    print(i, "prim", ref["primitiveIndex"][i], "t", fast["distance"][i], ref["distance"][i], "uv fast", fast["coordinates"][i], "ref", ref["coordinates"][i], "o", rays["origin"][i], "d", rays["direction"][i], "mask", rays["mask"][i], "max", rays["maxDistance"][i])
