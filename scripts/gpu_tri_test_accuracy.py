"""How often does the shipped build's triangle test pick another primitive than the oracle, and how close to an edge are those rays?
python scripts/gpu_tri_test_accuracy.py [exp_build variant]   (the test's 60,000 random + 264 adversarial rays on the Cornell box)"""
import os, sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from toyraygun_amd import capi
if len(sys.argv) > 1:
    capi.HIP_SO = os.path.join("exp_build", sys.argv[1], "libtoyraygun_hip.so")
from oracle import pyoracle as O
import test_gpu_parity as T
scene = O.OracleScene.cornell_box()
c = T.make_ctx(O, scene, 256, 256)
rays = np.concatenate([T._rays(O, 60000, 21), T._adversarial_rays(O, scene)])
ref = O.intersect_nearest(scene, rays)
c.set_option(capi.OPT_STRICT, 0)
for fg in (0, 1):
    c.set_option(capi.OPT_FORCE_GLOBAL, fg)
    fast = c.trace(rays)
    diff = fast["primitiveIndex"] != ref["primitiveIndex"]
    _, _, margin = O.nearest_f64(scene, rays[diff])
    same = ~diff & (ref["primitiveIndex"] >= 0)
    dt = np.abs(fast["distance"][same] - ref["distance"][same]) / np.maximum(1.0, ref["distance"][same])
    duv = np.abs(fast["coordinates"][same] - ref["coordinates"][same]).max(1)
    print("force_global %d: %d of %d rays differ (%.5f), random part %d, adversarial part %d; margins of those: max %.2e p90 %.2e median %.2e; same-primitive hits: |dt|/max(1,t) max %.2e, |d(u,v)| max %.2e" % (
        fg, diff.sum(), len(rays), diff.mean(), diff[:60000].sum(), diff[60000:].sum(), margin.max(), np.percentile(margin, 90), np.median(margin), dt.max(), duv.max()))
c.close()
