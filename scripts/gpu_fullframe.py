"""WHOLE-frame parity of the full-size configurations against the oracle (minutes of oracle time on the box's host threads: run by hand, the
test suite compares sampled rows of C3 / C4 / C5 and the whole frame of C2 only):  python scripts/gpu_fullframe.py [c2 c3 c4 c5]
Per configuration: the strict build bit for bit (image and ray counts), the shipped build within the stated tolerance."""
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from toyraygun_amd import capi
from oracle import pyoracle as O
from util import image_metrics, TOL_RMSE, TOL_FRAC, TOL_FRAC_C4

CONFIGS = {"c2": (1920, 1080, 16, 3, "box"), "c3": (1920, 1080, 256, 8, "box"), "c4": (1920, 1080, 16, 3, "lattice"), "c5": (3840, 2160, 64, 3, "box")}
bad = 0
for name in (sys.argv[1:] or ["c2", "c3", "c4", "c5"]):
    w, h, spp, bnc, kind = CONFIGS[name]
    scene = O.OracleScene.cornell_lattice(44) if kind == "lattice" else O.OracleScene.cornell_box()
    b = scene.buffers()
    off = O.pixel_offsets(w, h)
    c = capi.Context(w, h)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(O.uniforms_bytes(O.make_uniforms(w, h))); c.set_pixel_offsets(off)
    t0 = time.time()
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref, rst = O.render(scene, w, h, spp, bnc, offsets=off)
    O.set_trig_mode(O.TRIG_LIBM)
    t_or = time.time() - t0
    c.set_option(capi.OPT_STRICT, 1); c.reset_stats(); c.render(0, spp, bnc)
    img, st = c.read_accum(), c.stats()
    same = np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    rays = (st.primary_rays, st.bounce_rays, st.shadow_rays) == (rst.primary_rays, rst.bounce_rays, rst.shadow_rays)
    print("%s strict: %dx%d %d spp %d bounces, %d rays: image %s, ray counts %s (oracle %.0f s, GPU %.1f ms)" % (
        name, w, h, spp, bnc, st.rays, "bit-identical" if same else "DIFFERS in %d pixels" % int((img.view(np.uint32) != ref.view(np.uint32)).any(-1).sum()),
        "equal" if rays else "DIFFER", t_or, st.last_render_ms), flush=True)
    ref, _ = O.render(scene, w, h, spp, bnc, offsets=off)
    c.set_option(capi.OPT_STRICT, 0); c.render(0, spp, bnc)
    rmse, frac_ok, worst = image_metrics(c.read_accum(), ref)
    tol = TOL_FRAC_C4 if name == "c4" else TOL_FRAC
    ok = rmse <= TOL_RMSE and frac_ok >= tol
    print("%s shipped: rmse %.3g (<= %.0e), pixels within tolerance %.5f (>= %.4f), worst %.3g: %s" % (name, rmse, TOL_RMSE, frac_ok, tol, worst, "ok" if ok else "OUT OF TOLERANCE"), flush=True)
    bad += (not same) + (not rays) + (not ok)
    c.close()
sys.exit(1 if bad else 0)
