"""Where does path regeneration start to pay?  Lock-step vs regeneration on lattices of growing size (1920x1080, 16 spp, 3 bounces,
one launch alone, ms): python scripts/regen_crossover.py"""
import sys; sys.path.insert(0, ".")
import os
from toyraygun_amd import capi, host
if os.environ.get("TRG_VARIANT"):
    capi.HIP_SO = os.path.join("exp_build", os.environ["TRG_VARIANT"], "libtoyraygun_hip.so")
W, H = 1920, 1080
for n in [int(a) for a in sys.argv[1:]] or (0, 4, 6, 9, 12, 16, 20, 28, 36):
    sc = host.Scene.cornell_lattice(n) if n else host.Scene.cornell_box()
    b = sc.buffers()
    c = capi.Context(W, H)
    c.set_option(capi.OPT_FORCE_GLOBAL, 1)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    out = []
    for regen in (0, 1):
        c.set_option(capi.OPT_REGEN, regen)
        c.render(0, 16, 3)
        ts = []
        for _ in range(3):
            c.reset_stats(); c.render(0, 16, 3); ts.append(c.stats().last_render_ms)
        out.append(min(ts))
    st = c.stats()
    print("lattice %2d: %8d triangles, depth4 %2d: lock step %7.2f ms, regeneration %7.2f ms (%.2fx)" % (n, len(b["material_ids"]), st.bvh_depth4, out[0], out[1], out[0] / out[1]), flush=True)
    c.close()
