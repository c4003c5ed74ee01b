import sys; sys.path.insert(0, ".")  # run from the repo root: python tests/dev/<script>.py
import numpy as np
from oracle import pyoracle as O
from toyraygun_amd import capi
from tests.util import make_ctx
s = O.OracleScene.cornell_box()
for (w, h, spp, b) in ((96, 64, 4, 3), (64, 64, 5, 8), (33, 17, 3, 15), (256, 256, 1, 1)):
    off = O.pixel_offsets(w, h)
    c = make_ctx(O, s, w, h, offsets=off)
    O.set_trig_mode(O.TRIG_PORTABLE)
    ref, st = O.render(s, w, h, spp, b, offsets=off)
    O.set_trig_mode(O.TRIG_LIBM)
    c.set_option(capi.OPT_STRICT, 1)
    for kern in (capi.KERNEL_DIRECT, capi.KERNEL_POOL):
        c.set_option(capi.OPT_KERNEL, kern); c.reset_stats()
        c.render(0, spp, b)
        img, gs = c.read_accum(), c.stats()
        print(f"{w}x{h} spp{spp} b{b} kernel={kern}: bitexact={bool((img.view(np.uint32)==ref.view(np.uint32)).all())} rays gpu={gs.rays} cpu={st.rays} "
              f"(p{gs.primary_rays}/{st.primary_rays} b{gs.bounce_rays}/{st.bounce_rays} s{gs.shadow_rays}/{st.shadow_rays}) maxdiff={np.abs(img-ref).max():.2e}")
    c.close()
from toyraygun_amd import host
W, H = 1920, 1080
c = capi.Context(W, H)
b_ = s.buffers(); c.load_scene(b_["positions"], b_["normals"], b_["colors"], b_["indices"], b_["material_ids"])
c.set_uniforms(O.uniforms_bytes(O.make_uniforms(W, H))); c.set_pixel_offsets_seed()
for kern in (0, 1, 0, 1):
    c.set_option(capi.OPT_KERNEL, kern)
    for bn in (3, 8):
        ts = []
        for i in range(4):
            c.reset_stats(); c.render(0, 16, bn); st = c.stats(); ts.append(st.last_render_ms)
        print(f"kernel={kern} C2 bounces={bn}: {min(ts):.3f} ms {st.rays/min(ts)/1e6:.1f} Grays/s lds={st.lds_bytes}")
c.set_option(capi.OPT_COUNTERS, 1)
for kern in (0, 1):
    c.set_option(capi.OPT_KERNEL, kern)
    for bn in (3, 8):
        c.reset_stats(); c.render(0, 16, bn); st = c.stats()
        print(f"kernel={kern} b={bn}: node util {st.node_fetches/(64.0*st.wave_node_iters):.2f} tri util {st.tri_tests/(64.0*st.wave_tri_iters):.2f} nodes/ray {st.node_fetches/st.rays:.2f} tris/ray {st.tri_tests/st.rays:.2f}")
