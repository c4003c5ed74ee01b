import sys, os, numpy as np
sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
buf = host.Scene.cornell_box().buffers(); u = host.uniforms(W, H)[0]
c = capi.Context(W, H)
c.load_scene(buf["positions"], buf["normals"], buf["colors"], buf["indices"], buf["material_ids"])
c.set_uniforms(u); c.set_pixel_offsets_seed()
for b in (0, 1, 2, 3, 8):
    ts = []
    for i in range(5):
        c.reset_stats(); c.render(0, 16, b); st = c.stats(); ts.append(st.last_render_ms)
    print("bounces %d: %.3f ms, rays %d (p %d b %d s %d) %.1f Grays/s" % (b, min(ts), st.rays, st.primary_rays, st.bounce_rays, st.shadow_rays, st.rays / min(ts) / 1e6))
c.set_option(capi.OPT_COUNTERS, 1)
for b in (1, 3, 8):
    c.reset_stats(); c.render(0, 16, b); st = c.stats()
    print("bounces %d: nodes/ray %.2f tris/ray %.2f | wave node iters %d tri iters %d | node-loop lane util %.2f tri-loop util %.2f | per wave-trace: %.1f node iters %.1f tri iters" % (
        b, st.node_fetches / st.rays, st.tri_tests / st.rays, st.wave_node_iters, st.wave_tri_iters,
        st.node_fetches / (64.0 * st.wave_node_iters), st.tri_tests / (64.0 * st.wave_tri_iters),
        st.wave_node_iters / (st.rays / 64.0), st.wave_tri_iters / (st.rays / 64.0)))
