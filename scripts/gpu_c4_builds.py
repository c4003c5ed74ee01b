"""C4 (1.06 M triangles, 1920x1080, 3 bounces): build time, tree size and trace cost of the host builder (0) and the three device
builders (1 binned SAH, 2 Karras LBVH, 3 PLOC).  python scripts/gpu_c4_builds.py [builders...]"""
import sys, time, os; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
which = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
for gpu in which:
    c = capi.Context(W, H)
    c.set_option(capi.OPT_GPU_BUILD, gpu)
    builds = []
    for rep in range(3):
        c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
        builds.append(c.stats().last_build_ms)
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    c.set_option(capi.OPT_COUNTERS, 1); c.reset_stats(); c.render(0, 4, 3); s0 = c.stats(); c.set_option(capi.OPT_COUNTERS, 0)
    ts = []
    for i in range(4):
        c.reset_stats(); c.render(0, 16, 3); st = c.stats(); ts.append(st.last_render_ms)
    print("gpu_build=%d build %s ms nodes4 %d depth4 %d render %.2f ms boxes/ray %.1f tris/ray %.2f" % (gpu, " ".join("%.1f" % x for x in builds), st.bvh_nodes4, st.bvh_depth4, min(ts), 2.0*s0.node_fetches/s0.rays, s0.tri_tests/s0.rays), flush=True)
    c.close()
