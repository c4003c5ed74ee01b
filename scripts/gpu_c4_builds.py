"""C4 with the host SAH build vs the GPU LBVH build: build time, node counts, render time."""
import sys, time; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
for gpu in (0, 1):
    c = capi.Context(W, H)
    c.set_option(capi.OPT_GPU_BUILD, gpu)
    t0 = time.time(); c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"]); wall = time.time() - t0
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    ts = []
    for i in range(4):
        c.reset_stats(); c.render(0, 16, 3); st = c.stats(); ts.append(st.last_render_ms)
    print("gpu_build=%d build %.1f ms (load wall %.2f s) nodes4 %d depth4 %d scene %.1f MB render %.2f ms" % (
        gpu, st.last_build_ms, wall, st.bvh_nodes4, st.bvh_depth4, st.scene_bytes / 1e6, min(ts)), flush=True)
    c.close()
