"""C4: Cornell box + 44^3 cubes (1,022,244 triangles), 1920x1080, 16 spp, 3 bounces; scene in HBM."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from toyraygun_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
W, H = 1920, 1080
t0 = time.time(); b = host.Scene.cornell_lattice(n).buffers(); print("scene", b["material_ids"].shape[0], "tris in %.1fs" % (time.time() - t0))
c = capi.Context(W, H)
t0 = time.time(); c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"]); print("load+bvh %.1fs" % (time.time() - t0))
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
st = c.stats(); print({k: v for k, v in st.as_dict().items() if k.startswith(("bvh", "scene", "lds"))})
for kern in (0, 1):
    c.set_option(capi.OPT_KERNEL, kern)
    ts = []
    for i in range(4):
        c.reset_stats(); c.render(0, 16, 3); st = c.stats(); ts.append(st.last_render_ms)
    print("kernel %d: %.2f ms  %d rays -> %.2f Grays/s" % (kern, min(ts), st.rays, st.rays / min(ts) / 1e6))
c.set_option(capi.OPT_COUNTERS, 1)
for kern in (1, 0):
    c.set_option(capi.OPT_KERNEL, kern); c.reset_stats(); c.render(0, 16, 3); st = c.stats()
    print("kernel %d counters: node util %.2f tri util %.2f wave node iters/trace %.0f, lds %d" % (kern, st.node_fetches / (64.0 * st.wave_node_iters), st.tri_tests / (64.0 * st.wave_tri_iters), st.wave_node_iters / (st.rays / 64.0), st.lds_bytes))
rays = st.rays; nn = 2.0 * st.node_fetches / rays; nt = st.tri_tests / rays; ps = st.shaded_hits / rays; rb = rays / (W * H * 16)
bpr = 32 * nn + 48 * nt + 76 * ps + 20 / rb
print("nodes/ray(bvh2-eq) %.1f tris/ray %.2f shaded %.2f rays/px-sample %.2f -> %.0f B/ray; node util %.2f tri util %.2f" % (nn, nt, ps, rb, bpr, st.node_fetches / (64.0 * st.wave_node_iters), st.tri_tests / (64.0 * st.wave_tri_iters)))
print("algorithmic GB/s at best time: %.0f" % (rays * bpr / (min(ts) * 1e-3) / 1e9))
from PIL import Image
c.set_option(capi.OPT_COUNTERS, 0); c.render(0, 16, 3)
Image.fromarray(c.postprocess()).save("gpurun_out/c4.png")
