"""Traversal steps per ray of an HBM-resident configuration from the in-kernel counters (the COUNT instantiation), for the shipped library or an
exp_build variant:  python scripts/gpu_steps.py <c4|c4xl> <variant|shipped> [regen 0|1]
node units are BVH2-equivalent fetches (a 4-wide step counts 2, an 8-wide step 4); wave_* = iterations in which ANY lane of a wavefront did that kind of work."""
import os, sys
sys.path.insert(0, ".")
from toyraygun_amd import capi
cfg, name = sys.argv[1], sys.argv[2]
if name != "shipped":
    capi.HIP_SO = os.path.join("exp_build", name, "libtoyraygun_hip.so")
from toyraygun_amd import host
n = {"c4": 44, "c4xl": 96, "c4s": 20}[cfg]
b = host.Scene.cornell_lattice(n).buffers()
W, H = 1920, 1080
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
if len(sys.argv) > 3:
    c.set_option(capi.OPT_REGEN, int(sys.argv[3]))
c.set_option(capi.OPT_COUNTERS, 1)
c.reset_stats(); c.render(0, 16, 3)
st = c.stats()
r = float(st.rays)
per = 4.0 if "w8" in name else 2.0
print("%-5s %-10s rays %.1f M | node steps/ray %.2f (BVH2-eq fetches %.1f) tri tests/ray %.2f | wave node iters %.1f M, wave tri iters %.1f M | lanes per node iteration %.1f, per tri iteration %.1f | %.2f ms (COUNT build)"
      % (cfg, name, r / 1e6, st.node_fetches / per / r, st.node_fetches / r, st.tri_tests / r, st.wave_node_iters / 1e6, st.wave_tri_iters / 1e6,
         st.node_fetches / per / max(st.wave_node_iters, 1), st.tri_tests / max(st.wave_tri_iters, 1), st.last_render_ms), flush=True)
