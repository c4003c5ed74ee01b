"""Traversal counters of a config: python scripts/gpu_counters.py [bounces] [spp]"""
import sys
sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
b = host.Scene.cornell_box().buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
c.set_option(capi.OPT_COUNTERS, 1)
c.reset_stats(); c.render(0, spp, bn); st = c.stats()
waves = W * H / 64.0
print("rays %d primary %d bounce %d shadow %d shaded %d" % (st.rays, st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits))
print("node_fetches %d tri_tests %d wave_node_iters %d wave_tri_iters %d" % (st.node_fetches, st.tri_tests, st.wave_node_iters, st.wave_tri_iters))
print("per wave-sample: node iters %.1f tri iters %.1f ; lane util nodes %.3f tris %.3f" % (
    st.wave_node_iters / (waves * spp), st.wave_tri_iters / (waves * spp),
    st.node_fetches / (64.0 * st.wave_node_iters), st.tri_tests / (64.0 * st.wave_tri_iters)))
print("ms %.3f" % st.last_render_ms)
