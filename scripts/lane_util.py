"""Lane utilisation of the traversal loop on C4 from the in-kernel counters: python scripts/lane_util.py  (TRG_HIP_SO selects a variant)"""
import sys; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
c.set_option(capi.OPT_COUNTERS, 1)
c.reset_stats(); c.render(0, 16, 3); s = c.stats()
print("rays %d node_fetches %d wave_node_iters %d tri_tests %d wave_tri_iters %d -> lanes per wave step: nodes %.1f tris %.1f" % (
    s.rays, s.node_fetches, s.wave_node_iters, s.tri_tests, s.wave_tri_iters, s.node_fetches / 2.0 / max(s.wave_node_iters, 1), s.tri_tests / max(s.wave_tri_iters, 1)))
