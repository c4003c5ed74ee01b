import sys; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
c.reset_stats(); c.render(0, 16, 3); s = c.stats()
it = s.wave_node_iters
print("iterations %d: waiting %.1f lanes, out of work %.1f lanes, traversing %.1f" % (it, s.node_fetches / it, s.tri_tests / it, 64 - (s.node_fetches + s.tri_tests) / it))
