"""Where do C2's wave-level instructions go?  The COUNT instantiation's wave-level loop trips (node steps, leaf steps) next to the ray counts:
python scripts/gpu_c2_breakdown.py [c2|c3]   -> per wave-sample: node iterations, leaf iterations, traces."""
import sys; sys.path.insert(0, ".")
import bench
from toyraygun_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[name]
W, H, SPP, B = cfg["w"], cfg["h"], cfg["spp"], cfg["bounces"]
buffers, uniforms = bench.scene_buffers(cfg)
c = capi.Context(W, H)
c.load_scene(buffers["positions"], buffers["normals"], buffers["colors"], buffers["indices"], buffers["material_ids"])
c.set_uniforms(uniforms); c.set_pixel_offsets_seed()
c.set_option(capi.OPT_COUNTERS, 1)
spp = min(SPP, 16)
c.reset_stats(); c.render(0, spp, B)
st = c.stats()
waves = ((W + 15) // 16) * ((H + 15) // 16) * 4
ws = waves * spp
print("rays %d  per pixel-sample %.3f (primary %d bounce %d shadow %d) shaded %d" % (st.rays, st.rays / (W * H * spp), st.primary_rays, st.bounce_rays, st.shadow_rays, st.shaded_hits))
print("lane node steps %d (%.2f / ray), lane tri tests %d (%.3f / ray)" % (st.node_fetches, st.node_fetches / st.rays, st.tri_tests, st.tri_tests / st.rays))
print("wave node iterations %d = %.1f per wave-sample; wave leaf iterations %d = %.1f per wave-sample" % (st.wave_node_iters, st.wave_node_iters / ws, st.wave_tri_iters, st.wave_tri_iters / ws))
print("lanes per node iteration %.1f, per leaf iteration %.1f" % (st.node_fetches / st.wave_node_iters, st.tri_tests / max(1, st.wave_tri_iters)))
c.close()
