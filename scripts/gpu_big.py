"""Lattice scenes beyond the Infinity Cache: python scripts/gpu_big.py N [gpu_build]  (N^3 cubes x 12 triangles + the box)"""
import sys, time, json; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
from toyraygun_amd.roofline import algorithmic_bytes_per_ray
n = int(sys.argv[1]); gpu = len(sys.argv) > 2
W, H, SPP, B = 1920, 1080, 16, 3
t0 = time.time(); b = host.Scene.cornell_lattice(n).buffers(); t_scene = time.time() - t0
c = capi.Context(W, H)
c.set_option(capi.OPT_GPU_BUILD, int(gpu))
t0 = time.time(); c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"]); t_load = time.time() - t0
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
ts = []
for i in range(3):
    c.reset_stats(); c.render(0, SPP, B); st = c.stats(); ts.append(st.last_render_ms)
c.set_option(capi.OPT_COUNTERS, 1); c.reset_stats(); c.render(0, SPP, B); cs = c.stats()
bpr, mix = algorithmic_bytes_per_ray(cs, W * H * SPP)
ms = min(ts)
print(json.dumps(dict(lattice=n, tris=int(b["material_ids"].shape[0]), gpu_build=int(gpu), scene_s=round(t_scene, 1), load_s=round(t_load, 2), build_ms=round(st.last_build_ms, 1),
                      scene_MB=round(st.scene_bytes / 1e6, 1), nodes4=int(st.bvh_nodes4), depth4=int(st.bvh_depth4), ms=round(ms, 2), rays=int(st.rays),
                      grays_per_s=round(st.rays / ms / 1e6, 2), bytes_per_ray=round(bpr, 1), algorithmic_TBps=round(st.rays * bpr / (ms * 1e-3) / 1e12, 2),
                      nodes_per_ray=round(mix["nodes_per_ray"], 1), tris_per_ray=round(mix["tris_per_ray"], 2))), flush=True)
