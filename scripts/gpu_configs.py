"""Runs BASELINE configs C1..C5 on one GPU and prints one JSON line per config (developer script; results go to DESIGN.md)."""
import sys, json, time; sys.path.insert(0, ".")
import numpy as np
from toyraygun_amd import capi, host
from toyraygun_amd.roofline import algorithmic_bytes_per_ray

def run(tag, w, h, spp, bounces, scene, kernel=0, reps=3):
    c = capi.Context(w, h)
    b = scene.buffers()
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(w, h)[0]); c.set_pixel_offsets_seed()
    c.set_option(capi.OPT_KERNEL, kernel)
    ts = []
    for i in range(reps):
        c.reset_stats(); c.render(0, spp, bounces); st = c.stats(); ts.append(st.last_render_ms)
    c.set_option(capi.OPT_COUNTERS, 1); c.reset_stats(); c.render(0, min(spp, 16), bounces); cs = c.stats()
    bpr, mix = algorithmic_bytes_per_ray(cs, w * h * min(spp, 16))
    nn, nt, rb = mix["nodes_per_ray"], mix["tris_per_ray"], mix["rays_per_pixel_sample"]
    ms = min(ts)
    out = dict(config=tag, w=w, h=h, spp=spp, bounces=bounces, tris=int(b["material_ids"].shape[0]), kernel=["direct", "pool"][kernel],
               ms=round(ms, 3), rays=int(st.rays), grays_per_s=round(st.rays / ms / 1e6, 2), bytes_per_ray=round(bpr, 1),
               algorithmic_TBps=round(st.rays * bpr / (ms * 1e-3) / 1e12, 2), frac_of_8TBps=round(st.rays * bpr / (ms * 1e-3) / 8e12, 3),
               nodes_per_ray=round(nn, 2), tris_per_ray=round(nt, 2), rays_per_pixel_sample=round(rb, 2),
               scene_in_lds=int(st.scene_in_lds), lds_bytes=int(st.lds_bytes), bvh_nodes=int(st.bvh_nodes), bvh_depth=int(st.bvh_depth))
    print(json.dumps(out), flush=True)
    c.close()

cornell = host.Scene.cornell_box()
run("C1", 256, 256, 1, 1, cornell)
run("C2", 1920, 1080, 16, 3, cornell)
run("C2-pool", 1920, 1080, 16, 3, cornell, kernel=1)
run("C3", 1920, 1080, 256, 8, cornell, reps=2)
run("C3-pool", 1920, 1080, 256, 8, cornell, kernel=1, reps=2)
run("C5-1gpu", 3840, 2160, 64, 3, cornell, reps=2)
run("C4", 1920, 1080, 16, 3, host.Scene.cornell_lattice(44))
