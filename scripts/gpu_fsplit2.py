"""Full-frame kernel time (median of reps) per frame-split setting for configs C2/C3/C5/C4."""
import sys
import numpy as np
sys.path.insert(0, ".")
from toyraygun_amd import capi, host

def run(name, scene, W, H, spp, bnc, reps):
    b = scene.buffers()
    c = capi.Context(W, H)
    c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
    c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
    out = []
    for fs in (1, 2, 4, 1, 2, 4):
        c.set_option(capi.OPT_FRAME_SPLIT, fs)
        ts = []
        for _ in range(reps):
            c.render(0, spp, bnc); ts.append(c.stats().last_render_ms)
        out.append("fs%d med %.3f min %.3f" % (fs, float(np.median(ts)), min(ts)))
    print(name, " | ".join(out), flush=True)
    c.close()

box = host.Scene.cornell_box()
run("C2", box, 1920, 1080, 16, 3, 15)
run("C1", box, 256, 256, 1, 3, 15)
run("C2-1spp", box, 1920, 1080, 1, 3, 15)
run("C3", box, 1920, 1080, 64, 8, 5)
run("C5", box, 3840, 2160, 16, 3, 8)
run("C4", host.Scene.cornell_lattice(44), 1920, 1080, 1, 3, 6)
run("C4-16spp", host.Scene.cornell_lattice(44), 1920, 1080, 16, 3, 3)
