"""Per-band kernel times of C2 (and optionally the C4 lattice) for each frame-split setting:
python scripts/gpu_fsplit.py [c4]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from toyraygun_amd import capi, host
from toyraygun_amd.dist import band_rows

W, H = 1920, 1080
scene = host.Scene.cornell_lattice(44) if "c4" in sys.argv[1:] else host.Scene.cornell_box()
spp = int([a[4:] for a in sys.argv[1:] if a.startswith("spp=")][0]) if any(a.startswith("spp=") for a in sys.argv[1:]) else 16
b = scene.buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
for fs in (1, 2, 4, 0):
    c.set_option(capi.OPT_FRAME_SPLIT, fs)
    out = []
    for n in (1, 2, 3, 4, 6, 8):
        ts = []
        for r in range(n):
            row0, rows = band_rows(H, n, r)
            best = 1e9
            for _ in range(4):
                c.render(0, spp, 3, row0, rows); best = min(best, c.stats().last_render_ms)
            ts.append(best)
        out.append("N=%d %.3f" % (n, max(ts)))
    print("fsplit=%d spp=%d  %s" % (fs, spp, " | ".join(out)), flush=True)
