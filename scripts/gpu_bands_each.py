"""Kernel time of every row band for N ranks (auto frame split): python scripts/gpu_bands_each.py"""
import sys; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
from toyraygun_amd.dist import band_rows
W, H = 1920, 1080
b = host.Scene.cornell_box().buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
for n in (2, 4, 8):
    ts = []
    for r in range(n):
        row0, rows = band_rows(H, n, r)
        best = 1e9
        for _ in range(5):
            c.render(0, 16, 3, row0, rows); best = min(best, c.stats().last_render_ms)
        ts.append(best)
    print("N=%d bands: %s  max/mean %.3f" % (n, " ".join("%.3f" % t for t in ts), max(ts) / (sum(ts) / n)), flush=True)
