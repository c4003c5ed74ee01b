"""Per-band kernel times of C2 for N = 2, 4, 8 row bands (what each rank of a multi-GPU run would execute)."""
import sys; sys.path.insert(0, ".")
import os
from toyraygun_amd import capi, host
if os.environ.get("TRG_EXP_LIB"):
    b_ = host.Scene.cornell_box().buffers(); u_ = bytes(host.uniforms(1920, 1080)[0])
from toyraygun_amd.dist import band_rows
W, H = 1920, 1080
b = host.Scene.cornell_box().buffers()
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
c.render(0, 16, 3)
full = min(c.stats().last_render_ms for _ in range(3) if not c.render(0, 16, 3))
print("full frame %.3f ms" % full)
for n in (2, 4, 8):
    ts = []
    for r in range(n):
        row0, rows = band_rows(H, n, r)
        best = 1e9
        for _ in range(4):
            c.render(0, 16, 3, row0, rows); best = min(best, c.stats().last_render_ms)
        ts.append(best)
    print("N=%d bands: %s  max %.3f mean %.3f -> ideal speedup %.2f, band-limited speedup %.2f" % (n, " ".join("%.3f" % t for t in ts), max(ts), sum(ts) / n, full / (sum(ts) / n), full / max(ts)))
