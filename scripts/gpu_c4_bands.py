"""An HBM-resident lattice scene (default 44 = C4, 1.02 M triangles; python scripts/gpu_c4_bands.py 6 = 2,628 triangles, 12 = 20,772) at
1920 wide, 16 spp, 3 bounces on row bands of a multi-GPU job / small windows: ms alone per schedule."""
import sys; sys.path.insert(0, ".")
from toyraygun_amd import capi, host
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 44
b = host.Scene.cornell_lattice(N).buffers()
print("lattice %d: %d triangles" % (N, len(b["material_ids"])))
c = capi.Context(W, H)
c.load_scene(b["positions"], b["normals"], b["colors"], b["indices"], b["material_ids"])
c.set_uniforms(host.uniforms(W, H)[0]); c.set_pixel_offsets_seed()
for rows, row0 in ((135, 472), (270, 405), (540, 270), (1080, 0)):
    for name, fs, rg in (("auto", 0, -1), ("fsplit1 lock", 1, 0), ("fsplit1 regen", 1, 1), ("fsplit4 lock", 4, 0), ("fsplit2 lock", 2, 0), ("fsplit2 regen", 2, 1), ("fsplit4 regen", 4, 1)):
        c.set_option(capi.OPT_FRAME_SPLIT, fs); c.set_option(capi.OPT_REGEN, rg)
        c.render(0, 16, 3, row0, rows)
        ts = []
        for _ in range(3):
            c.reset_stats(); c.render(0, 16, 3, row0, rows); st = c.stats(); ts.append(st.last_render_ms)
        print("band %4d rows: %-14s %7.2f ms (frame split %d, regen %d)" % (rows, name, min(ts), st.last_frame_split, st.last_regen), flush=True)
