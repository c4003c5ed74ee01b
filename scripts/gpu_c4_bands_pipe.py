"""C4 row bands with four launches in flight (the multi-GPU pipeline on one rank): ms per step for the regeneration kernel with 1 / 2 / 4
frame lanes and for the automatic choice.  python scripts/gpu_c4_bands_pipe.py"""
import os, sys, time; sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from toyraygun_amd import capi, host
from toyraygun_amd.dist import DistributedRenderer
W, H = 1920, 1080
b = host.Scene.cornell_lattice(44).buffers()
r = DistributedRenderer(W, H, 0, pipelined=True)
r.load_scene(b); r.ctx.set_uniforms(host.uniforms(W, H)[0]); r.ctx.set_pixel_offsets_seed()
r.ctx.set_option(capi.OPT_TIMING, 0)
for rows, row0 in ((135, 472), (270, 405), (540, 270)):
    r.row0, r.rows = row0, rows
    for name, fs in (("auto", 0), ("1 lane", 1), ("2 lanes", 2), ("4 lanes", 4)):
        r.ctx.set_option(capi.OPT_FRAME_SPLIT, fs)
        for _ in range(8):
            r.render(0, 16, 3, gather=False)
        r.synchronize()
        K = 24
        t0 = time.perf_counter()
        for _ in range(K):
            r.render(0, 16, 3, gather=False)
        r.synchronize()
        print("band %4d rows, 4 in flight: %-8s %6.2f ms per step (frame split %d)" % (rows, name, (time.perf_counter() - t0) / K * 1e3, r.ctx.stats().last_frame_split), flush=True)
