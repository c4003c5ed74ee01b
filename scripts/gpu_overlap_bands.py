"""Steady-state ms per step of one rank's row band vs frames in flight (no gather), one process per configuration
(streams map round-robin onto GPU_MAX_HW_QUEUES hardware queues): python scripts/gpu_overlap_bands.py [depth]"""
import sys, time; sys.path.insert(0, ".")
import torch
from toyraygun_amd import capi, host
from toyraygun_amd.dist import DistributedRenderer, band_rows
W, H = 1920, 1080
b = host.Scene.cornell_box().buffers()
import os, subprocess
if len(sys.argv) == 1:
    for d in (1, 2, 4, 0):
        subprocess.run([sys.executable, __file__, str(d)], env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    sys.exit(0)
depth = int(sys.argv[1]); fs = int(os.environ.get("FS", "0"))   # FS: frame-split override (0 = automatic)
for pipelined in (depth != 1,):
    r = DistributedRenderer(W, H, 0, pipelined=pipelined, depth=depth or None)
    r.load_scene(b); r.ctx.set_uniforms(host.uniforms(W, H)[0]); r.ctx.set_pixel_offsets_seed()
    r.ctx.set_option(capi.OPT_TIMING, 0)
    r.ctx.set_option(capi.OPT_FRAME_SPLIT, fs)
    out = []
    for n in (1, 2, 3, 4, 6, 8):
        r.row0, r.rows = band_rows(H, n, n // 2)
        for _ in range(5):
            r.render(0, 16, 3, gather=False)
        r.synchronize()
        t0 = time.perf_counter()
        K = 40
        for _ in range(K):
            r.render(0, 16, 3, gather=False)
        r.synchronize()
        out.append("N=%d %.3f" % (n, (time.perf_counter() - t0) / K * 1e3))
    print("%s streams" % (depth or "auto(full frame)"), " | ".join(out), "overlap", r._overlap, [x.cuda_stream for x in r.render_streams], flush=True)
    r.close()
