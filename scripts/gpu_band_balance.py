"""How evenly do the row bands of an N-GPU job cost?  Every band of the frame is rendered on THIS GPU by itself -- what its own GPU
would do in the N-GPU job -- once alone (HIP events around the launch) and once as the steady state of four frames in flight, and the
spread (max / mean) is printed next to the bands' ray counts.  SURVEY 8(e): above 5 % the bands should interleave.

  python scripts/gpu_band_balance.py [c2|c4|c5] [N ...]      -> one JSON line per N and band rule (contiguous rows / interleaved micro-bands)
"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import bench  # noqa: E402
from toyraygun_amd import capi  # noqa: E402
from toyraygun_amd.dist import DistributedRenderer, band_rows  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
ns = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
cfg = bench.CONFIGS[name]
W, H, SPP, B = cfg["w"], cfg["h"], cfg["spp"], cfg["bounces"]
buffers, uniforms = bench.scene_buffers(cfg)
K = int(os.environ.get("K", "24" if name != "c4" else "8"))

r = DistributedRenderer(W, H, 0, pipelined=True)
r.load_scene(buffers)
r.ctx.set_uniforms(uniforms)
r.ctx.set_pixel_offsets_seed()
import torch  # noqa: E402
from toyraygun_amd.dist import microband_rows  # noqa: E402
r._unpack = lambda *a: None        # the bands' kernels only: the unpack is the same streaming pass for every rank
for n, mode in [(n, m) for n in ns for m in ("contiguous", "interleaved")]:
    alone, piped, rays = [], [], []
    r.world, r.interleaved = n, mode == "interleaved"
    if r.interleaved:
        stride = microband_rows(H, n, 0)[1]
        r.compact = [torch.zeros((n * stride, W, 4), dtype=torch.float32, device=r.device) for _ in r.frames]
    for g in range(n):
        r.rank = g
        r.row0, r.rows = band_rows(H, n, g)
        r.il_rows, r.il_stride = microband_rows(H, n, g)
        for _ in range(len(r.render_streams)):
            r.render(0, SPP, B, gather=False)
        r.synchronize()
        r.ctx.set_option(capi.OPT_TIMING, 1)
        r.ctx.reset_stats()
        for _ in range(3):
            r.launch_band(0, SPP, B)
        st = r.ctx.stats()
        alone.append(st.total_render_ms / 3)
        rays.append(st.rays / 3)
        r.ctx.set_option(capi.OPT_TIMING, 0)
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            r.render(0, SPP, B, gather=False)
        r.synchronize()
        piped.append((time.perf_counter() - t0) / K * 1e3)

    def spread(v):
        m = sum(v) / len(v)
        return {"per_band": [round(x, 4) for x in v], "max": max(v), "mean": m, "max_over_mean": max(v) / m}
    print(json.dumps({"config": name, "n": n, "bands": mode, "sum_pipelined_ms": sum(piped), "rows_per_band": -(-H // n), "alone_ms": spread(alone), "pipelined_ms_per_step": spread(piped),
                      "rays": spread(rays)}), flush=True)
r.close()
