"""Host-side cost of DistributedRenderer.render with the gather enqueued (single-rank RCCL group): is the host the limit
at 0.27 ms per step?  python scripts/gpu_host_overhead.py"""
import os, sys, time; sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16"); os.environ["TRG_FORCE_GATHER"] = "1"
import torch, torch.distributed as dist
from toyraygun_amd import capi, host
from toyraygun_amd.dist import DistributedRenderer, band_rows
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29561", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H = 1920, 1080
b = host.Scene.cornell_box().buffers()
r = DistributedRenderer(W, H, 0, pipelined=True)
r.load_scene(b); r.ctx.set_uniforms(host.uniforms(W, H)[0]); r.ctx.set_pixel_offsets_seed()
r.ctx.set_option(capi.OPT_TIMING, 0)
for n in (1, 4, 8):
    r.row0, r.rows = band_rows(H, n, n // 2)
    for gather in (False, True):
        for _ in range(8):
            r.render(0, 16, 3, gather=gather)
        r.synchronize()
        K = 60
        t0 = time.perf_counter()
        for _ in range(K):
            r.render(0, 16, 3, gather=gather)
        t_host = time.perf_counter() - t0
        r.synchronize()
        t_all = time.perf_counter() - t0
        print("rows 1/%d gather=%d: host enqueue %.3f ms/step, steady state %.3f ms/step" % (n, gather, t_host / K * 1e3, t_all / K * 1e3), flush=True)
dist.destroy_process_group()
