#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + bounce + shadow rays) on the Cornell box.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] ("C2"): Cornell box (src/cornellBox.h), 1920x1080, 16 spp, 3 bounces,
synthetic inputs of SURVEY 8d (seeded per-pixel Halton offsets).  One STEP = one pass of the hot path
over the whole frame: ONE megakernel launch per GPU (raygen -> 3 x [nearest, shade, shadow] ->
accumulate for all 16 samples) and, for N > 1, one RCCL all-gather of the row bands.  The frame is
fixed, so N > 1 is STRONG scaling: rank g renders rows [g*h/N, (g+1)*h/N).

Rank 0 prints one JSON line.  `value` counts rays actually traversed (in-kernel counters), inputs
resident in HBM before the timed region.  `roofline` prices the megakernel with SURVEY 8(d)'s
algorithmic bytes per ray; `cpu_baseline` times the CPU oracle (a port: the reference has no CPU path)
on the host cores, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

# two launches overlap only when their streams sit on different hardware queues: HIP spreads streams round-robin over
# GPU_MAX_HW_QUEUES (default 4) queues, and this process has seven streams (four render, one communication, the
# context's own, torch's) plus RCCL's -- give every stream its own queue (measured with a process group up: 8 queues
# still collide, 1/8-frame steps 0.375 ms; 12 or more 0.28 ms).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP, BOUNCES = 1920, 1080, 16, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cornell_buffers():
    """Scene + uniforms from the PRODUCT's host library (libtoyraygun.so), not from the oracle."""
    from toyraygun_amd import host
    return host.Scene.cornell_box().buffers(), host.uniforms(W, H)[0]


from toyraygun_amd.roofline import algorithmic_bytes_per_ray  # SURVEY 8(d) figure from the kernel's counters


def cpu_baseline(buffers_unused):
    """The CPU oracle (project restatement of the Metal semantics, brute-force intersector over the 36
    triangles, OpenMP over row bands) on the same C2 workload, all host cores."""
    from oracle import pyoracle as O
    scene = O.OracleScene.cornell_box()
    # the GPU box gives one GPU a CPU share of 16 cores even though it shows more hardware threads
    threads = max(1, min(O.num_threads(), len(os.sched_getaffinity(0)), 16))
    t0 = time.perf_counter()
    _, st = O.render(scene, W, H, SPP, BOUNCES, nthreads=threads)
    dt = time.perf_counter() - t0
    return {"value": st.rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "the whole C2 frame (1920x1080, 16 spp, 3 bounces), %d rays in %.1f s" % (st.rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run (also with N = 1)
    if distributed or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))

    from toyraygun_amd import capi
    from toyraygun_amd.dist import DistributedRenderer

    buffers, uniforms = cornell_buffers()
    # four frame buffers on four alternating render streams (+ a communication stream): consecutive steps are independent
    # images, so the next steps fill the CUs that the tail of step k leaves idle (or that a small row band never fills),
    # and the gather of k overlaps the renders that follow
    r = DistributedRenderer(W, H, local_rank, pipelined=not os.environ.get("TRG_BENCH_SERIAL"))
    r.load_scene(buffers)
    r.ctx.set_uniforms(uniforms)
    r.ctx.set_pixel_offsets_seed()
    dev = r.device

    def sync_all():
        r.synchronize()
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # counters pass (untimed): algorithmic bytes per ray for the roofline object
    r.ctx.set_option(capi.OPT_COUNTERS, 1)
    r.ctx.reset_stats()
    r.ctx.render(0, SPP, BOUNCES, r.row0, r.rows)
    cst = r.ctx.stats()
    bytes_per_ray, mix = algorithmic_bytes_per_ray(cst, r.rows * W * SPP)
    rays_per_launch = cst.rays
    r.ctx.set_option(capi.OPT_COUNTERS, 0)

    gather = distributed or bool(os.environ.get("TRG_FORCE_GATHER"))
    for _ in range(args.warmup):
        r.render(0, SPP, BOUNCES, gather=gather)
    sync_all()
    # average launch duration of the megakernel: HIP events on the stream it runs on (trg_render brackets the launch with
    # hipEventRecord on the context's stream), measured here on 3 launches that run alone -- the event wait is a host
    # sync, which would serialise the pipeline inside the timed region (and there two launches overlap on purpose).
    r.ctx.reset_stats()
    for _ in range(3):
        r.ctx.render(0, SPP, BOUNCES, r.row0, r.rows)
    pst = r.ctx.stats()
    kernel_ms_pre = pst.total_render_ms / max(pst.renders, 1)
    r.ctx.set_option(capi.OPT_TIMING, 0)
    sync_all()
    r.ctx.reset_stats()
    r.time_launches = True   # HIP events around every launch, on the stream it runs on, no host sync
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.render(0, SPP, BOUNCES, gather=gather)
    sync_all()
    dt = time.perf_counter() - t0
    r.time_launches = False
    launch_ms = r.launch_ms()
    st = r.ctx.stats()

    rays_local = float(st.rays)
    # average launch duration over the timed region (what rocprofv3 --kernel-trace reports for the same command).  With
    # several launches in flight each one shares the GPU and lasts longer than alone; `concurrency` = sum of the launch
    # durations / wall time says how many overlapped, and kernel_ms / concurrency is the time the GPU spent per launch.
    kernel_ms = sum(launch_ms) / max(len(launch_ms), 1)
    concurrency = max(1.0, sum(launch_ms) / (dt * 1e3))
    if distributed:
        t = torch.tensor([dt, rays_local, kernel_ms / concurrency], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_total, kernel_eff_ms = float(tmax[0]), float(tsum[1]), float(tmax[2])
    else:
        rays_total = rays_local
        kernel_eff_ms = kernel_ms / concurrency

    if rank == 0:
        value = rays_total / dt / 1e6
        achieved = rays_per_launch * bytes_per_ray / (kernel_eff_ms * 1e-3) / 1e9  # GB/s, dominant kernel (slowest rank)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_c2.json")
        if world == 1 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary+shadow+bounce) at 1920x1080",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cornell box (36 triangles) 1920x1080, 16 spp, 3 bounces (BASELINE configs[1])",
                       "rays_per_step": rays_total / args.steps, "sharding": "row bands + RCCL all-gather (gather of step k overlaps the renders that follow)" if distributed else "none",
                       "pipeline": ("%d frames in flight on alternating streams: later steps fill the CUs the tail of step k leaves idle "
                                    "(roofline.kernel_ms = average launch duration while they overlap; kernel_alone_ms = one launch by itself)" % len(r.render_streams)) if getattr(r, "_overlap", False) else "serial launches",
                       "kernel": ("render_kernel" if st.last_frame_split <= 1 else "render_fp_kernel, %d frame lanes" % st.last_frame_split) + "<LDS scene> (fast build)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": kernel_ms, "launches_in_flight": concurrency, "kernel_ms_per_launch_effective": kernel_eff_ms,
                         "kernel_alone_ms": kernel_ms_pre, "kernel_alone_mrays_per_s": rays_per_launch / (kernel_ms_pre * 1e-3) / 1e6,
                         "algorithmic_bytes_per_ray": bytes_per_ray,
                         "bytes_per_launch": rays_per_launch * bytes_per_ray, **mix,
                         "note": "the scene (%.1f KB on the device) is LDS-resident:" % (cst.scene_bytes / 1024.0) + " the algorithmic bytes are served from LDS, so frac > 1 is expected; real HBM traffic (`traffic`, rocprofv3 FETCH_SIZE+WRITE_SIZE) is the offset read, the float4 write and spill scratch (SURVEY 8d caveat)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(buffers)
        print(json.dumps(out), flush=True)
    r.close()
    if distributed or launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
