#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + bounce + shadow rays) on the Cornell box.

  python bench.py --gpus 1 --steps K --warmup W [--config c2|c3|c4|c5]
  python bench.py --gpus N ...                      (N > 1, no launcher: ONE process, one context + host thread per GPU -- trg_group_*)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W     (one process per GPU, torch.distributed over RCCL)

Default workload = BASELINE.json configs[1] ("C2"): Cornell box (src/cornellBox.h), 1920x1080, 16 spp, 3 bounces,
synthetic inputs of SURVEY 8d (seeded per-pixel Halton offsets).  `--config c4` = configs[3]: the same settings on the
1,022,244-triangle replicated-mesh scene (the only configuration whose working set lives in HBM); c3 / c5 = configs[2] /
configs[4] on one GPU.  One STEP = one pass of the hot path over the whole frame: ONE megakernel launch per GPU (raygen ->
bounces x [nearest, shade, shadow] -> accumulate for all samples) and, for N > 1, one RCCL all-gather of the row bands.
The frame is fixed, so N > 1 is STRONG scaling: rank g renders its band of rows.

Rank 0 prints ONE JSON line of about 3 KB.  `value` counts rays actually traversed (in-kernel counters), inputs resident in HBM
before the timed region.  The driver's record keeps the SCALAR values directly under `config`, `roofline` and `cpu_baseline` (nested
objects and long strings are dropped: BENCH_r04.parsed), so everything a reader needs to recompute a fraction is a flat scalar there:
  * `roofline`: the headline's binding resource -- bound, achieved, peak, unit, frac, traffic -- and beside it valu_insts_per_launch,
    lanes (of 64 active per VALU instruction), lane_weighted_frac, hbm_measured_frac, lds_frac / hbm_algorithmic_frac, kernel_alone_ms ...
  * `config.<leg>_*` for the secondary legs of the default run (N = 1, C2): c4 (BASELINE configs[3]: the scene lives in HBM / Infinity
    Cache), c3 (configs[2]: deep-bounce divergence) and c4xl (10.6 M triangles, 2.9 GB: the one leg whose memory-side
    traffic is DRAM traffic) -- mrays, ms_per_step, alone_ms, rays_per_step, bound, frac, valu_insts, valu_frac, lanes, traffic_bytes,
    hbm_frac (measured memory-side bytes per step / step time / 8 TB/s), alg_bytes (SURVEY 8(d)'s algorithmic bytes per step), l2_hit.
Everything long -- the CPU legs with their samples, notes, counter sources, per-band tables -- goes to a side file, named in
`config.detail_file` (gpurun_out/bench_detail_<config>_n<N>.json; TRG_BENCH_DETAIL overrides the path).

`roofline.bound` names the resource that BINDS the dominant kernel and `frac` is never above 1:
  * scene staged in LDS (C2/C3/C5): VALU issue -- wave-level VALU instructions per launch (rocprofv3 SQ_INSTS_VALU, imported from
    profiles/ and accepted only if the kernel sources hash to what they were measured on) x 2 cycles / (1024 SIMDs x 2.4 GHz) against
    the launch duration measured live with HIP events; without usable imported counters: "lds" (SURVEY 8(d)'s algorithmic bytes -- LDS
    reads here -- against the ~150 TB/s ds_read_b128 aggregate, measured entirely in this run).
  * scene in HBM (C4, c4xl): "hbm" = algorithmic bytes per launch / launch duration against 8 TB/s, unless that rate exceeds the peak (L2
    and the Infinity Cache serve it): then VALU issue as above, with the MEASURED memory-side traffic in `traffic` / `hbm_measured_frac`.
  * N > 1: the same bound as N = 1, DERIVED -- VALU instructions per ray of the N = 1 counters x the rays of the slowest band, against that
    band's device time per launch (`roofline.derived` says so).
`cpu_baseline` times the CPU oracle (a port: the reference has no CPU path) on the host cores, rank 0, N = 1 only; its legs are in the
side file.
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

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
LDS_PEAK_GBS = 150000.0     # aggregate ds_read_b64/b128 rate with every CU streaming (MI355X_MICROARCH.md, LDS)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0  # wave-level VALU instructions per ns: 1024 SIMDs, 2 cycles per wave64 instruction, 2.4 GHz
PROFILE_ROUNDS = ("r05", "r04", "r03", "r02")   # newest first: the first counters file whose kernel-source hash matches this tree is used

CONFIGS = {
    "c2": dict(w=1920, h=1080, spp=16, bounces=3, scene="cornell", short="C2: Cornell box 1920x1080, 16 spp, 3 bounces (BASELINE configs[1])",
               workload="Cornell box (36 triangles) 1920x1080, 16 spp, 3 bounces (BASELINE configs[1])"),
    "c3": dict(w=1920, h=1080, spp=256, bounces=8, scene="cornell", short="C3: Cornell box 1920x1080, 256 spp, 8 bounces (BASELINE configs[2])",
               workload="Cornell box (36 triangles) 1920x1080, 256 spp, 8 bounces (BASELINE configs[2], deep-bounce divergence stress)"),
    "c4": dict(w=1920, h=1080, spp=16, bounces=3, scene="lattice44", short="C4: Cornell box + 44^3 cubes = 1,022,244 triangles, 1920x1080, 16 spp, 3 b (BASELINE configs[3])",
               workload="Cornell box + 44x44x44 replicated cubes = 1,022,244 triangles, 1920x1080, 16 spp, 3 bounces (BASELINE configs[3], BVH/HBM stress)"),
    "c4xl": dict(w=1920, h=1080, spp=16, bounces=3, scene="lattice96", short="c4xl: Cornell box + 96^3 cubes = 10,616,868 triangles (2.9 GB), 1920x1080, 16 spp, 3 b (not in BASELINE)",
                 workload="Cornell box + 96x96x96 replicated cubes = 10,616,868 triangles (2.9 GB on the device: beyond the 256 MiB Infinity Cache), 1920x1080, 16 spp, 3 bounces "
                          "(not a BASELINE configuration: the leg where memory-side traffic is DRAM traffic)"),
    "c5": dict(w=3840, h=2160, spp=64, bounces=3, scene="cornell", short="C5: Cornell box 3840x2160, 64 spp, 3 bounces (BASELINE configs[4])",
               workload="Cornell box (36 triangles) 3840x2160, 64 spp, 3 bounces (BASELINE configs[4]; row bands over the GPUs)"),
}
# (config, steps, warmup, device builder) timed after the headline of the default run.  c4xl is built on the host since the box leaves (round 5:
# 1.9 s on 16 host threads against 0.08 s on the device -- the same split rule, but only the host builder recognises the cubes)
SECONDARY = (("c4", 5, 1, 0), ("c3", 5, 2, 0), ("c4xl", 3, 1, 0))


def sig(x, n=5):
    """A float with n significant digits (the JSON line stays short); ints, None and strings as they are."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (n, x))


def flat(d, n=5):
    return {k: sig(v, n) for k, v in d.items()}


def scene_buffers(cfg):
    """Scene + uniforms from the PRODUCT's host library (libtoyraygun.so), not from the oracle."""
    from toyraygun_amd import host
    sc = host.Scene.cornell_lattice(int(cfg["scene"][7:])) if cfg["scene"].startswith("lattice") else host.Scene.cornell_box()
    return sc.buffers(), host.uniforms(cfg["w"], cfg["h"])[0]


from toyraygun_amd.roofline import algorithmic_bytes_per_ray  # SURVEY 8(d) figure from the kernel's counters


def _oracle_leg(O, scene, name, w, h, spp, bounces, full_spp, threads, tuned=False):
    t0 = time.perf_counter()
    _, st = O.render(scene, w, h, spp, bounces, nthreads=threads, tuned=tuned)
    dt = time.perf_counter() - t0
    return {"config": name, "build": "tuned" if tuned else "checker", "value": st.rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "rays": int(st.rays), "seconds": round(dt, 3),
            "sample": "%dx%d, %d of %d spp, %d bounces" % (w, h, spp, full_spp, bounces),
            "extrapolation": "x%g in spp to the full configuration (cost is linear in spp: frames are independent)" % (full_spp / spp),
            "full_config_seconds_estimate": round(dt * full_spp / spp, 2)}


CPU_BUILDS = {
    "checker": "oracle/liboracle.so: gcc -O2 -ffp-contract=off (no -march), brute force over the 36 triangles of the Cornell box, its own median-split BVH on the lattice "
               "-- the bit-exact checker of the parity tests, timed as it is",
    "tuned": "oracle/liboracle_tuned.so: the same source at gcc -O3 -march=x86-64-v3 -ffp-contract=fast (contraction allowed), Halton with compile-time bases, walking "
             "THE PRODUCT'S BVH2 (trg_debug_build_bvh, host-only) on every scene -- BASELINE.md section 3's baseline; validated against the checker in tests/test_oracle.py",
}


def cpu_baseline(config_name):
    """The CPU oracle (project restatement of the Metal semantics; OpenMP over row bands) on the GPU box's host cores, in two builds of
    the same source (CPU_BUILDS): the bit-exact CHECKER as it is, and the TUNED build BASELINE.md section 3 plans (-O3, the same BVH as
    the HIP path).  Headline = the tuned build on the benched configuration on all cores; the other legs are the bounded samples
    SURVEY 8(d) / BASELINE.md ask for, with the extrapolation factor stated."""
    from oracle import pyoracle as O
    from toyraygun_amd import capi
    box = O.OracleScene.cornell_box()
    # the GPU box gives one GPU a CPU share of 16 cores even though it shows more hardware threads
    threads = max(1, min(O.num_threads(), len(os.sched_getaffinity(0)), 16))
    legs = []
    lattice = None
    if config_name in ("c4", "c4xl") or not os.environ.get("TRG_BENCH_FEW_CPU_LEGS"):
        lattice = O.OracleScene.cornell_lattice(44)
        O.render(lattice, 16, 16, 1, 1, nthreads=threads)   # builds the oracle's BVH (not part of the timed sample)

    def run_legs(tuned):
        legs.append(_oracle_leg(O, box, "C1", 256, 256, 1, 1, 1, 1, tuned))
        legs.append(_oracle_leg(O, box, "C2", 1920, 1080, 16, 3, 16, threads, tuned))
        legs.append(_oracle_leg(O, box, "C2", 1920, 1080, 1, 3, 16, 1, tuned))
        legs.append(_oracle_leg(O, box, "C3", 1920, 1080, 2, 8, 256, threads, tuned))
        if lattice is not None:
            legs.append(_oracle_leg(O, lattice, "C4", 1920, 1080, 1, 3, 16, threads, tuned))
        legs.append(_oracle_leg(O, box, "C5", 3840, 2160, 1, 3, 64, threads, tuned))
    run_legs(False)
    # the tuned build walks the product's BVH (host build, no GPU involved); handed to the oracle's scenes for these legs only
    for sc in (box, lattice):
        if sc is not None:
            b = sc.buffers()
            nodes, tris, _ = capi.debug_build_bvh(b["positions"], b["indices"], b["material_ids"])
            O.set_bvh(sc, nodes, tris)
    try:
        run_legs(True)
    finally:
        for sc in (box, lattice):
            if sc is not None:
                O.set_bvh(sc)
    want = "C4" if config_name == "c4xl" else config_name.upper()   # (c4xl has no CPU leg of its own: the C4 sample stands beside it)
    head = next(l for l in legs if l["config"] == want and l["cores"] == threads and l["build"] == "tuned")
    chk = next(l for l in legs if l["config"] == want and l["cores"] == threads and l["build"] == "checker")
    short = {"value": sig(head["value"]), "unit": "Mrays/s", "cores": threads, "kind": "port", "build": "tuned",
             "sample": "%s %s: %d rays in %.1f s" % (head["config"], head["sample"], head["rays"], head["seconds"]),
             "checker_value": sig(chk["value"]), "legs": len(legs)}
    detail = {"builds": CPU_BUILDS, "legs": legs,
              "note": "project CPU restatement of ToyRaygun's Metal semantics (oracle/trg_oracle.c), never a reference CPU path: the reference has none; "
                      "headline = the tuned build on the benched configuration on all cores, `checker_value` = the bit-exact checker as it is"}
    return short, detail


def imported_counters(config_name):
    """Per-launch PMC counters of the benched megakernel from the committed rocprofv3 passes (scripts/profile_round.sh ->
    profiles/<round>/<config>_counters.json).  bench.py cannot run the profiler on itself; the numbers are labelled imported,
    and they are only USED when the file's kernel-source hash equals this tree's (toyraygun_amd/srchash.py): instruction counts
    of a different kernel say nothing about this one.  Returns (counters or None, path, stale_counters or None)."""
    from toyraygun_amd.srchash import kernel_source_hash
    try:
        here = kernel_source_hash()
    except RuntimeError as e:   # no build record and no hipcc: said aloud, on stderr and in the line -- never a silent fallback
        sys.stderr.write("bench.py: %s -- imported profiler counters cannot be matched to this build and are NOT used\n" % e)
        return None, None, {"path": None, "reason": "kernel-source hash unavailable: %s" % e}
    stale = None
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, "%s_counters.json" % config_name)
        if not os.path.exists(path):
            continue
        try:
            rec = json.load(open(path))
        except Exception:
            continue
        rel = os.path.relpath(path, ROOT)
        if rec.get("kernel_source_hash") == here:
            return rec, rel, None
        if stale is None:
            stale = dict(rec, path=rel, reason="kernel sources changed since these counters were taken (hash %s, this tree %s)" % (rec.get("kernel_source_hash"), here))
    return None, None, stale


def build_roofline(config_name, cst, rays_per_launch, bytes_per_ray, mix, kernel_ms, concurrency, kernel_eff_ms, kernel_ms_alone, lane_util_nodes,
                   use_imported=True, derive=None):
    """The roofline of one timed configuration (module docstring): (flat scalars for the JSON line, verbose detail for the side file).
    derive = (rays of the band the line is about, rays of the whole frame): an N > 1 line takes the VALU instructions and the memory-side
    bytes PER RAY from the N = 1 counters of the configuration and scales them to the band -- labelled `derived`."""
    in_lds = bool(cst.scene_in_lds)
    sec = kernel_eff_ms * 1e-3                       # GPU time per launch in the timed region (slowest rank)
    bytes_per_launch = rays_per_launch * bytes_per_ray
    algorithmic_gbs = bytes_per_launch / sec / 1e9
    imp, imp_path, stale = imported_counters(config_name) if use_imported else (None, None, None)
    scale = (derive[0] / derive[1]) if (derive and derive[1]) else 1.0
    traffic = imp.get("hbm_bytes_per_launch") * scale if imp and imp.get("hbm_bytes_per_launch") else None
    valu_insts = imp.get("valu_insts_per_launch") * scale if imp and imp.get("valu_insts_per_launch") else None
    lanes = imp.get("lanes_active_per_valu_inst") if imp else None
    valu_g = valu_insts / sec / 1e9 if valu_insts else None
    valu_frac = valu_g / VALU_PEAK_GINST if valu_g else None
    hbm_meas_gbs = traffic / sec / 1e9 if traffic else None
    f = {"kernel_ms": kernel_ms, "launches_in_flight": concurrency, "kernel_ms_per_launch_effective": kernel_eff_ms,
         "kernel_alone_ms": kernel_ms_alone, "algorithmic_bytes_per_ray": bytes_per_ray, "bytes_per_launch": bytes_per_launch,
         "nodes_per_ray": mix["nodes_per_ray"], "tris_per_ray": mix["tris_per_ray"], "shaded_per_ray": mix["shaded_per_ray"],
         "rays_per_pixel_sample": mix["rays_per_pixel_sample"], "bytes_per_box": mix["bytes_per_box"],
         "traffic": traffic, "hbm_measured_frac": hbm_meas_gbs / HBM_PEAK_GBS if hbm_meas_gbs else None,
         "valu_insts_per_launch": valu_insts, "valu_frac": valu_frac, "lanes": lanes,
         "lane_weighted_frac": valu_frac * lanes / 64.0 if (valu_frac and lanes) else None}
    if imp:
        if imp.get("tcc_hit") and imp.get("tcc_miss"):
            f["l2_hit"] = imp["tcc_hit"] / (imp["tcc_hit"] + imp["tcc_miss"])
        if imp.get("wave_cycles"):
            f["wait_mem_frac"] = (imp.get("wait_any_cycles") or 0.0) / imp["wave_cycles"]
            f["wait_issue_frac"] = (imp.get("wait_inst_any_cycles") or 0.0) / imp["wave_cycles"]
        f["counters"] = "%s @%s" % (imp_path, imp.get("kernel_source_hash"))
    if derive:
        f["derived"] = "valu_insts / traffic = N=1 counters per ray x this band's rays"
    if stale:
        f["imported_stale"] = True
    if in_lds:
        f["lds_frac"] = algorithmic_gbs / LDS_PEAK_GBS
        if valu_g:
            head = {"bound": "valu_issue", "achieved": valu_g, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": valu_frac}
        else:
            head = {"bound": "lds", "achieved": algorithmic_gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": algorithmic_gbs / LDS_PEAK_GBS}
        note = ("LDS-resident scene: VALU-issue bound at partial lane utilisation; SURVEY 8(d)'s algorithmic bytes are LDS reads here (the %.1f KB scene is staged "
                "per workgroup), the HBM side only sees the 4-byte offset read and the 16-byte accumulation write per pixel" % (cst.scene_bytes / 1024.0))
    else:
        f["hbm_algorithmic_frac"] = algorithmic_gbs / HBM_PEAK_GBS    # (may exceed 1: L2 / Infinity Cache serve it -- never the headline frac then)
        note = ("algorithmic bytes (SURVEY 8d: 16 B per box of the 64-byte 4-wide node + 48 B per triangle test + 76 B per shaded hit) against the HBM peak; "
                "`traffic` = what left L2 towards Infinity Cache / HBM (2 x FETCH_SIZE + WRITE_SIZE, per step)")
        if algorithmic_gbs > HBM_PEAK_GBS and valu_g:
            head = {"bound": "valu_issue", "achieved": valu_g, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": valu_frac}
            note += "; the algorithmic rate is above the HBM peak (the caches serve it), so the primary bound is VALU issue"
        elif algorithmic_gbs > HBM_PEAK_GBS:
            head = {"bound": "hbm", "achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None}
            note += "; the algorithmic rate is ABOVE the HBM peak because the caches serve most of it: no fraction of anything (frac null), and no usable instruction counters"
        else:
            head = {"bound": "hbm", "achieved": algorithmic_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algorithmic_gbs / HBM_PEAK_GBS}
    detail = {"note": note, "lane_utilisation_node_loop": lane_util_nodes,
              "imported": ("%s (rocprofv3 --pmc passes of this build, kernel-source hash %s: %s)" % (imp_path, imp.get("kernel_source_hash"), imp.get("source", "")) if imp else None),
              "imported_record": imp,
              "imported_stale_detail": ({k: stale.get(k) for k in ("path", "reason", "commit", "valu_insts_per_launch", "hbm_bytes_per_launch")} if stale else None),
              "peaks": {"hbm_gbs": HBM_PEAK_GBS, "lds_gbs": LDS_PEAK_GBS, "valu_ginst": VALU_PEAK_GINST,
                        "valu_source": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md)"}}
    return flat({**head, **f}), detail


def leg_keys(name, s):
    """The flat `config.<leg>_*` scalars of one secondary leg (module docstring) from its run_torch result."""
    rf = s["roofline"]
    k = {"mrays": s["value"], "ms_per_step": s["ms_per_step"], "steps": s["steps"], "alone_ms": rf["kernel_alone_ms"], "rays_per_step": s["config"]["rays_per_step"],
         "tris": s["config"]["triangles"], "bound": rf["bound"], "frac": rf["frac"], "valu_insts": rf.get("valu_insts_per_launch"), "valu_frac": rf.get("valu_frac"),
         "lanes": rf.get("lanes"), "traffic_bytes": rf.get("traffic"), "hbm_frac": rf.get("hbm_measured_frac"), "alg_bytes": rf["bytes_per_launch"],
         "l2_hit": rf.get("l2_hit"), "build_ms": s["config"].get("build_ms")}
    return {"%s_%s" % (name, a): sig(b) for a, b in k.items() if b is not None}


def plugin_leg(frames=None):
    """The reference's OWN hot loop through the drop-in boundary (round-4 verdict, item 5): Engine::init(1024, 768) -> createRenderer()->init()
    -> loadScene(createCornellBoxScene()) -> K x renderFrame() -> finish(), exactly the call sequence of main.cpp:21-95, through
    libtoyraygun.so (toyraygun_amd/csrc/host/capi_host.cpp trh_run_app) -- next to the same K samples through ONE renderFrames(K) = one
    trg_render.  renderFrame() never waits (MetalRenderer.mm:377,385-387: three launches in flight); frames accepted while the device is busy
    share the next launch (HipRenderer.cpp).  Wall time from the first renderFrame() to the completion of the last frame on the device."""
    from toyraygun_amd import host
    W, H = 1024, 768                                   # main.cpp:22
    K = frames or int(os.environ.get("TRG_BENCH_PLUGIN_FRAMES", "1024"))
    host.run_app(W, H, 64, 3, batch=False)             # warm-up: library, code objects, allocator
    best = None
    for _ in range(3):                                 # (the loop is a few tens of milliseconds: the best of three, both ways)
        _, ms, rays, launches = host.run_app(W, H, K, 3, batch=False, want_launches=True)
        if best is None or ms < best[0]:
            best = (ms, rays, launches)
    bbest = None
    for _ in range(3):
        _, msb, raysb, lb = host.run_app(W, H, K, 3, batch=True, want_launches=True)
        if bbest is None or msb < bbest[0]:
            bbest = (msb, raysb, lb)
    ms, rays, launches = best
    msb, raysb, lb = bbest
    f = {"plugin_frames": K, "plugin_loop_ms": ms, "plugin_fps": K / (ms * 1e-3), "plugin_mrays": rays / (ms * 1e-3) / 1e6, "plugin_launches": launches,
         "plugin_frames_per_launch": K / max(launches, 1), "plugin_batched_ms": msb, "plugin_batched_mrays": raysb / (msb * 1e-3) / 1e6,
         "plugin_loop_over_batched": ms / msb}
    d = {"what": "main.cpp:21-95 call sequence at 1024x768, 1 spp per renderFrame(), 3 bounces, through libtoyraygun.so; batched = the same samples through one renderFrames(K)",
         "rays_loop": rays, "rays_batched": raysb, "launches_batched": lb}
    return flat(f), d


def write_detail(config_name, n, detail):
    """The long half of the result -> a side file; returns its path relative to the repo (or None when nothing could be written)."""
    path = os.environ.get("TRG_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail_%s_n%d.json" % (config_name, n))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(detail, fh, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        sys.stderr.write("bench.py: detail file %s not written: %s\n" % (path, e))
        return None


def kernel_name_long(st, in_lds):
    return (("render_regen_kernel (path regeneration: a job pool per workgroup) + regen_accumulate_kernel" if getattr(st, "last_regen", 0)
             else "render_kernel" if st.last_frame_split <= 1 else "render_fp_kernel, %d frame lanes" % st.last_frame_split)
            + ("<LDS scene>" if in_lds else "<HBM scene, quantised 4-wide BVH, 128-byte leaf records>")
            + (" + tail compaction from bounce %d (render_head / render_tail kernels)" % st.last_tail_bounce if st.last_tail_bounce else "") + " (fast build)")


def kernel_name(st, in_lds):
    """Short form for the JSON line (the driver's record cuts strings at ~120 characters)."""
    k = ("render_regen_kernel+regen_accumulate" if getattr(st, "last_regen", 0) else "render_kernel" if st.last_frame_split <= 1 else "render_fp_kernel/%d lanes" % st.last_frame_split)
    return k + ("<LDS>" if in_lds else "<HBM>") + ("+head/tail from bounce %d" % st.last_tail_bounce if st.last_tail_bounce else "") + " fast build"


def run_torch(config_name, steps, warmup, env, use_imported=True, gpu_build=None):
    """One configuration on this process' GPU (N = 1) or on its row band (one process per GPU under torch.distributed.run).
    Returns (result dict, detail dict) on rank 0, (None, None) elsewhere."""
    import torch
    from toyraygun_amd import capi
    from toyraygun_amd.dist import DistributedRenderer
    dist, distributed, rank, world, local_rank = env["dist"], env["distributed"], env["rank"], env["world"], env["local_rank"]
    cfg = CONFIGS[config_name]
    W, H, SPP, BOUNCES = cfg["w"], cfg["h"], cfg["spp"], cfg["bounces"]
    buffers, uniforms = scene_buffers(cfg)
    # four frame buffers on four alternating render streams (+ a communication stream): consecutive steps are independent
    # images, so the next steps fill the CUs that the tail of step k leaves idle (or that a small row band never fills),
    # and the gather of k overlaps the renders that follow
    r = DistributedRenderer(W, H, local_rank, pipelined=not os.environ.get("TRG_BENCH_SERIAL"))
    if gpu_build is None:
        gpu_build = int(os.environ.get("TRG_BENCH_GPU_BUILD", "0"))   # 1 = device binned SAH, 2 = LBVH, 3 = PLOC (default: the host SAH builder)
    if gpu_build:
        r.ctx.set_option(capi.OPT_GPU_BUILD, gpu_build)
    t_load = time.perf_counter()
    r.load_scene(buffers)
    load_s = time.perf_counter() - t_load
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
    r.ctx.set_option(capi.OPT_TIMING, 1)
    r.ctx.set_option(capi.OPT_COUNTERS, 1)
    r.ctx.reset_stats()
    r.launch_band(0, SPP, BOUNCES)
    cst = r.ctx.stats()
    bytes_per_ray, mix = algorithmic_bytes_per_ray(cst, r.owned_rows * W * SPP)
    rays_per_launch = cst.rays
    lane_util_nodes = cst.node_fetches / (64.0 * cst.wave_node_iters * (1 if cst.scene_in_lds else 2)) if cst.wave_node_iters else None
    r.ctx.set_option(capi.OPT_COUNTERS, 0)

    gather = distributed or bool(os.environ.get("TRG_FORCE_GATHER"))
    # untimed priming: one step per render stream, so that every stream's per-launch scratch (traversal-stack overflow, tail-compaction
    # queues) exists before anything is timed -- the first use of a stream allocates it behind a device synchronise
    for _ in range(len(r.render_streams)):
        r.render(0, SPP, BOUNCES, gather=False)
    sync_all()
    for _ in range(warmup):
        r.render(0, SPP, BOUNCES, gather=gather)
    sync_all()
    # launch duration of the megakernel when it runs ALONE: HIP events on the stream it runs on (trg_render brackets the
    # launch with hipEventRecord on the context's stream), 3 launches -- the event wait is a host sync, which would
    # serialise the pipeline inside the timed region (and there several launches overlap on purpose).
    r.ctx.reset_stats()
    for _ in range(3):
        r.launch_band(0, SPP, BOUNCES)
    pst = r.ctx.stats()
    kernel_ms_alone = pst.total_render_ms / max(pst.renders, 1)
    r.ctx.set_option(capi.OPT_TIMING, 0)
    sync_all()
    r.ctx.reset_stats()
    r.time_launches = True   # HIP events around every launch, on the stream it runs on, no host sync
    # THE TIMED REGION: `steps` steps, the exchange north_star names behind each (N > 1: the gather of the bands to rank 0, overlapping the
    # renders that follow)
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render(0, SPP, BOUNCES, gather=gather)
    sync_all()
    dt = time.perf_counter() - t0
    r.time_launches = False
    launch_ms = r.launch_ms()
    st = r.ctx.stats()

    def timed_again(mode):   # the same steps with another exchange, after the timed region: reported beside the headline
        t1 = time.perf_counter()
        for _ in range(steps):
            r.render(0, SPP, BOUNCES, gather=mode)
        sync_all()
        return time.perf_counter() - t1
    dt_all = timed_again("all") if gather else None
    dt_none = timed_again(False) if gather else None

    rays_local = float(st.rays)
    # average launch duration over the timed region (what rocprofv3 --kernel-trace reports for the same command).  With
    # several launches in flight each one shares the GPU and lasts longer than alone; `concurrency` = sum of the launch
    # durations / wall time says how many overlapped, and kernel_ms / concurrency is the time the GPU spent per launch.
    kernel_ms = sum(launch_ms) / max(len(launch_ms), 1)
    concurrency = max(1.0, sum(launch_ms) / (dt * 1e3))
    band_rays = float(rays_per_launch)
    if distributed:
        cpu = env.get("backend") == "gloo"
        t = torch.tensor([dt, rays_local, kernel_ms / concurrency, dt_all or 0.0, dt_none or 0.0], dtype=torch.float64, device="cpu" if cpu else dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        # the slowest rank's band: its device time per launch and ITS rays (for the derived roofline)
        mine = torch.tensor([kernel_ms / concurrency, band_rays, kernel_ms, concurrency, kernel_ms_alone], dtype=torch.float64, device="cpu" if cpu else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        slow = max(range(world), key=lambda k: float(allr[k][0]))
        dt, rays_total, kernel_eff_ms = float(tmax[0]), float(tsum[1]), float(tmax[2])
        dt_all, dt_none = float(tmax[3]), float(tmax[4])
        band_rays, kernel_ms, concurrency, kernel_ms_alone = float(allr[slow][1]), float(allr[slow][2]), float(allr[slow][3]), float(allr[slow][4])
        frame_rays = sum(float(a[1]) for a in allr)
    else:
        rays_total = rays_local
        kernel_eff_ms = kernel_ms / concurrency
        frame_rays = band_rays

    out, detail = None, None
    if rank == 0:
        in_lds = bool(cst.scene_in_lds)
        rf, rf_detail = build_roofline(config_name, cst, band_rays, bytes_per_ray, mix, kernel_ms, concurrency, kernel_eff_ms, kernel_ms_alone, lane_util_nodes,
                                       use_imported=use_imported, derive=(band_rays, frame_rays) if world > 1 else None)
        builder = {0: "host SAH", 1: "device binned SAH", 2: "device LBVH", 3: "device PLOC"}[gpu_build]
        config = {"workload": cfg["short"], "name": config_name, "rays_per_step": rays_total / steps,
                  "sharding": ("none" if not distributed else "%s bands, 1 process/GPU, %s gather to rank 0 per frame" % ("interleaved 8-row" if r.interleaved else "contiguous row", "gloo (host memory: rehearsal)" if env.get("backend") == "gloo" else "RCCL")),
                  "pipeline": "%d frames in flight on alternating streams" % len(r.render_streams) if getattr(r, "_overlap", False) else "serial launches",
                  "kernel": kernel_name(st, in_lds), "triangles": int(len(buffers["material_ids"])), "scene_bytes": int(cst.scene_bytes),
                  "builder": builder, "build_ms": sig(cst.last_build_ms), "load_scene_s": sig(load_s), "bvh_depth4": int(cst.bvh_depth4), "nodes4": int(cst.bvh_nodes4),
                  # geometry-specific leaves, said next to the number: parallelogram pairs tested as one quad, addCube groups that are one box leaf (LDS scenes)
                  "quads": int(cst.bvh_quads), "box_leaves": int(cst.bvh_boxes)}
        if gather:
            config.update({"exchange": "root gather" if r.gather_mode == "root" else "all-gather",
                           "gather_root_ms_per_step": sig(max(0.0, (dt - dt_none) / steps * 1e3)), "gather_all_ms_per_step": sig(max(0.0, (dt_all - dt_none) / steps * 1e3)),
                           "ms_per_step_all_gather": sig(dt_all / steps * 1e3), "ms_per_step_without_gather": sig(dt_none / steps * 1e3)})
            if distributed and env.get("backend") != "gloo":
                try:
                    v = torch.cuda.nccl.version()
                    config["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
                except Exception:
                    pass
                config["rccl_ranks"] = dist.get_world_size()
        out = {
            "metric": "Mrays/s (primary+shadow+bounce) at %dx%d" % (W, H),
            "value": sig(rays_total / dt / 1e6, 7), "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": sig(dt / steps * 1e3, 6), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config, "roofline": rf,
        }
        detail = {"workload": cfg["workload"], "roofline": rf_detail,
                  "pipeline": "frames in flight on alternating streams: later steps fill the CUs the tail of step k leaves idle (roofline.kernel_ms = average launch duration "
                              "while they overlap; kernel_alone_ms = one launch by itself)",
                  "kernel": kernel_name_long(st, in_lds)}
    r.close()
    return out, detail


def run_group(args, n=None, use_imported=False):
    """`--gpus N` without a launcher: ONE process, trg_group_* (one context + one host thread per device, row bands, one
    in-place ncclAllGather per frame over xGMI).  Same method as the N = 1 headline (run_torch): TRG_PIPE_DEPTH (4) frames in flight per
    device -- frame i renders on stream i % 4 into buffer i % 4, its exchange follows on the device's exchange stream and overlaps the
    renders behind it (trg_group_set_pipeline) -- HIP events around every launch on its own stream, no host sync inside the timed
    region.  Same JSON line; `config.bands` carries what each rank did: rays, the band's launch alone, its launches in the pipeline."""
    import torch   # device count only: nothing here initialises the GPU before the library does
    from toyraygun_amd import capi
    n = n or args.gpus
    have = torch.cuda.device_count()
    # TRG_BENCH_DEVICES="0,0": an explicit device list -- with TRG_GROUP_EXCHANGE=copy the contexts of a group may share a device, which
    # rehearses this whole path (bands, host threads, exchange by peer copies, this function) on a one-GPU box; never a scaling number
    devices = [int(x) for x in os.environ.get("TRG_BENCH_DEVICES", "").split(",") if x.strip()] or list(range(n))
    if len(devices) != n:
        raise SystemExit("bench.py --gpus %d: TRG_BENCH_DEVICES names %d device(s)" % (n, len(devices)))
    if max(devices) >= have or min(devices) < 0 or (len(set(devices)) == n and have < n):
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (n, have))
    shared = len(set(devices)) < n
    cfg = CONFIGS[args.config]
    W, H, SPP, BOUNCES = cfg["w"], cfg["h"], cfg["spp"], cfg["bounces"]
    buffers, uniforms = scene_buffers(cfg)
    g = capi.Group(devices, W, H)
    try:
        g.load_scene(buffers["positions"], buffers["normals"], buffers["colors"], buffers["indices"], buffers["material_ids"])
        g.set_uniforms(uniforms)
        g.set_pixel_offsets_seed()
        # interleaved 8-row micro-bands even out what the bands cost (TRG_BANDS=contiguous: rows [g*B, (g+1)*B) per device)
        interleaved = n > 1 and os.environ.get("TRG_BANDS", "interleaved") != "contiguous"
        g.set_bands(capi.BANDS_INTERLEAVED if interleaved else capi.BANDS_CONTIGUOUS)
        # counters pass (untimed, one buffer, synchronous): rays per step and the algorithmic bytes per ray, summed over the bands
        g.set_option(capi.OPT_COUNTERS, 1)
        g.render(0, SPP, BOUNCES, capi.GATHER_NONE)
        g.sync()
        cst = g.stats()
        bytes_per_ray, mix = algorithmic_bytes_per_ray(cst, H * W * SPP)
        rays_per_step = float(cst.rays)
        per_rank_rays = g.rank_rays()
        g.set_option(capi.OPT_COUNTERS, 0)
        # every band's launch ALONE on its device (HIP events around the launch; the ranks run at the same time, each on its own GPU)
        g.reset_stats()
        for _ in range(3):
            g.render(0, SPP, BOUNCES, capi.GATHER_NONE)
        alone = []
        for r in range(n):
            st = g.rank_stats(r)
            alone.append(st.total_render_ms / max(st.renders, 1))
        kernel_ms_alone = max(alone)
        depth = 1 if os.environ.get("TRG_BENCH_SERIAL") else int(os.environ.get("TRG_PIPE_DEPTH", "4"))
        g.set_option(capi.OPT_TIMING, 0)          # asynchronous from here on: a step only enqueues
        g.set_pipeline(depth)
        fences = min(8, 2 * depth)                # host run-ahead bound (MetalRenderer.mm:377: a semaphore of three)

        def timed(gather, k, timeit=False):
            g.sync()
            g.time_launches(timeit)
            t0 = time.perf_counter()
            for i in range(k):
                g.fence_wait(i % fences)
                g.render(0, SPP, BOUNCES, gather)
                g.fence_record(i % fences)
            g.sync()
            dt_ = time.perf_counter() - t0
            g.time_launches(False)
            return dt_

        timed(capi.GATHER_NONE, depth)            # priming: every render stream's per-launch scratch exists before anything is timed
        # THE TIMED REGION: north_star's exchange -- the GATHER of the bands to device 0 (grouped ncclSend / ncclRecv: the root ingests N - 1 bands
        # over N - 1 independent xGMI links) -- behind every step; the in-place all-gather and no exchange at all are timed after it and reported beside it
        timed(capi.GATHER_ROOT, max(args.warmup, 1))
        g.reset_stats()
        dt = timed(capi.GATHER_ROOT, args.steps, timeit=True)
        launch = [g.launch_ms(r) for r in range(n)]
        dt_all = timed(capi.GATHER_ALL, args.steps)
        dt_nogather = timed(capi.GATHER_NONE, args.steps)
        in_lds = bool(cst.scene_in_lds)
        exchange = g.exchange
        st0 = g.rank_stats(0)
        piped = [sum(l) / max(len(l), 1) for l in launch]               # mean launch duration per rank while its frames overlap
        conc = [max(1.0, sum(l) / (dt * 1e3)) for l in launch]          # launches of a rank in flight on average
        eff = [p / c for p, c in zip(piped, conc)]                      # device time per launch and rank
        slow = max(range(n), key=lambda r: eff[r])
        rf, rf_detail = build_roofline(args.config, cst, float(per_rank_rays[slow]), bytes_per_ray, mix, piped[slow], conc[slow], eff[slow], alone[slow], None,
                                       use_imported=use_imported, derive=(float(per_rank_rays[slow]), rays_per_step) if n > 1 else None)

        def spread(v):
            m = sum(v) / len(v)
            return {"per_rank": [round(x, 4) for x in v], "slowest": max(v), "mean": m, "max_over_mean": (max(v) / m) if m > 0 else None}
        rccl = g.rccl_info()   # {"version", "ranks": [ncclCommCount per communicator]} -- None without communicators (one device, copy exchange)
        xname = {capi.EXCHANGE_RCCL: "rccl", capi.EXCHANGE_COPY: "copy"}.get(exchange, "none")
        config = {"workload": cfg["short"], "name": args.config, "rays_per_step": rays_per_step,
                  "sharding": "1 process, trg_group: %d contexts + host threads, %s bands" % (n, "interleaved 8-row" if interleaved else "contiguous ceil(h/N)-row"),
                  "pipeline": "%d frames in flight per device, exchange on its own stream" % depth if depth > 1 else "serial launches",
                  "exchange": xname, "exchange_mode": "root gather (ncclSend/ncclRecv to device 0)" if xname == "rccl" else ("root gather (peer copies)" if xname == "copy" else "none"),
                  "rccl_version": rccl["version"] if rccl else None, "rccl_ranks": min(rccl["ranks"]) if rccl else None,
                  "devices": ",".join(str(d) for d in devices), "shared_device": shared,
                  "gather_root_ms_per_step": sig(max(0.0, (dt - dt_nogather) / args.steps * 1e3)),
                  "gather_all_ms_per_step": sig(max(0.0, (dt_all - dt_nogather) / args.steps * 1e3)),
                  "ms_per_step_all_gather": sig(dt_all / args.steps * 1e3), "ms_per_step_without_gather": sig(dt_nogather / args.steps * 1e3),
                  "band_ms_slowest": sig(max(eff)), "band_ms_max_over_mean": sig(max(eff) / (sum(eff) / n)), "band_rays_max_over_mean": sig(max(per_rank_rays) / (sum(per_rank_rays) / n)),
                  "kernel": ("bands of: " if n > 1 else "") + kernel_name(st0, in_lds),
                  "triangles": int(len(buffers["material_ids"])), "scene_bytes": int(cst.scene_bytes)}
        out = {
            "metric": "Mrays/s (primary+shadow+bounce) at %dx%d" % (W, H),
            "value": sig(rays_per_step * args.steps / dt / 1e6, 7), "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sig(dt / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": config, "roofline": rf,
        }
        detail = {"workload": cfg["workload"], "roofline": rf_detail, "exchange_note": g.exchange_note or None, "rccl": rccl,
                  "per_rank_rays_per_step": per_rank_rays,
                  "bands": {"rays": spread([float(x) for x in per_rank_rays]),
                            "kernel_alone_ms": spread(alone), "kernel_ms_in_pipeline": spread(piped), "device_ms_per_launch": spread(eff),
                            "note": ("the contexts SHARE a device (rehearsal): their `alone` launches ran side by side on it" if shared else
                                     "HIP events per rank on the stream of each launch; alone = 3 launches per band with nothing else on its GPU")},
                  "kernel": ("per-band launches of: " if n > 1 else "") + kernel_name_long(st0, in_lds),
                  "exchange_model": "DESIGN.md section 5: bytes per device and per xGMI link, expected ms and efficiency for the root gather and the all-gather"}
        return out, detail
    finally:
        g.close()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C4 / C3 legs the default run adds under `secondary`")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run (also with N = 1)
    if not launched and (args.gpus > 1 or os.environ.get("TRG_BENCH_GROUP")):
        # plain invocation: the single-process design of SURVEY 8(e); nothing is re-executed.  (TRG_BENCH_GROUP=1: N = 1 through the same
        # group path, to check it against the headline)
        out, detail = run_group(args, use_imported=True)
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], detail["cpu_baseline"] = cpu_baseline(args.config)
        out["config"]["detail_file"] = write_detail(args.config, args.gpus, {"line": out, **detail})
        print(json.dumps(out), flush=True)
        return
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or plainly (no launcher) for the single-process group" % (args.gpus, world, args.gpus))

    import torch
    # (a LAUNCHED run of one rank with TRG_FORCE_GATHER=1 takes every N > 1 branch -- communicator, barrier, reductions, the exchange per
    # frame -- through RCCL with a one-rank communicator: what a one-GPU box can run of the path the driver's scaling run takes)
    distributed = world > 1 or (launched and bool(os.environ.get("TRG_FORCE_GATHER")))
    dist = None
    # TRG_BENCH_BACKEND=gloo + TRG_BENCH_DEVICES="0,0": the launched multi-rank path rehearsed on a one-GPU box -- the ranks share device 0
    # (NCCL refuses such a communicator) and exchange their bands through host memory; never a scaling number
    backend = os.environ.get("TRG_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("TRG_BENCH_BACKEND must be nccl or gloo")
    devmap = [int(x) for x in os.environ.get("TRG_BENCH_DEVICES", "").split(",") if x.strip()]
    if devmap and len(devmap) != world:
        raise SystemExit("TRG_BENCH_DEVICES names %d device(s) for %d rank(s)" % (len(devmap), world))
    device_index = devmap[local_rank] if devmap else local_rank
    if distributed or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo")
    env = dict(dist=dist, distributed=distributed, rank=rank, world=world, local_rank=device_index, backend=backend if (distributed or launched) else None)

    out, detail = run_torch(args.config, args.steps, args.warmup, env)
    secondary_failed = []
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_secondary:
        # the configurations where the roofline question lives, timed in the same (driver-witnessed) run: same method, fewer steps.  Their
        # numbers go into `config` as flat scalars (module docstring), their verbose halves into the side file
        detail["secondary"] = {}
        for name, k, wu, builder in SECONDARY:
            try:
                s_out, s_detail = run_torch(name, k, wu, env, gpu_build=builder)
                out["config"].update(leg_keys(name, s_out))
                detail["secondary"][name] = {"line": s_out, **s_detail}
            except Exception as e:   # the headline is still printed -- and the run then ends NON-ZERO: a failed leg is not a green run
                out["config"]["%s_error" % name] = ("%s: %s" % (type(e).__name__, e))[:110]
                secondary_failed.append(name)
                break                # (a device error leaves the context in doubt: no further leg on it)
        if not secondary_failed and not os.environ.get("TRG_BENCH_NO_PLUGIN"):
            try:
                p_flat, p_detail = plugin_leg()
                out["config"].update(p_flat)
                detail["plugin"] = p_detail
            except Exception as e:
                out["config"]["plugin_error"] = ("%s: %s" % (type(e).__name__, e))[:110]
                secondary_failed.append("plugin")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], detail["cpu_baseline"] = cpu_baseline(args.config)
        out["config"]["detail_file"] = write_detail(args.config, world, {"line": out, **detail})
        print(json.dumps(out), flush=True)
    if distributed or launched:
        dist.barrier()
        dist.destroy_process_group()
    if secondary_failed:
        sys.stderr.write("bench.py: secondary leg(s) failed: %s\n" % ", ".join(secondary_failed))
        sys.exit(1)


if __name__ == "__main__":
    main()
