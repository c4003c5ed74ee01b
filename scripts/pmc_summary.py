"""Condenses the rocprofv3 passes of scripts/profile_round.sh:  python3 scripts/pmc_summary.py <dir> c2 [c4]
-> <dir>/summary.json (everything), <dir>/<cfg>_kernel_stats.csv (the --stats table) and <dir>/<cfg>_counters.json, the
per-launch counters of the benched megakernel that bench.py imports into its roofline object."""
import collections
import csv
import re
import glob
import json
import subprocess
import sys

out = sys.argv[1]
cfgs = sys.argv[2:] or ["c2", "c4"]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyraygun_amd.srchash import kernel_source_hash
commit = os.environ.get("TRG_COMMIT", "")
if not commit:
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = ""
summary = {}
for cfg in cfgs:
    counters = {}
    for f in glob.glob(f"{out}/{cfg}_*/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "trgk_" in k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            vs = sorted(v)
            counters.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v), "median": vs[len(vs) // 2]}
    stats = []
    for f in glob.glob(f"{out}/{cfg}_trace/*/*_kernel_stats.csv"):
        stats = [r for r in csv.DictReader(open(f))][:4]
        open(f"{out}/{cfg}_kernel_stats.csv", "w").write(open(f).read())
    summary[cfg] = {"kernel_stats_top": stats, "pmc_per_launch": counters}
    # per-launch view of the trace: bench.py runs a few launches alone (HIP-event timed) and then keeps 4 in flight, so the
    # kernel_stats average mixes two populations; list them apart (overlap = launches whose interval intersects another's)
    by_kernel = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{cfg}_trace/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "render" in r["Kernel_Name"] and "_kernel<" in r["Kernel_Name"]:
                by_kernel[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    launches = sorted(max(by_kernel.values(), key=len)) if by_kernel else []  # the timed instantiation (the COUNT one runs once)
    alone, shared = [], []
    for i, (a, b) in enumerate(launches):
        over = any(j != i and launches[j][0] < b and launches[j][1] > a for j in range(len(launches)))
        (shared if over else alone).append((b - a) / 1e6)
    if launches:
        summary[cfg]["launches"] = {"alone_ms": alone, "overlapped_ms": shared,
                                    "overlapped_mean_ms": sum(shared) / max(len(shared), 1), "alone_mean_ms": sum(alone) / max(len(alone), 1),
                                    "pipeline_ms_per_launch": (launches[-1][1] - launches[len(alone)][0]) / 1e6 / max(len(shared), 1) if shared else None}
        print(cfg, "launches alone:", [round(x, 3) for x in alone], "overlapped:", [round(x, 3) for x in shared])
    # ---- the counters file bench.py imports: EVERY field per bench STEP (one trg_render of the whole configuration), summed over every
    #      kernel launch the step consists of -- the megakernel alone (C2, C5), regeneration + its fold per 16-frame chunk (C4), head +
    #      tail levels + fold per chunk (C3).  The counted (`true>`) instantiations of the untimed counters pass and one-off helper kernels
    #      (offsets, probes) are left out; which kernels went in, and how often per step, is part of the record.
    if counters:
        import bench   # CONFIGS: spp of the profiled configuration -> chunks per step
        spp = bench.CONFIGS.get(cfg, {}).get("spp", 16)

        def timed(k):
            base = k.split("(")[0]
            if "trgk_" not in base or re.search(r"(offsets|xcc_probe|halton|raygen|sample|trace|postprocess|unpack_bands)_kernel", base):
                return False
            # the COUNT instantiations (the untimed counters pass): COUNT is the second template argument of render_kernel / render_fp_kernel /
            # render_pool_kernel / wf_trace_kernel, the first (or only) one of render_regen_kernel / render_head_kernel / render_tail_kernel
            if re.search(r"render_(kernel|fp_kernel|pool_kernel)<\w+, true", base) or re.search(r"wf_trace_kernel<\w+, true", base):
                return False
            if re.search(r"render_(regen|head|tail)_kernel<true", base):
                return False
            return True
        tk = {k: v for k, v in counters.items() if timed(k)}
        launches = lambda k: max(v["launches"] for v in tk[k].values())
        lead = [k for k in tk if "render_head_kernel" in k] or [k for k in tk if "render_regen_kernel" in k]
        if lead:
            chunks_per_step = -(-spp // 16)            # head / regeneration launches run one 16-frame chunk each (trg_capi.cpp kTailChunkFrames)
            steps = launches(lead[0]) / chunks_per_step
        else:
            lead = [max((k for k in tk if "render_" in k), key=launches)]
            chunks_per_step = 1
            steps = float(launches(lead[0]))
        tot = collections.defaultdict(float)
        for k, cs in tk.items():
            for n, v in cs.items():
                tot[n] += v["mean"] * v["launches"]
        per_step = {n: v / steps for n, v in tot.items()}
        c = per_step
        # FETCH_SIZE: rocprofv3 reports KiB and, on gfx950, tallies every 128-byte memory-side read request at 64 bytes
        # (TCC_BUBBLE reads 0): profiles/r02/fetch_calibration.md -- x2 for every access shape.  WRITE_SIZE is exact.
        fetch_raw = c.get("FETCH_SIZE", 0.0) * 1024.0
        fetch = 2.0 * fetch_raw
        write = c.get("WRITE_SIZE", 0.0) * 1024.0
        dram32 = c.get("TCC_EA0_RDREQ_DRAM_32B_sum")
        rec = {"unit": "EVERY counter below is per bench step (one trg_render of the whole configuration): the sum over the kernels listed in `kernels`",
               "kernel": " + ".join(sorted(k.split("(")[0].replace("void ", "").replace("trgk_regen_fast::", "").replace("trgk_fast::", "") for k in tk)),
               "kernels": {k.split("(")[0].replace("void ", "").replace("trgk_regen_fast::", "").replace("trgk_fast::", ""): {"launches_per_step": launches(k) / steps, "launches_profiled": launches(k),
                           "valu_insts_per_launch": tk[k].get("SQ_INSTS_VALU", {}).get("median"),
                           # lanes of 64 active per VALU instruction = thread-cycles / cycles a VALU instruction was active (round-4 verdict: divided by the
                           # instruction COUNT the full-wave fold kernels read 67 - 69 "of 64"; by SQ_ACTIVE_INST_VALU they read 63 - 64)
                           "lanes_active_per_valu_inst": (tk[k]["SQ_THREAD_CYCLES_VALU"]["median"] / tk[k]["SQ_ACTIVE_INST_VALU"]["median"]) if "SQ_THREAD_CYCLES_VALU" in tk[k] and tk[k].get("SQ_ACTIVE_INST_VALU", {}).get("median") else None}
                           for k in tk},
               "steps_profiled": steps, "chunks_per_step": chunks_per_step,
               "commit": commit, "kernel_source_hash": kernel_source_hash(),
               "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "salu_insts_per_launch": c.get("SQ_INSTS_SALU"), "lds_insts_per_launch": c.get("SQ_INSTS_LDS"),
               "lanes_active_per_valu_inst": (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None,
               "lanes_definition": "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (a kernel whose every VALU instruction has 64 live lanes reads 64)",
               "thread_cycles_valu": c.get("SQ_THREAD_CYCLES_VALU"), "active_inst_valu": c.get("SQ_ACTIVE_INST_VALU"),
               "fetch_bytes": fetch, "fetch_size_raw_bytes": fetch_raw, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
               "rdreq_dram_32b_bytes": (dram32 * 32.0) if dram32 else None,
               "wave_cycles": c.get("SQ_WAVE_CYCLES"), "wait_any_cycles": c.get("SQ_WAIT_ANY"), "wait_inst_any_cycles": c.get("SQ_WAIT_INST_ANY"), "busy_cycles": c.get("SQ_BUSY_CYCLES"),
               "lds_idx_active": c.get("SQ_LDS_IDX_ACTIVE"), "lds_bank_conflict": c.get("SQ_LDS_BANK_CONFLICT"),
               "tcc_hit": c.get("TCC_HIT_sum"), "tcc_miss": c.get("TCC_MISS_sum"), "ta_busy_avr": c.get("TA_BUSY_avr"), "gui_active": c.get("GRBM_GUI_ACTIVE"),
               "alone_mean_ms": summary[cfg].get("launches", {}).get("alone_mean_ms"),
               "note": "`*_per_launch` keeps its name for bench.py: a launch IS a step here (for C2 / C5 literally one kernel launch)",
               "source": "scripts/profile_round.sh at commit %s: one rocprofv3 --pmc pass per counter group of `python3 bench.py%s`, mean per launch x launches / steps profiled, summed over the step's kernels; "
                         "fetch_bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts each 128-byte request as 64: profiles/%s/fetch_calibration.md), write_bytes = WRITE_SIZE x 1024" % (commit, " --no-secondary" if cfg == "c2" else " --config " + cfg, "r02")}
        json.dump(rec, open(f"{out}/{cfg}_counters.json", "w"), indent=1)
        print(cfg, json.dumps({k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in rec.items() if k not in ("source", "kernels", "unit", "note")}))
    for r in stats[:2]:
        print(cfg, r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
