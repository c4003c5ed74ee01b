#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box through gpurun).  Kernel-trace/stats passes and PMC passes
# are separate runs (gpurun refuses mixed ones); FETCH_SIZE and WRITE_SIZE need separate passes (TCC slot limit).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles; rm -rf "$OUT"; mkdir -p "$OUT"
C2="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline"
C4="python3 scripts/c4_run.py 3"
run() { name=$1; shift; rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -5 "$OUT/$name.log"; }; }
run c2_trace --kernel-trace --stats --output-format csv -d "$OUT/c2_trace" -- $C2
run c2_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/c2_fetch" -- $C2
run c2_write --pmc WRITE_SIZE --output-format csv -d "$OUT/c2_write" -- $C2
run c2_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/c2_sq1" -- $C2
run c2_sq2 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/c2_sq2" -- $C2
run c4_trace --kernel-trace --stats --output-format csv -d "$OUT/c4_trace" -- $C4
run c4_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/c4_fetch" -- $C4
run c4_write --pmc WRITE_SIZE --output-format csv -d "$OUT/c4_write" -- $C4
run c4_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/c4_tcc" -- $C4
run c4_sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/c4_sq" -- $C4
python3 - <<'PY'
import csv, glob, collections, json, os
out = "gpurun_out/profiles"
summary = {}
for cfg in ("c2", "c4"):
    counters = {}
    for f in glob.glob(f"{out}/{cfg}_*/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"] and ", true>" not in r["Kernel_Name"]:
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            counters.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
    stats = []
    for f in glob.glob(f"{out}/{cfg}_trace/*/*_kernel_stats.csv"):
        stats = [r for r in csv.DictReader(open(f))][:4]
        open(f"{out}/{cfg}_kernel_stats.csv", "w").write(open(f).read())
    summary[cfg] = {"kernel_stats_top": stats, "pmc_per_launch": counters}
# per-launch view of the C2 trace: bench.py runs 3 launches alone (HIP-event timed) and then keeps 4 in flight, so the
# kernel_stats average mixes two populations; list them apart (overlap = launches whose interval intersects another one's)
launches = []
for f in glob.glob(f"{out}/c2_trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel<true, false>" in r["Kernel_Name"]:
            launches.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
launches.sort()
alone, shared = [], []
for i, (a, b) in enumerate(launches):
    over = any(j != i and launches[j][0] < b and launches[j][1] > a for j in range(len(launches)))
    (shared if over else alone).append((b - a) / 1e6)
if launches:
    summary["c2"]["launches"] = {"alone_ms": alone, "overlapped_ms": shared,
                                 "overlapped_mean_ms": sum(shared) / max(len(shared), 1), "alone_mean_ms": sum(alone) / max(len(alone), 1)}
    print("c2 launches alone:", [round(x, 3) for x in alone], "overlapped:", [round(x, 3) for x in shared])
json.dump(summary, open(f"{out}/summary.json", "w"), indent=1)
for cfg in summary:
    for k, cs in summary[cfg]["pmc_per_launch"].items():
        print(cfg, k[:60], {c: round(v["mean"]) for c, v in cs.items()})
    for r in summary[cfg]["kernel_stats_top"][:2]:
        print(cfg, r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
