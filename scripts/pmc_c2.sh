#!/bin/bash
# SQ counters of the C2 render kernel for a library variant (run on the GPU box):  bash scripts/pmc_c2.sh <tag> [exp_build variant]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
tag=${1:-shipped}
[ -n "$2" ] && export TRG_HIP_SO=$PWD/exp_build/$2/libtoyraygun_hip.so
OUT=$PWD/gpurun_out/pmc_c2_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel<true, false>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
for k in sorted(m): print("%-26s %.5g" % (k, m[k]))
if "SQ_INSTS_VALU" in m: print("lanes per VALU instruction %.2f; LDS / VALU instructions %.3f; ready-without-slot %.3f, parked %.3f of the wave cycles" % (
    m["SQ_THREAD_CYCLES_VALU"] / m["SQ_INSTS_VALU"], m["SQ_INSTS_LDS"] / m["SQ_INSTS_VALU"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
grep -o '"value": [0-9.]*, "unit": "Mrays/s", "n_gpus": 1, "steps": 6, "warmup": 1, "ms_per_step": [0-9.]*' "$OUT/sq.log" | tail -1
