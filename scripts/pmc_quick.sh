#!/bin/bash
# quick counter comparison of a library variant on C4 (run on the GPU box): bash scripts/pmc_quick.sh <so path or ""> <tag>
# two rocprofv3 --pmc passes of scripts/c4_run.py; prints per-launch medians of the render kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
[ -n "$1" ] && export TRG_HIP_SO=$1
tag=${2:-shipped}
OUT=$PWD/gpurun_out/pmcq_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/a" -- python3 scripts/c4_run.py 3 > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY --output-format csv -d "$OUT/b" -- python3 scripts/c4_run.py 3 > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_" in r["Kernel_Name"] and "_kernel<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = sorted(agg[k]); print("%-24s %.4g" % (k, v[len(v) // 2]))
PY
grep "C4" "$OUT/a.log" | tail -1
