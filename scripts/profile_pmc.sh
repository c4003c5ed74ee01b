#!/bin/bash
# SQ-level PMC profile of the bench workload (two passes of <= 8 SQ counters).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/p1" -- $CMD > "$OUT/p1.log" 2>&1 || tail -5 "$OUT/p1.log"
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p2" -- $CMD > "$OUT/p2.log" 2>&1 || tail -5 "$OUT/p2.log"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 --output-format csv -d "$OUT/p3" -- $CMD > "$OUT/p3.log" 2>&1 || tail -5 "$OUT/p3.log"
python3 - <<'PY'
import csv, glob, collections
for p in ("p1","p2","p3"):
    for f in glob.glob(f"gpurun_out/pmc/{p}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render_kernel<true, false>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(p, k, "%.4g" % (sum(v)/len(v)), "n=%d" % len(v))
PY
