#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/c4prof
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 scripts/c4_run.py 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || tail -5 "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1 || tail -5 "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1 || tail -5 "$OUT/write.log"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d "$OUT/tcc" -- $CMD > "$OUT/tcc.log" 2>&1 || tail -5 "$OUT/tcc.log"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d "$OUT/tcp" -- $CMD > "$OUT/tcp.log" 2>&1 || tail -5 "$OUT/tcp.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.log" 2>&1 || tail -5 "$OUT/sq.log"
python3 - <<'PY'
import csv, glob, collections
for p in ("fetch","write","tcc","tcp","sq"):
    for f in glob.glob(f"gpurun_out/c4prof/{p}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items(): print(p, k, "%.5g" % (sum(v)/len(v)), "n=%d" % len(v))
for f in glob.glob("gpurun_out/c4prof/trace/*/*_kernel_stats.csv"):
    print(open(f).read()[:600])
PY
