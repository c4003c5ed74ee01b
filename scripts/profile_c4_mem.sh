#!/bin/bash
# memory-pipeline counters of the C4 megakernel (TA / TCP / UTCL1), one rocprofv3 --pmc pass per group
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c4mem; rm -rf "$OUT"; mkdir -p "$OUT"
C4="python3 scripts/c4_run.py 2"
run() { name=$1; shift; echo "pass $name"; timeout -k 10 150 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; grep -m2 -i "error code\|exceeds" "$OUT/$name.log"; }; }
run ta1 --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta1" -- $C4
run ta2 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d "$OUT/ta2" -- $C4
run tcp1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/tcp1" -- $C4
run tcp2 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/tcp2" -- $C4
run tcp3 --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d "$OUT/tcp3" -- $C4
run tcp4 --pmc TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum --output-format csv -d "$OUT/tcp4" -- $C4
run utcl --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d "$OUT/utcl" -- $C4
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_c4mem"
for f in sorted(glob.glob(f"{out}/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[2], {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
