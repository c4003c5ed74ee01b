#!/bin/bash
# issue / stall breakdown of the C2 megakernel (SQ counters, one rocprofv3 --pmc pass per group)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_c2sq; rm -rf "$OUT"; mkdir -p "$OUT"
C2="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline"
run() { name=$1; shift; echo "pass $name"; timeout -k 10 200 rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; grep -m2 -i "error code\|exceeds" "$OUT/$name.log"; }; }
run g1 --pmc SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/g1" -- $C2
run g2 --pmc SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_IFETCH SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d "$OUT/g2" -- $C2
run g3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d "$OUT/g3" -- $C2
python3 - <<'PY'
import csv, glob, collections
out = "gpurun_out/prof_c2sq"
for f in sorted(glob.glob(f"{out}/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel<true, false>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[2], {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
