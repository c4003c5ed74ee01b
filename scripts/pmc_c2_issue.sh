#!/bin/bash
# what keeps C2's render kernel from issuing?  instruction-cache and issue-stall counters of one pipelined bench run (run on the GPU box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_c2_issue; rm -rf "$OUT"; mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -3 "$OUT/$name.log"; }; }
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES
run dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES
run stall SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_kernel<true, false>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print("%-24s median %.5g  (n %d)" % (k, sorted(agg[k])[len(agg[k]) // 2], len(agg[k])))
PY
