#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box through gpurun):  bash scripts/profile_round.sh [c2] [c4]
# Kernel-trace/stats passes and PMC passes are separate runs (gpurun refuses mixed ones); FETCH_SIZE and WRITE_SIZE need
# separate passes (TCC slot limit).  Output: gpurun_out/profiles/{summary.json, c2_counters.json, c4_counters.json,
# *_kernel_stats.csv}; copy what should be judged into profiles/rNN/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
# the profiler's preloaded library starts the HIP runtime before python runs bench.py's os.environ.setdefault: export the
# queue count here so that the profiled overlap is the benched one
export GPU_MAX_HW_QUEUES=16
OUT=$PWD/gpurun_out/profiles; mkdir -p "$OUT"
WHAT="${*:-c2 c4}"
C2="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"
C4="python3 bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline"
C3="python3 bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline"
C5="python3 bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline"
C4XL="python3 bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; rm -rf "$OUT/$name"; rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name"; tail -5 "$OUT/$name.log"; }; }
for cfg in $WHAT; do
  if [ "$cfg" = c2 ]; then CMD=$C2; elif [ "$cfg" = c3 ]; then CMD=$C3; elif [ "$cfg" = c5 ]; then CMD=$C5; elif [ "$cfg" = c4xl ]; then CMD=$C4XL; else CMD=$C4; fi
  run ${cfg}_trace --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_trace" -- $CMD
  run ${cfg}_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/${cfg}_fetch" -- $CMD
  run ${cfg}_write --pmc WRITE_SIZE --output-format csv -d "$OUT/${cfg}_write" -- $CMD
  run ${cfg}_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/${cfg}_sq1" -- $CMD
  run ${cfg}_sq2 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${cfg}_sq2" -- $CMD
  run ${cfg}_dram --pmc TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/${cfg}_dram" -- $CMD
  if [ "$cfg" = c4 ] || [ "$cfg" = c4xl ]; then
    run ${cfg}_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/${cfg}_tcc" -- $CMD
  fi
  if [ "$cfg" = c4 ]; then
    run c4_ta --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d "$OUT/c4_ta" -- $CMD
  fi
done
python3 scripts/pmc_summary.py "$OUT" $WHAT
