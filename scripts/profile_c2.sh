#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box through gpurun).
# Pass 1: kernel trace + stats; passes 2/3: HBM-side PMC counters (FETCH_SIZE and WRITE_SIZE need separate passes).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1 || { tail -20 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1 || { tail -20 "$OUT/pmc_write.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- $CMD > "$OUT/pmc_sq.log" 2>&1 || { tail -20 "$OUT/pmc_sq.log"; }
find "$OUT" -name "*.csv" | head -30
