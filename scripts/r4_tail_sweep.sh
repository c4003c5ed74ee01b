# round 4, after the quad leaves: the tail-compaction knobs once more (K = first compacted bounce, re-compaction step) on C3, and K on C2
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quads
L=gpurun_out/quads/tail_sweep.log
for k in -1 1 2 3; do
  echo "c3 K=$k" | tee -a $L
  TRG_EXP_OPTS="9=$k" timeout -k 10 200 python scripts/exp_ab.py --one=c3:shipped 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for v in tstep1 tstep3; do
  echo "c3 $v" | tee -a $L
  timeout -k 10 200 python scripts/exp_ab.py --one=c3:$v 2>&1 | grep -v amdgpu.ids | tee -a $L
done
for k in -1 1 2; do
  echo "c2 K=$k" | tee -a $L
  TRG_EXP_OPTS="9=$k" timeout -k 10 200 python scripts/exp_ab.py --one=c2:shipped 2>&1 | grep -v amdgpu.ids | tee -a $L
done
