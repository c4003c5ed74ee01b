# round 4, after the quad leaves: does the SAH's traversal-cost constant or the job-queue tile order move on C4 / C2?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quads
for tc in 1.2 0.8 1.0 1.6 2.2; do
  echo "travcost=$tc" | tee -a gpurun_out/quads/sweep.log
  TRG_BVH_TRAVCOST=$tc timeout -k 10 240 python scripts/exp_ab.py --one=c4:shipped 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/quads/sweep.log
done
for to in 0 66; do
  echo "tile_order=$to" | tee -a gpurun_out/quads/sweep.log
  TRG_EXP_OPTS="12=$to" timeout -k 10 240 python scripts/exp_ab.py --one=c4:shipped 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/quads/sweep.log
done
for tc in 1.2 0.8 1.6 2.2; do
  echo "c2 travcost=$tc" | tee -a gpurun_out/quads/sweep.log
  TRG_BVH_TRAVCOST=$tc timeout -k 10 240 python scripts/exp_ab.py --one=c2:shipped 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/quads/sweep.log
done
