# A/B of experimental builds on the GPU box:  bash scripts/ab.sh <config> <variant> [<variant> ...]   (variants of scripts/exp_build.sh, or "shipped")
set -o pipefail
cd "$GRAFT_REPO_ROOT"
cfg=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in "$@"; do
    timeout -k 10 180 python scripts/exp_ab.py --one=$cfg:$v 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/$cfg.log
  done
done
