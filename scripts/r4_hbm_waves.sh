set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/hbmw; mkdir -p $O
for rep in 1 2; do
  for v in shipped hbmw8 hbmw6; do
    timeout -k 10 240 python scripts/exp_ab.py --one=c4:$v 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
  done
done
for v in shipped hbmw8 hbmw6; do
  timeout -k 10 240 python scripts/exp_ab.py --one=c4xl:$v 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
