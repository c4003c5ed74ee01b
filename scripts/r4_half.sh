# round 4 experiment: half-precision sign-ordered LDS nodes (TRG_TRAV_LDS=7, exp_build variant half7) -- parity of the variant, then A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/half7; mkdir -p $O
TRG_HIP_SO=$PWD/exp_build/half7/libtoyraygun_hip.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "not layout and not plugin and not reference_app and not async and not host" > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
for cfg in c2 c3; do
  for rep in 1 2 3; do
    for v in shipped half7; do
      timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:$v 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
    done
  done
done
