# round 4 experiment (scratch sources, exp_build variant "dense"): the quad X test records of an all-quad scene packed at 64 bytes -- what would a dense leaf-test array buy?
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dense; mkdir -p $O
TRG_HIP_SO=$PWD/exp_build/dense/libtoyraygun_hip.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "c4_million or full_size_c4 or regeneration_is_bit" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in c4 c4xl; do
  for rep in 1 2; do
    for v in shipped dense; do
      timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:$v 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
    done
  done
done
