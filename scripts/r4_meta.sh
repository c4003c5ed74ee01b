# round 4: index / mask of the plane-form leaf records in the record's FIRST 64 bytes -- HBM parity tests, A/B against the old layout, memory-side counters
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/meta
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "regeneration or device_build or leaf_records or c4 or lattice or texture or soup or gpu_build or ties" > gpurun_out/meta/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/meta/pytest.log
[ $rc -eq 0 ] || exit $rc
for cfg in c4 c4xl; do
  for rep in 1 2; do
    for v in shipped meta_last; do
      timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:$v 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/meta/ab.log
    done
  done
done
bash scripts/pmc_c4.sh meta_first > gpurun_out/meta/pmc_first.txt 2>&1; tail -25 gpurun_out/meta/pmc_first.txt
