# round 4: quad leaves -- the GPU parity suite, then A/B against TRG_BVH_QUADS=0 (same library, the builder switch)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quads
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_host_surface.py::test_committed_counters_belong_to_these_kernel_sources > gpurun_out/quads/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/quads/pytest.log
[ $rc -eq 0 ] || exit $rc
for cfg in c2 c3 c4 c4xl; do
  for rep in 1 2; do
    for q in 1 0; do
      echo "quads=$q" | tee -a gpurun_out/quads/ab.log
      TRG_BVH_QUADS=$q timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:shipped 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/quads/ab.log
    done
  done
done
