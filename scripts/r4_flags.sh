# round 4: the build's new scheduler flags (max-ilp everywhere; the regeneration kernels in a unit of their own without the post-RA scheduler) -- GPU suite, then A/B against the old flags
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/flags; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_host_surface.py::test_committed_counters_belong_to_these_kernel_sources > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2 3; do
  for cfg in c2 c3 c4 c4xl; do
    for v in shipped head_build; do
      timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:$v 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
    done
  done
done
