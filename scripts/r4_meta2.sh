set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/meta2
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_host_surface.py::test_committed_counters_belong_to_these_kernel_sources > gpurun_out/meta2/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/meta2/pytest.log
[ $rc -eq 0 ] || exit $rc
for cfg in c2 c3; do
  for rep in 1 2 3; do
    for v in shipped head_build; do
      timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:$v 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/meta2/ab.log
    done
  done
done
