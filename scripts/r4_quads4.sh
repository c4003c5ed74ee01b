# round 4: quad leaves from the device builders too -- GPU suite, fuzz, then C4 / c4xl with the host and the device SAH builder, quads on and off
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quads
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_host_surface.py::test_committed_counters_belong_to_these_kernel_sources > gpurun_out/quads/pytest4.log 2>&1; rc=$?
tail -3 gpurun_out/quads/pytest4.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/gpu_fuzz.py 400 43 > gpurun_out/quads/fuzz_strict4.log 2>&1; rc=$?; tail -2 gpurun_out/quads/fuzz_strict4.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/gpu_fuzz.py 400 44 fast > gpurun_out/quads/fuzz_fast4.log 2>&1; rc=$?; tail -2 gpurun_out/quads/fuzz_fast4.log
[ $rc -eq 0 ] || exit $rc
for cfg in c4 c4xl; do
  for gb in 0 1; do
    for q in 1 0; do
      echo "$cfg gpu_build=$gb quads=$q" | tee -a gpurun_out/quads/ab_devbuild.log
      TRG_BVH_QUADS=$q TRG_EXP_OPTS="6=$gb" timeout -k 10 240 python scripts/exp_ab.py --one=$cfg:shipped 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/quads/ab_devbuild.log
    done
  done
done
