# round 4: quad leaves -- GPU suite, fuzz with parallelograms in the soups (strict bit-exact, shipped within tolerance), bench lines
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quads
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_host_surface.py::test_committed_counters_belong_to_these_kernel_sources > gpurun_out/quads/pytest2.log 2>&1; rc=$?
tail -3 gpurun_out/quads/pytest2.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/gpu_fuzz.py 500 41 > gpurun_out/quads/fuzz_strict.log 2>&1; rc=$?; tail -2 gpurun_out/quads/fuzz_strict.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/gpu_fuzz.py 500 42 fast > gpurun_out/quads/fuzz_fast.log 2>&1; rc=$?; tail -2 gpurun_out/quads/fuzz_fast.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/quads/bench_c2.json 2> gpurun_out/quads/bench_c2.err; rc=$?; cat gpurun_out/quads/bench_c2.json
exit $rc
