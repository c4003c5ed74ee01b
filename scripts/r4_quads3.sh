set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quads
timeout -k 10 400 python scripts/gpu_fuzz.py 500 42 fast > gpurun_out/quads/fuzz_fast.log 2>&1; rc=$?; tail -2 gpurun_out/quads/fuzz_fast.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/quads/bench_c2.json 2> gpurun_out/quads/bench_c2.err; rc=$?; cat gpurun_out/quads/bench_c2.json
exit $rc
