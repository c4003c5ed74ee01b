set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4fuzz; mkdir -p $O
timeout -k 10 560 python scripts/gpu_fuzz.py 2500 301 > $O/fuzz_strict_seed301.txt 2>&1; tail -1 $O/fuzz_strict_seed301.txt
timeout -k 10 560 python scripts/gpu_fuzz.py 2500 302 fast > $O/fuzz_shipped_seed302.txt 2>&1; tail -1 $O/fuzz_shipped_seed302.txt
