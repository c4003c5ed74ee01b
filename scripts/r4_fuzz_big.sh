set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4fuzz; mkdir -p $O
timeout -k 10 560 python scripts/gpu_fuzz.py 2500 401 > $O/fuzz_strict_seed401.txt 2>&1; tail -1 $O/fuzz_strict_seed401.txt
timeout -k 10 560 python scripts/gpu_fuzz.py 2500 402 fast > $O/fuzz_shipped_seed402.txt 2>&1; tail -1 $O/fuzz_shipped_seed402.txt
grep -B1 MISMATCH $O/fuzz_shipped_seed402.txt | cut -c1-200
