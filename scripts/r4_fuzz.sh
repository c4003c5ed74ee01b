O=gpurun_out/r4fuzz; mkdir -p $O
timeout -k 10 500 python scripts/gpu_fuzz.py 1500 101 > $O/fuzz_strict_101.log 2>&1; tail -2 $O/fuzz_strict_101.log
timeout -k 10 400 python scripts/gpu_fuzz.py 1000 102 fast > $O/fuzz_fast_102.log 2>&1; tail -2 $O/fuzz_fast_102.log
grep -c MISMATCH $O/*.log
