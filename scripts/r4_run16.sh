O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_c3_and_c5" > $O/tests.log 2>&1; tail -3 $O/tests.log
bash scripts/ab.sh c4 shipped rgmin12 rgmin20 rgper2 rgper8 rgmw24 rgmw40 2>&1 | tail -14
