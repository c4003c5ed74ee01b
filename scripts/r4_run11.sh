O=gpurun_out/r4j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
python bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4xl_host.json 2> $O/c4xl_host.err; echo "c4xl host rc $?"
TRG_BENCH_GPU_BUILD=1 python bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4xl_dev1.json 2> $O/c4xl_dev1.err; echo "c4xl dev1 rc $?"
python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/c5.err; echo "c5 rc $?"
