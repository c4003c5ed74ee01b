# round 4, final kernels: the bench lines (with the r04 counters committed), the whole-frame parity test, fuzz
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4final4; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/c5.err; echo "c5 rc $?"
python bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4xl.json 2> $O/c4xl.err; echo "c4xl rc $?"
TRG_BENCH_GPU_BUILD=1 python bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4xl_device_build.json 2> $O/c4xl_dev.err; echo "c4xl dev rc $?"
TRG_BENCH_GROUP=1 python bench.py --no-cpu-baseline > $O/bench_n1_through_group_path.json 2> $O/n1g.err; echo "n1 group rc $?"
TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 --no-cpu-baseline > $O/bench_g8_rehearsal_one_device.json 2> $O/g8.err; echo "g8 rc $?"
TRG_RUN_SLOW=1 timeout -k 10 600 python -m pytest tests/test_gpu_fullframe.py -q -m "gpu and slow" > $O/fullframe_parity_test.txt 2>&1; tail -3 $O/fullframe_parity_test.txt
timeout -k 10 200 python scripts/gpu_fuzz.py 400 201 > $O/fuzz_strict_seed201.txt 2>&1; tail -1 $O/fuzz_strict_seed201.txt
timeout -k 10 200 python scripts/gpu_fuzz.py 400 202 fast > $O/fuzz_shipped_seed202.txt 2>&1; tail -1 $O/fuzz_shipped_seed202.txt
