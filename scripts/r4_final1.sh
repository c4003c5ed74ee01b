O=gpurun_out/r4final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; tail -4 $O/tests_gpu.log
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 > $O/bench_g8_rehearsal_one_device.json 2> $O/g8.err; echo "g8 rc $?"
TRG_BENCH_GROUP=1 python bench.py --no-cpu-baseline > $O/bench_n1_through_group_path.json 2> $O/n1g.err; echo "n1 group rc $?"
python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/c5.err; echo "c5 rc $?"
python bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4xl.json 2> $O/c4xl.err; echo "c4xl rc $?"
TRG_BENCH_GPU_BUILD=1 python bench.py --config c4xl --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4xl_device_build.json 2> $O/c4xl_dev.err; echo "c4xl dev rc $?"
TRG_COMMIT=$1 bash scripts/profile_round.sh c2 c4 c5 c3 c4xl > $O/profile_round.log 2>&1; tail -3 $O/profile_round.log
python scripts/kernel_resources.py > $O/kernel_resources.txt 2>&1
