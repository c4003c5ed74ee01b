set -x
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py -x -q -k "group or pipelined or bench" > gpurun_out/r4a/tests.log 2>&1 || { tail -40 gpurun_out/r4a/tests.log; exit 1; }
tail -3 gpurun_out/r4a/tests.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r4a/n1.json 2> gpurun_out/r4a/n1.err &&
TRG_BENCH_GROUP=1 python bench.py --no-cpu-baseline > gpurun_out/r4a/n1_group.json 2> gpurun_out/r4a/n1_group.err &&
TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 > gpurun_out/r4a/g8.json 2> gpurun_out/r4a/g8.err &&
TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0 python bench.py --gpus 2 > gpurun_out/r4a/g2.json 2> gpurun_out/r4a/g2.err &&
python scripts/gpu_band_balance.py c2 2 4 8 > gpurun_out/r4a/balance_c2.jsonl 2> gpurun_out/r4a/balance_c2.err &&
python scripts/gpu_band_balance.py c4 8 > gpurun_out/r4a/balance_c4.jsonl 2> gpurun_out/r4a/balance_c4.err
echo done $?
