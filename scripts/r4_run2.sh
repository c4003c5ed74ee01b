set -x
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python bench.py --no-cpu-baseline > $O/n1.json 2> $O/n1.err &&
TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 > $O/g8_il.json 2> $O/g8_il.err &&
TRG_BANDS=contiguous TRG_GROUP_EXCHANGE=copy TRG_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --gpus 8 > $O/g8_ct.json 2> $O/g8_ct.err &&
python scripts/gpu_band_balance.py c2 2 4 8 > $O/balance_c2.jsonl 2> $O/balance_c2.err &&
python scripts/gpu_band_balance.py c4 2 8 > $O/balance_c4.jsonl 2> $O/balance_c4.err &&
python scripts/gpu_band_balance.py c5 8 > $O/balance_c5.jsonl 2> $O/balance_c5.err
echo done $?
