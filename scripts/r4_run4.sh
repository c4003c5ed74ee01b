O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "tail_compaction or tile_order_never" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export TMPDIR=/tmp
for mode in 0 1 2 3; do
  TRG_EXP_OPTS="14=$mode" timeout -k 10 200 python scripts/exp_ab.py --one=c3:shipped 2>&1 | grep -v amdgpu.ids | sed "s/^/sort $mode: /" | tee -a $O/c3_sort.log
done
# per-kernel times: kernel trace of the C3 workload (64 spp) per mode
for mode in 0 1 3; do
  TRG_EXP_OPTS="14=$mode" rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -- python3 scripts/exp_ab.py --one=c3:shipped > $O/trace_$mode.log 2>&1
  f=$(ls $O/trace_$mode/*/*_kernel_stats.csv | head -1); echo "== mode $mode"; cut -d, -f1-4,8 "$f" | head -6
done
