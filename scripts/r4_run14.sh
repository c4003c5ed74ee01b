O=gpurun_out/r4m; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "tile_order_never or path_regeneration" > $O/tests.log 2>&1; tail -8 $O/tests.log
for ord in 0 66 65 72; do
  TRG_EXP_OPTS="12=$ord" timeout -k 10 200 python scripts/exp_ab.py --one=c4:shipped 2>&1 | grep -v amdgpu.ids | sed "s/^/order $ord: /"
done
for ord in 0 66; do
  TRG_EXP_OPTS="6=1,12=$ord" timeout -k 10 300 python scripts/exp_ab.py --one=c4xl:shipped 2>&1 | grep -v amdgpu.ids | sed "s/^/order $ord: /"
done
for ord in 66; do
  timeout -k 10 600 bash scripts/pmc_c4.sh order$ord "12=$ord" > $O/pmc_order$ord.txt 2>&1
  grep -E "read_GB|l2_hit|wait_share|GRBM_GUI|l1_to_l2|SQ_INSTS_VALU|SQ_WAVE_CYC" $O/pmc_order$ord.txt | sed "s/^/order $ord: /"
done
