O=gpurun_out/r4c; mkdir -p $O
for ord in 0 17 18 20 24 2; do
  TRG_EXP_OPTS="12=$ord" timeout -k 10 200 python scripts/exp_ab.py --one=c4:shipped 2>&1 | grep -v amdgpu.ids | sed "s/^/order $ord: /" | tee -a $O/c4_orders.log
done
for ord in 0 17 18 20; do
  timeout -k 10 600 bash scripts/pmc_c4.sh order$ord "12=$ord" > $O/pmc_order$ord.txt 2>&1
  grep -E "read_GB|l2_hit|wait_share|lanes_per|SQ_INSTS_VALU|SQ_WAVE_CYCLES|GRBM_GUI|l1_to_l2" $O/pmc_order$ord.txt | sed "s/^/order $ord: /"
done
echo done
