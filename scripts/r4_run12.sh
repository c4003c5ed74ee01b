O=gpurun_out/r4k; mkdir -p $O
for ord in 0 2 1 4 17 20; do
  TRG_EXP_OPTS="6=1,12=$ord" timeout -k 10 300 python scripts/exp_ab.py --one=c4xl:shipped 2>&1 | grep -v amdgpu.ids | sed "s/^/order $ord: /" | tee -a $O/c4xl_orders.log
done
