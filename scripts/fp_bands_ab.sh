# frame-parallel kernel variants (run on the GPU box): small HBM scenes in bands, and C2 bands (LDS scene) per occupancy
cd "$GRAFT_REPO_ROOT"
for v in shipped fph7; do
  for n in 6 12; do
    if [ $v = shipped ]; then unset TRG_HIP_SO; else export TRG_HIP_SO=$PWD/exp_build/$v/libtoyraygun_hip.so; fi
    echo "== $v lattice $n"; python scripts/gpu_c4_bands.py $n 2>&1 | grep "fsplit4 lock\|fsplit2 lock"
  done
done
for v in shipped fpw7 fpw8; do
  if [ $v = shipped ]; then unset TRG_HIP_SO; else export TRG_HIP_SO=$PWD/exp_build/$v/libtoyraygun_hip.so; fi
  echo "== $v C2 bands"; python scripts/gpu_fsplit.py 2>&1 | grep "fsplit=[24]"
done
