O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
for cfg in c2 c3 c4; do
  timeout -k 10 200 python scripts/exp_ab.py --one=$cfg:shipped 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
done
