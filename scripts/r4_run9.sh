O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -12 $O/tests.log
for cfg in c2 c3 c4; do bash scripts/ab.sh $cfg shipped noplanes 2>&1 | tail -4; done
python scripts/gpu_tri_test_accuracy.py 2>&1 | grep -v amdgpu
