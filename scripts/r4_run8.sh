O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -12 $O/tests.log
bash scripts/ab.sh c4 shipped planeshbm0 2>&1 | tail -4
python scripts/gpu_fullframe.py c4 2>&1 | grep -v amdgpu
