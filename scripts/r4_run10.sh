O=gpurun_out/r4i; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
for rep in 1 2; do
python scripts/regen_crossover.py 0 6 12 20 2>&1 | grep -v amdgpu | sed 's/^/planes:   /'
TRG_VARIANT=noplanes python scripts/regen_crossover.py 0 6 12 20 2>&1 | grep -v amdgpu | sed 's/^/noplanes: /'
done
bash scripts/ab.sh c4 shipped noplanes 2>&1 | tail -4
