O=gpurun_out/r4n; mkdir -p $O
bash scripts/ab.sh c3 shipped tailk1 tailk3 tails1 tails3 2>&1 | tail -10
python scripts/gpu_leaf.py 2>&1 | grep -v amdgpu | tee $O/leaf_c2.txt
python scripts/gpu_leaf.py c4 2>&1 | grep -v amdgpu | tee $O/leaf_c4.txt
