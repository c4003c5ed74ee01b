python scripts/gpu_tri_test_accuracy.py 2>&1 | grep -v amdgpu
python scripts/gpu_tri_test_accuracy.py planes0 2>&1 | grep -v amdgpu
python scripts/gpu_fullframe.py c2 c5 2>&1 | grep -v amdgpu
