timeout 400 python -m pytest tests/test_sr_gpu.py -m gpu -q 2>&1 | tail -3
K4_B6_NW1=8 timeout 300 python scratch/sr_bench.py bf16x6 2>&1 | grep -v amdgpu.ids | tail -1
K4_B6_NW1=4 timeout 300 python scratch/sr_bench.py bf16x6 2>&1 | grep -v amdgpu.ids | tail -1
