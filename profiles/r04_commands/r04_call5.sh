cd /root/repo
mkdir -p gpurun_out/r4c5
timeout 300 python tests/debug/p16_frame_diff.py rand > gpurun_out/r4c5/diff_rand.log 2>&1; grep -v amdgpu gpurun_out/r4c5/diff_rand.log
timeout 300 python tests/debug/p16_frame_diff.py smooth > gpurun_out/r4c5/diff_smooth.log 2>&1; grep -v amdgpu gpurun_out/r4c5/diff_smooth.log
timeout 900 python -m pytest tests/test_sr_gpu.py tests/test_march_gpu.py -q -k "p16 or f16x3p or golden_fused" > gpurun_out/r4c5/tests.log 2>&1; tail -15 gpurun_out/r4c5/tests.log
