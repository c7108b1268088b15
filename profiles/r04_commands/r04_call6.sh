cd /root/repo
mkdir -p gpurun_out/r4c6
timeout 300 python tests/debug/p16_frame_diff.py rand > gpurun_out/r4c6/diff_rand.log 2>&1; grep -v amdgpu gpurun_out/r4c6/diff_rand.log | head -14
timeout 300 python tools/sr_frame_time.py f16x3 f16x3p f16x3 f16x3p > gpurun_out/r4c6/frame.log 2>&1; grep -v amdgpu gpurun_out/r4c6/frame.log
K4_SR_STREAMS=2 timeout 300 python tools/sr_frame_time.py f16x3 f16x3p f16x3p > gpurun_out/r4c6/frame_s2.log 2>&1; grep -v amdgpu gpurun_out/r4c6/frame_s2.log
K4_SR_STREAMS=4 timeout 300 python tools/sr_frame_time.py f16x3p f16x3p > gpurun_out/r4c6/frame_s4.log 2>&1; grep -v amdgpu gpurun_out/r4c6/frame_s4.log
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r4c6/gpu_tests.log 2>&1; tail -12 gpurun_out/r4c6/gpu_tests.log
