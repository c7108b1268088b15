cd /root/repo
mkdir -p gpurun_out/r4c13
timeout 600 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -x -q > gpurun_out/r4c13/tests.log 2>&1; tail -3 gpurun_out/r4c13/tests.log
timeout 300 python tools/joint_step_time.py 2>&1 | grep -v amdgpu | tail -6 | tee gpurun_out/r4c13/joint.log
