cd /root/repo
mkdir -p gpurun_out/r4c29
for v in 1 0 1 0; do echo "K4_TRAIN_WGRAD_STREAM=$v"; K4_TRAIN_WGRAD_STREAM=$v ITERS=12 timeout 300 python tools/joint_step_time.py 2>/dev/null | grep "joint iteration"; done | tee gpurun_out/r4c29/joint.log
timeout 1500 python -m pytest tests/test_sr_train_gpu.py tests/test_train_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/r4c29/tests.log
