cd /root/repo
mkdir -p gpurun_out/r4c47
timeout 1800 python -m pytest tests/test_train_gpu.py tests/test_march_gpu.py tests/test_train_ops_gpu.py -x -q 2>&1 | tail -2 | cut -c1-200 | tee gpurun_out/r4c47/tests.log
for f in 1 0 1 0; do echo "K4_TRAIN_PREFILTER=$f"; K4_TRAIN_PREFILTER=$f ITERS=12 timeout 300 python tools/joint_step_time.py 2>/dev/null | grep "joint iteration"; done | tee gpurun_out/r4c47/joint.log
