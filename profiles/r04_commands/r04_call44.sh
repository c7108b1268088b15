cd /root/repo
mkdir -p gpurun_out/r4c44
for cfg in "1 1" "1 0" "0 0" "1 1" "1 0" "0 0"; do set -- $cfg; echo "K4_TRAIN_NATIVE_RDB=$1 K4_TRAIN_WGRAD_STREAM=$2"; K4_TRAIN_NATIVE_RDB=$1 K4_TRAIN_WGRAD_STREAM=$2 ITERS=12 timeout 300 python tools/joint_step_time.py 2>/dev/null | grep "joint iteration"; done | tee gpurun_out/r4c44/joint.log
timeout 1500 python -m pytest tests/test_sr_train_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -v "Warning\|warn" | tail -3 | tee gpurun_out/r4c44/tests.log
