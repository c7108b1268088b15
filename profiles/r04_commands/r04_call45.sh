cd /root/repo
mkdir -p gpurun_out/r4c45
for cfg in "1 1" "0 0" "1 1" "0 0"; do set -- $cfg; echo "K4_TRAIN_NATIVE_RDB=$1 K4_TRAIN_WGRAD_STREAM=$2 (+ MaskedAdam fast path in both)"; K4_TRAIN_NATIVE_RDB=$1 K4_TRAIN_WGRAD_STREAM=$2 ITERS=12 timeout 300 python tools/joint_step_time.py 2>/dev/null | grep "joint iteration"; done | tee gpurun_out/r4c45/joint.log
timeout 1800 python -m pytest tests/test_optim_gpu.py tests/test_sr_train_gpu.py tests/test_train_gpu.py tests/test_train_ops_gpu.py -x -q 2>&1 | tail -2 | cut -c1-200 | tee gpurun_out/r4c45/tests.log
