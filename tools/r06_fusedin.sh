timeout 1200 python -m pytest tests/test_train_ops_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -4
for i in 1 2; do timeout 600 python tools/joint_phase_events.py 2>/dev/null; done
BLOCKS=8 SHOW_BLOCKS=1 timeout 600 python tools/joint_step_time.py 2>/dev/null | tail -2
