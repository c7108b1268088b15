O=gpurun_out/r06_defer; mkdir -p $O
timeout 1200 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 600 python tools/joint_phase_events.py 2>/dev/null; done
BLOCKS=8 SHOW_BLOCKS=1 timeout 600 python tools/joint_step_time.py 2>/dev/null | tail -2
K4_TRAIN_TAPE=1 OUT=r06_defer/tape1 timeout 600 bash tools/joint_timeline_detail.sh > $O/detail_tape1.log 2>&1
head -30 $O/detail_tape1.log | grep -v "^W2026" | cut -c1-150
