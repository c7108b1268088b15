set -x
mkdir -p gpurun_out/r06_tape2
timeout 900 python -m pytest tests/test_sr_train_gpu.py -x -q -m gpu -k "tape" > gpurun_out/r06_tape2/tests_tape.log 2>&1; echo rc=$? >> gpurun_out/r06_tape2/tests_tape.log
tail -5 gpurun_out/r06_tape2/tests_tape.log
K4_TRAIN_TAPE=1 OUT=r06_tape2/tape1 timeout 600 bash tools/joint_timeline_detail.sh > gpurun_out/r06_tape2/detail_tape1.log 2>&1
K4_TRAIN_TAPE=0 OUT=r06_tape2/tape0 timeout 600 bash tools/joint_timeline_detail.sh > gpurun_out/r06_tape2/detail_tape0.log 2>&1
cat gpurun_out/r06_tape2/detail_tape1.log
