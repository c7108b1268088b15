set -x
mkdir -p gpurun_out/r06_tape1
timeout 900 python -m pytest tests/test_sr_train_gpu.py -x -q -m gpu -k "tape or decoder" > gpurun_out/r06_tape1/tests_tape.log 2>&1; echo rc=$? >> gpurun_out/r06_tape1/tests_tape.log
tail -30 gpurun_out/r06_tape1/tests_tape.log
for t in 1 0 1 0; do
  echo "== K4_TRAIN_TAPE=$t" >> gpurun_out/r06_tape1/joint_time.log
  K4_TRAIN_TAPE=$t BLOCKS=6 SHOW_BLOCKS=1 timeout 600 python tools/joint_step_time.py >> gpurun_out/r06_tape1/joint_time.log 2>&1
done
cat gpurun_out/r06_tape1/joint_time.log | grep -v "^\[" | tail -20
