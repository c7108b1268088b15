set -x
O=gpurun_out/r06_tape3; mkdir -p $O
timeout 900 python -m pytest tests/test_sr_train_gpu.py -x -q -m gpu -k "tape" > $O/tests_tape.log 2>&1; echo rc=$? >> $O/tests_tape.log
tail -3 $O/tests_tape.log
for v in "K4_TRAIN_TAPE=1 SIDE_PRIO=1" "K4_TRAIN_TAPE=1 SIDE_PRIO=0" "K4_TRAIN_TAPE=0 SIDE_PRIO=1" "K4_TRAIN_TAPE=0 SIDE_PRIO=0" "K4_TRAIN_TAPE=1 SIDE_PRIO=1"; do
  echo "== $v" >> $O/phases.log
  env $v timeout 600 python tools/joint_phase_events.py >> $O/phases.log 2>/dev/null
done
cat $O/phases.log
