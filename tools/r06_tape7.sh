set -x
O=gpurun_out/r06_tape7; mkdir -p $O
timeout 1200 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -x -q > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -4 $O/tests.log
for v in "K4_SR_DEBUG=0" "K4_SR_DEBUG=0"; do
  echo "== $v" >> $O/phases.log
  env $v timeout 600 python tools/joint_phase_events.py >> $O/phases.log 2>/dev/null
done
cat $O/phases.log
BLOCKS=8 SHOW_BLOCKS=1 timeout 600 python tools/joint_step_time.py 2>/dev/null | tail -3
python tools/train_kernels_time.py 2>/dev/null > $O/kernels_small.log
cat $O/kernels_small.log
