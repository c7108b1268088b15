set -x
O=gpurun_out/r06_tape4; mkdir -p $O
timeout 1200 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log
tail -8 $O/tests.log
for v in "K4_SR_DEBUG=0" "K4_SR_DEBUG=2048" "K4_SR_DEBUG=0" "K4_SR_DEBUG=2048"; do
  echo "== $v" >> $O/phases.log
  env $v timeout 600 python tools/joint_phase_events.py >> $O/phases.log 2>/dev/null
done
cat $O/phases.log
python tools/train_kernels_time.py > $O/kernels_small.log 2>&1
K4_SR_DEBUG=2048 python tools/train_kernels_time.py > $O/kernels_row.log 2>&1
cat $O/kernels_small.log $O/kernels_row.log
