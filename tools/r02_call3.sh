#!/bin/bash
O=gpurun_out/r2x
mkdir -p $O
python -m pytest tests/test_train_ops_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "tests_rc=$?"; tail -3 $O/tests.log
python tools/train_step_time.py 2>&1 | tail -1
K4_RGBNET=torch python tools/train_step_time.py 2>&1 | tail -1
for s in 2 3 4 6; do
  python bench.py --steps 80 --warmup 8 --no-cpu-baseline --no-extras --sr-frames 0 --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams', $s, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'iso', d['mrays_isolated'])"
done
