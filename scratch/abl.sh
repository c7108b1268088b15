timeout 600 python -m pytest tests/test_march_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -4
for cfg in "K4_X=1" "K4_NO_SKIP=1" "K4_X=1 K4_DEBUG=128" "K4_NO_SKIP=1 K4_DEBUG=128"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --steps 20 --sr-frames 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',r['value'],'iso_ms',r['roofline']['kernel_ms'],'ovl',r['ms_per_step'])"
done
