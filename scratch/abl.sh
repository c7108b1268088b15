for cfg in "K4_MLP=b3" "K4_MLP=fp32" "K4_MLP=b3 K4_DEBUG=3"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --steps 20 --sr-frames 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',r['value'],'iso_ms',r['roofline']['kernel_ms'],'ovl',r['ms_per_step'])"
done
