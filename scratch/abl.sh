timeout 600 python -m pytest tests/test_march_gpu.py -m gpu -x -q 2>&1 | tail -5
for cfg in "K4_GEOM_SPLIT=1" "K4_GEOM_SPLIT=0 K4_GEOM_WPB=1" "K4_GEOM_SPLIT=1 K4_DEBUG=128" "K4_GEOM_SPLIT=1 K4_DEBUG=128 K4_GEOM_LDSPAD=6000" "K4_GEOM_SPLIT=1 K4_DEBUG=128 K4_GEOM_LDSPAD=12000"; do
  echo "== $cfg"
  env $cfg timeout 200 python bench.py --steps 20 --sr-frames 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',r['value'],'iso_ms',r['roofline']['kernel_ms'],'ovl',r['ms_per_step'])"
done
