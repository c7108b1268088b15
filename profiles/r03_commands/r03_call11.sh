#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_march_gpu.py -m gpu -x -q > $O/tests_march.log 2>&1; echo "march_tests_rc=$?"; tail -3 $O/tests_march.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --sr-frames 0"
for i in 1 2; do
  timeout 300 $B > $O/b_$i.json 2> $O/b_$i.err
  python - "$O/b_$i.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('value', d['value'], 'median_ms', d.get('ms_per_step_median'), 'isolated_ms', d['roofline']['kernel_ms'], 'mrays_iso', d['mrays_isolated'], 'frac', d['roofline']['frac'])
PY
done
K4_LIB=$R/4k-nerf_amd/lib4k_hip_timing.so python tools/shade_timing.py 2>&1 | grep -v Warning | tail -11
