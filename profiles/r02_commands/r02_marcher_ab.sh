#!/bin/bash
# A/B of geometry-kernel knobs on the GPU box: isolated call time + pipelined rate per env configuration (arguments), then tests
mkdir -p gpurun_out/r2b
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --sr-frames 0"
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg $B > gpurun_out/r2b/$tag.json 2> gpurun_out/r2b/$tag.err
  python - "$tag" <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/r2b/%s.json' % sys.argv[1]))
    print(sys.argv[1], 'value', d['value'], 'isolated_ms', d['roofline']['kernel_ms'], 'mrays_iso', d['mrays_isolated'], 'frac', d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
if [ -n "$RUN_TESTS" ]; then
python -m pytest $RUN_TESTS -m gpu -q > gpurun_out/r2b/tests.log 2>&1; echo tests_rc=$?
tail -8 gpurun_out/r2b/tests.log
fi
