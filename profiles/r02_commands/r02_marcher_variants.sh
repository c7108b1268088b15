#!/bin/bash
# compile-time variants of the marcher kernels, rebuilt and timed on the GPU box in ONE call (arguments: hipcc -D flag sets; each
# is measured REPS times, interleaved, because call-to-call differences between boxes exceed the effects being measured)
mkdir -p gpurun_out/r2b
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --sr-frames 0"
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  touch 4k-nerf_amd/csrc/k4_march.hip
  K4_EXTRA_HIPCC_FLAGS="$v" python 4k-nerf_amd/build.py > /dev/null 2>&1 || echo "build failed: $v"
  $B > gpurun_out/r2b/variant.json 2> gpurun_out/r2b/variant.err
  python - "$v" <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/r2b/variant.json'))
    print('%-40s' % sys.argv[1], 'value', d['value'], 'isolated_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
done
touch 4k-nerf_amd/csrc/k4_march.hip
