#!/bin/bash
# decoder A/B on the GPU box: ms per 4K frame (tools/sr_frame_time.py) per env configuration (arguments), then the SR / e2e tests
mkdir -p gpurun_out/r2d
for cfg in "$@"; do
  echo "== $cfg: $(env $cfg python tools/sr_frame_time.py bf16x6 2>&1 | tail -1)"
done
if [ -n "$RUN_TESTS" ]; then
python -m pytest $RUN_TESTS -m gpu -q > gpurun_out/r2d/tests.log 2>&1; echo tests_rc=$?
tail -12 gpurun_out/r2d/tests.log
fi
