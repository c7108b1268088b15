#!/bin/bash
# shading kernel: rgbnet on 2-term fp16 splits, second form (fma_mix splits with the tile scale folded in, maxima on bit patterns, one-instruction
# ReLU, row factors pushed through the ReLUs into the next layer's packed weights) against the bf16x6 form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_march_gpu.py tests/test_e2e_gpu.py -m gpu -q -x -s 2>&1 | grep -E "rgbnet arithmetics|passed|failed|Error|assert" | tail -8
for m in "" bf16x6 "" bf16x6; do echo "== K4_MLP=$m"; K4_MLP=$m timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --sr-frames 0 2>&1 | grep "\"metric\"" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"mrays_isolated\"], d[\"roofline\"][\"kernel_ms\"], d[\"roofline\"][\"frac\"])"; done
