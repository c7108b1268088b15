#!/bin/bash
# Round 3, GPU call 3: f16x3 as the default decoder arithmetic (3 workgroups per CU, DPP reduction), shading kernel at 2 / 3 / 4 waves per SIMD.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_march_gpu.py -m gpu -x -q -k "live_mask" > $O/tests_live.log 2>&1; echo "live_tests_rc=$?"; tail -3 $O/tests_live.log
timeout 900 python -m pytest tests/test_sr_gpu.py -m gpu -q -s > $O/tests_sr.log 2>&1; echo "sr_tests_rc=$?"
grep -E "PSNR|dB|max err|passed|failed|Error|assert" $O/tests_sr.log | head -30
python tools/sr_frame_time.py f16x3 bf16x6 bf16x3 f16x3 2>&1 | grep ms/frame
for r in 3 4; do echo "K4_SR_2T_RPW=$r"; K4_SR_2T_RPW=$r python tools/sr_frame_time.py f16x3 2>&1 | grep ms/frame; done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --sr-frames 0"
for wps in 2 3 4 2 3 4; do
  K4_SHADE_WPS=$wps timeout 300 $B > $O/wps_$wps.json 2> $O/wps_$wps.err
  python - "$O/wps_$wps.json" "K4_SHADE_WPS=$wps" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], 'value', d['value'], 'median_ms', d.get('ms_per_step_median'), 'isolated_ms', d['roofline']['kernel_ms'], 'mrays_iso', d['mrays_isolated'], 'frac', d['roofline']['frac'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
for wps in 3 4; do
  K4_SHADE_WPS=$wps timeout 600 python -m pytest tests/test_march_gpu.py -m gpu -x -q > $O/tests_march_wps$wps.log 2>&1; echo "wps$wps march_tests_rc=$?"; tail -2 $O/tests_march_wps$wps.log
done
