#!/bin/bash
# Round 3, GPU call 5: weights through LDS in the fp16 3x3 convolution (K4_SR_WLDS x K4_SR_NBK A/B, tests, per-layer table).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_sr_gpu.py -m gpu -q -s > $O/tests_sr.log 2>&1; echo "sr_tests_rc=$?"
grep -E "PSNR|dB|max err|passed|failed|Error|assert" $O/tests_sr.log | head -30
for cfg in "K4_SR_WLDS=1 K4_SR_NBK=2" "K4_SR_WLDS=1 K4_SR_NBK=1" "K4_SR_WLDS=0 K4_SR_NBK=1" "K4_SR_WLDS=1 K4_SR_NBK=2 K4_SR_2T_RPW=3" "K4_SR_WLDS=1 K4_SR_NBK=1 K4_SR_2T_RPW=3" "K4_SR_WLDS=1 K4_SR_NBK=1 K4_SR_2T_RPW=4"; do
  echo "$cfg"; env $cfg python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
done
cd /tmp && export TMPDIR=/tmp
for cfg in "K4_SR_WLDS=1 K4_SR_NBK=2" "K4_SR_WLDS=1 K4_SR_NBK=1"; do
rm -rf $R/gpurun_out/prof_tmp
env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/sr_frame_time.py f16x3 > $O/sr_trace.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_trace.csv" | head -1)
echo "== $cfg"
python $R/tools/sr_layer_table.py "$f" | head -14
done
rm -rf $R/gpurun_out/prof_tmp
