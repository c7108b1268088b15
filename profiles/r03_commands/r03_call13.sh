#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export K4_SR_MODE=f16x3
for cfg in "K4_SR_WLDS=0 K4_SR_NBK=1" "K4_SR_WLDS=1 K4_SR_NBK=1" "K4_SR_WLDS=1 K4_SR_NBK=2" "K4_SR_WLDS=0 K4_SR_NBK=2"; do
  echo "== $cfg"; env $cfg python tools/conv_layer_time.py 3 4 7 2>&1 | grep "cin"
  env $cfg python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
done
timeout 600 env K4_SR_WLDS=1 K4_SR_NBK=2 python -m pytest tests/test_sr_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 env K4_SR_WLDS=1 K4_SR_NBK=1 python -m pytest tests/test_sr_gpu.py -m gpu -q -x 2>&1 | tail -3
