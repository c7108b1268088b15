#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for pf in 0 1 2; do
  echo "== K4_SR_PF2=$pf"; K4_SR_PF2=$pf python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
  K4_SR_PF2=$pf python tools/sr_rank_share_time.py 2>&1 | tail -2
  K4_SR_PF2=$pf K4_SR_MODE=f16x3 python tools/conv_layer_time.py 3 4 9 2>&1 | grep cin
done
timeout 600 env K4_SR_PF2=1 python -m pytest tests/test_sr_gpu.py -m gpu -q -x 2>&1 | tail -3
