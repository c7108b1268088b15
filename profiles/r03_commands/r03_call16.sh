#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 0 1; do echo "== K4_SR_V3=$v"; K4_SR_V3=$v python tools/sr_frame_hash.py f16x3 2>&1 | grep sha1; done
timeout 600 python -m pytest tests/test_sr_gpu.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0 1 0; do
  echo "== K4_SR_V3=$v"; K4_SR_V3=$v python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
  K4_SR_V3=$v python tools/sr_rank_share_time.py 2>&1 | tail -1
done
for v in 1 0; do echo "== K4_SR_V3=$v"; K4_SR_V3=$v K4_SR_MODE=f16x3 python tools/conv_layer_time.py 0 3 4 7 9 2>&1 | grep cin; done
for r in 3 4; do echo "== V3 K4_SR_2T_RPW=$r"; K4_SR_2T_RPW=$r python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; done
