#!/bin/bash
# k4_conv_f16p_kernel (staging inside the matrix phase, one barrier per chunk) against the v2 fp16 kernel (K4_SR_DEBUG=32 selects it)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 0 32; do echo "== K4_SR_DEBUG=$v"; K4_SR_DEBUG=$v python tools/sr_frame_hash.py f16x3 2>&1 | grep -i sha1; done
timeout 900 python -m pytest tests/test_sr_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do for v in 0 32; do
  echo "== K4_SR_DEBUG=$v"; K4_SR_DEBUG=$v python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
  K4_SR_DEBUG=$v python tools/sr_rank_share_time.py 2>&1 | tail -1
done; done
for v in 0 32; do echo "== layers K4_SR_DEBUG=$v"; K4_SR_DEBUG=$v K4_SR_MODE=f16x3 python tools/conv_layer_time.py 0 3 4 7 9 2>&1 | grep cin; done
