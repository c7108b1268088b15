#!/bin/bash
# A-fragment reuse (dx-major walk) of the fp16 3x3 kernel: parity tests, frame time A/B against the previous walk and a 2-deep A ring
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_sr_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
for v in "" noreuse ring2; do
  lib=""; [ -n "$v" ] && lib=$R/4k-nerf_amd/lib4k_hip_$v.so
  echo "== variant '${v:-default (reuse, ring 3)}'"; K4_LIB=$lib python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
done; done
for v in "" noreuse; do lib=""; [ -n "$v" ] && lib=$R/4k-nerf_amd/lib4k_hip_$v.so; echo "== layers, variant '${v:-default}'"; K4_LIB=$lib K4_SR_MODE=f16x3 python tools/conv_layer_time.py 0 3 4 7 9 2>&1 | grep cin; done
