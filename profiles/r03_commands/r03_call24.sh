#!/bin/bash
# persistent workgroups with the next tile's first chunk prefetched (production fp16 3x3 kernel) against one workgroup per tile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/sr_frame_hash.py f16x3 2>&1 | grep -i sha1          # 6e617288a80579b8d7e0b43f340ff48a32a4086e
timeout 900 python -m pytest tests/test_sr_gpu.py tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do for v in "" nopersist; do lib=""; [ -n "$v" ] && lib=$R/4k-nerf_amd/lib4k_hip_$v.so
  echo "== ${v:-persistent}"; K4_LIB=$lib python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; K4_LIB=$lib python tools/sr_rank_share_time.py 2>&1 | tail -1; done; done
for v in "" nopersist; do lib=""; [ -n "$v" ] && lib=$R/4k-nerf_amd/lib4k_hip_$v.so; echo "== layers ${v:-persistent}"; K4_LIB=$lib K4_SR_MODE=f16x3 python tools/conv_layer_time.py 0 3 4 7 6 2>&1 | grep cin; done
