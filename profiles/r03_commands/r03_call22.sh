#!/bin/bash
# fewer vector instructions in the shipped fp16 3x3 kernel (fma_mix split, integer chunk maxima, buffer loads): bit identity, tests, timing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/sr_frame_hash.py f16x3 2>&1 | grep -i sha1          # before: 6e617288a80579b8d7e0b43f340ff48a32a4086e
timeout 900 python -m pytest tests/test_sr_gpu.py tests/test_sr_train_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; python tools/sr_rank_share_time.py 2>&1 | tail -1; done
K4_SR_MODE=f16x3 python tools/conv_layer_time.py 0 3 4 7 9 2>&1 | grep cin
