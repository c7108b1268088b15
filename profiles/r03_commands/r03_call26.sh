#!/bin/bash
# SFT kernel: residual rows requested at the tile top, weight image by global_load_lds -- bit identity, SFT tests, frame time, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/sft_ab.py 2>&1 | grep '^C=' | cut -c1-100
python tools/sr_frame_hash.py f16x3 2>&1 | grep -i sha1          # 6e617288a80579b8d7e0b43f340ff48a32a4086e
timeout 300 python -m pytest tests/test_sr_gpu.py -m gpu -q -x -k "sft or golden" 2>&1 | tail -2
for rep in 1 2; do python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; python tools/sr_rank_share_time.py 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s -o run -- python $R/tools/sr_frame_time.py f16x3 > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); grep -i "sft\|conv_b6v2\|taps" "$f" | cut -c1-140
rm -rf $R/gpurun_out/prof_s
