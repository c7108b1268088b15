#!/bin/bash
# conv_last (k4_conv_taps_b6_kernel) with the next chunk's activations and weights requested before the matrix instructions; window grouping A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/sr_frame_hash.py f16x3 2>&1 | grep -i sha1          # 6e617288a80579b8d7e0b43f340ff48a32a4086e
timeout 300 python -m pytest tests/test_sr_gpu.py -m gpu -q -x -k "golden or taps or conv_last" 2>&1 | tail -2
for g in 8 1 2; do echo "== K4_SR_GROUP=$g"; K4_SR_GROUP=$g python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s -o run -- python $R/tools/sr_frame_time.py f16x3 > /dev/null 2>&1
f=$(find $R/gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); grep -i "sft\|conv_b6v2\|taps" "$f" | cut -c1-140
rm -rf $R/gpurun_out/prof_s
