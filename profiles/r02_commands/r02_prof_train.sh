#!/bin/bash
# rocprofv3 --kernel-trace --stats of the marcher-only training iteration (tools/train_step_time.py) -> top kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2t
mkdir -p $O
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/train_step_time.py > $O/train_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -40 "$f" > $O/train_kernel_stats.csv
rm -rf $R/gpurun_out/prof_tmp
tail -2 $O/train_stats.log
cut -c1-200 $O/train_kernel_stats.csv | head -32
