#!/bin/bash
# Round 6 A/B of the rgbnet arithmetics on the bench frame (GPU box): isolated call time + rocprofv3 kernel stats per K4_MLP value.
# usage: bash tools/r06_march_ab.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ab}
OUT=$R/gpurun_out/r06_march_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mlp in b2 b3 b2 b3; do
  echo "== K4_MLP=$mlp" | tee -a $OUT/call_time.log
  K4_MLP=$mlp python $R/tools/march_call_time.py 2>&1 | tail -1 | tee -a $OUT/call_time.log
done
for mlp in b2 b3; do
  rm -rf /tmp/prof_$mlp
  K4_MLP=$mlp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mlp -o run -- python $R/tools/march_call_time.py > $OUT/prof_$mlp.log 2>&1
  f=$(find /tmp/prof_$mlp -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then head -8 "$f" > $OUT/kernel_stats_$mlp.csv; echo "== rocprofv3 K4_MLP=$mlp"; cut -d, -f1-4 $OUT/kernel_stats_$mlp.csv | cut -c1-160; fi
done
