#!/bin/bash
# rocprofv3 kernel stats of the isolated marcher call under a set of env knobs: bash tools/march_prof.sh <tag> [ENV=VAL ...]  (GPU box)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_march
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python $R/tools/march_call_time.py > $OUT/prof_$TAG.log 2>&1
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -10 "$f" > $OUT/kernel_stats_$TAG.csv; echo "== $TAG $*"; cut -d, -f1-4 $OUT/kernel_stats_$TAG.csv | cut -c1-150; fi
tail -1 $OUT/prof_$TAG.log | grep isolated
