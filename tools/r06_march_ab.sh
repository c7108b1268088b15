#!/bin/bash
# Round 6 A/B on the bench frame (GPU box): isolated marcher call time + rocprofv3 kernel stats per environment variant.
# usage: bash tools/r06_march_ab.sh <tag> "ENV=VAL ..." "ENV=VAL ..." ...     (each quoted argument is one variant; '' = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-ab}; shift
OUT=$R/gpurun_out/r06_march_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in "$@"; do
    echo "== [$v]" | tee -a $OUT/call_time.log
    env $v python $R/tools/march_call_time.py 2>&1 | tail -1 | tee -a $OUT/call_time.log
  done
done
i=0
for v in "$@"; do
  i=$((i+1)); rm -rf /tmp/prof_$i
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$i -o run -- python $R/tools/march_call_time.py > $OUT/prof_$i.log 2>&1
  f=$(find /tmp/prof_$i -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then head -8 "$f" > $OUT/kernel_stats_$i.csv; echo "== rocprofv3 [$v]"; python3 - "$OUT/kernel_stats_$i.csv" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if 'k4_' in r[0]: print('   %-62s calls %4s avg %9.1f us' % (r[0][:62], r[1], float(r[3]) / 1e3))
PY
  fi
done
