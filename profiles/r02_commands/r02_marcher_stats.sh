#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the marcher kernels for env configurations (arguments)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1"
for cfg in "$@"; do
  rm -rf /tmp/prof_tmp
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tmp -o run -- python $R/bench.py $ARGS > /tmp/stats.log 2>&1
  f=$(find /tmp/prof_tmp -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"
  python - "$f" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r and any(k in r[0] for k in ('k4_geom3_kernel<0, false', 'k4_shade_kernel', 'k4_order')):
        print('  %-46s calls %4s avg %8.1f us' % (r[0][:46], r[1], float(r[3]) / 1e3))
PY
done
