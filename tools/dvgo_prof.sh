#!/bin/bash
# rocprofv3 kernel stats of the fused DirectVoxGO call at 800x800 (tools/dvgo_call_time.py): which kernel holds the call
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_dvgo; mkdir -p $OUT
python $R/tools/dvgo_call_time.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dvgo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dvgo -o run -- python $R/tools/dvgo_call_time.py > $OUT/prof.log 2>&1
f=$(find /tmp/prof_dvgo -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
python - <<PY
import csv
rows = [r for r in csv.reader(open('$OUT/kernel_stats.csv')) if r and r[0] != 'Name']
for r in sorted(rows, key=lambda q: -float(q[2]))[:8]:
    print(f'{int(r[1]):5d} calls  avg {float(r[3]) / 1e3:9.1f} us  {r[0][:110]}')
PY
