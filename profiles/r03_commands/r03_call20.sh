#!/bin/bash
# where one rank's share of the 8-GPU 4K job (3 windows of 209x209) spends its ~8.5 ms: kernel stats of tools/sr_rank_share_time.py
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/sr_rank_share_time.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_rs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rs -o run -- python $R/tools/sr_rank_share_time.py > $R/gpurun_out/rank_share_prof.log 2>&1
f=$(find $R/gpurun_out/prof_rs -name "*kernel_stats.csv" | head -1); head -16 "$f" > $R/gpurun_out/rank_share_kernel_stats.csv
python - "$R/gpurun_out/rank_share_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms  x{int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:100]}")
PY
rm -rf $R/gpurun_out/prof_rs
