#!/bin/bash
# rocprofv3 --kernel-trace --stats of the joint training iteration (tools/joint_step_profile.py: 5 iterations) -> top kernels + total GPU time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/joint_step_profile.py > $O/joint_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); cp "$f" $O/joint_kernel_stats.csv
rm -rf $R/gpurun_out/prof_tmp
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/joint_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
calls = sum(int(r['Calls']) for r in rows)
print('total kernel ms', round(tot, 2), 'launches', calls, '(5 iterations incl. warm-up + first-use packing)')
for r in rows[:22]:
    print('%-90s %6d %9.2f ms %9.1f us' % (r['Name'].split('(')[0][:90], int(r['Calls']), float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
