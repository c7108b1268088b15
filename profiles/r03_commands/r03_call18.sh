#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -m gpu -q -x > $O/tests_train.log 2>&1; echo "train_tests_rc=$?"; tail -5 $O/tests_train.log
for m in fused convs fused; do echo "== K4_TRAIN_SFT=$m"; K4_TRAIN_SFT=$m python tools/joint_step_time.py 2>&1 | grep "joint iteration"; done; echo "== graph"; K4_TRAIN_GRAPH=1 timeout 300 python tools/joint_step_time.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tmp
ITERS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/joint_step_time.py > $O/joint_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -60 "$f" > $O/joint_kernel_stats.csv
python - "$O/joint_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('top kernels (8 iterations incl. warm-up): total listed', round(tot / 1e6, 2), 'ms')
for r in rows[:26]:
    print(f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms  x{int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:90]}")
PY
rm -rf $R/gpurun_out/prof_tmp
