#!/bin/bash
# rocprofv3 kernel stats of the joint training iteration (tools/joint_step_time.py, 3 warm-up + ITERS iterations): which kernels hold the GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_joint; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_joint
ITERS=${ITERS:-20} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_joint -o run -- python $R/tools/joint_step_time.py > $OUT/prof.log 2>&1
f=$(find /tmp/prof_joint -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
tail -1 $OUT/prof.log
python - <<PY
import csv
rows = [r for r in csv.reader(open('$OUT/kernel_stats.csv')) if r and r[0] != 'Name']
tot = sum(float(r[2]) for r in rows)
it = max(1, sum(int(r[1]) for r in rows if r[0].startswith('k_train_select_mpi')))      # one sample-selection launch per iteration (warm-up included)
print(f'GPU kernel time per iteration: {tot / it / 1e6:.2f} ms over {sum(int(r[1]) for r in rows) / it:.0f} launches')
for r in sorted(rows, key=lambda q: -float(q[2]))[:22]:
    print(f'{float(r[2]) / it / 1e3:8.1f} us/iter  {int(r[1]) / it:6.1f} calls/iter  avg {float(r[3]) / 1e3:7.1f} us  {r[0][:90]}')
PY
