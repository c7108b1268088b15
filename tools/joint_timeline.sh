#!/bin/bash
# Timeline of the joint training iteration from a rocprofv3 kernel trace: per iteration, the time the GPU has at least one kernel running (union of
# kernel intervals over all streams), the idle gaps, and the busy time per stream -- is the iteration paced by the host or by the GPU?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_joint; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_jt
ITERS=6 BLOCKS=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_jt -o run -- python $R/tools/joint_step_time.py > $OUT/timeline.log 2>&1
f=$(find /tmp/prof_jt -name '*kernel_trace.csv' | head -1)
tail -1 $OUT/timeline.log
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$f')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows)
# iteration boundaries: the sample-selection kernel runs once per iteration
starts = [s for s, e, n, q in ev if n.startswith('k_train_select_mpi')]
print('iterations seen', len(starts))
for a, b in list(zip(starts, starts[1:]))[-8:]:
    cur = [(s, e, n, q) for s, e, n, q in ev if a <= s < b]
    busy, last_end, gaps = 0, a, []
    for s, e, n, q in cur:
        if s > last_end:
            gaps.append((s - last_end, n))
            busy += e - s
            last_end = e
        elif e > last_end:
            busy += e - last_end
            last_end = e
    per_q = collections.Counter()
    for s, e, n, q in cur:
        per_q[q] += e - s
    big = sorted(gaps, reverse=True)[:4]
    print(f'iteration {(b - a) / 1e6:6.2f} ms: GPU busy (union) {busy / 1e6:6.2f} ms, idle {(b - a - busy) / 1e6:5.2f} ms in {len(gaps)} gaps; per stream/queue busy',
          {k: round(v / 1e6, 2) for k, v in per_q.items()}, '; largest gaps (us, before kernel):', [(round(g / 1e3), n[:28]) for g, n in big])
PY
