#!/bin/bash
# Kernel-by-kernel timeline of ONE joint training iteration (rocprofv3 kernel trace): start offset, duration, stream, gap to the previous kernel
# on the same stream -- where the GPU waits for the host and where kernels run back to back.  OUT=<dir under gpurun_out>, env passes through.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${OUT:-r06_joint_detail}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_jd
ITERS=6 BLOCKS=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_jd -o run -- python $R/tools/joint_step_time.py > $OUT/timeline.log 2>&1
f=$(find /tmp/prof_jd -name '*kernel_trace.csv' | head -1)
tail -1 $OUT/timeline.log
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$f')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows)
starts = [s for s, e, n, q in ev if n.startswith('k_train_select_mpi')]
a, b = starts[-3], starts[-2]
cur = [(s, e, n, q) for s, e, n, q in ev if a <= s < b]
main_q = collections.Counter(q for s, e, n, q in cur).most_common(1)[0][0]
last = {}
out = open('$OUT/iteration_kernels.csv', 'w')
out.write('start_us,dur_us,gap_same_stream_us,stream,kernel\n')
for s, e, n, q in cur:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    out.write(f'{(s - a) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},{q},"{n[:70]}"\n')
out.close()
# phases on the main stream: runs of kernels separated by gaps > 30 us
ph, cur_ph = [], None
prev_end = None
for s, e, n, q in cur:
    if q != main_q:
        continue
    if prev_end is None or s - prev_end > 30000:
        cur_ph = [s, e, 0, 0, n, (s - prev_end) / 1e3 if prev_end else 0.0]
        ph.append(cur_ph)
    cur_ph[1] = e; cur_ph[2] += 1; cur_ph[3] += e - s
    prev_end = e
print(f'iteration {(b - a) / 1e6:.2f} ms, {len(cur)} kernels, main stream {main_q}: {sum(1 for c in cur if c[3] == main_q)} kernels, busy {sum(e - s for s, e, n, q in cur if q == main_q) / 1e6:.2f} ms')
print('phases on the main stream (start ms, span ms, kernels, busy ms, gap before us, first kernel):')
for s, e, k, busy, n, gap in ph:
    print(f'  {(s - a) / 1e6:6.2f} {(e - s) / 1e6:6.2f} {k:4d} {busy / 1e6:6.2f} {gap:7.0f}  {n[:60]}')
per = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in cur:
    per[(q, n[:60])] += e - s; cnt[(q, n[:60])] += 1
print('top kernels of the iteration:')
for (q, n), t in per.most_common(16):
    print(f'  {t / 1e3:8.1f} us  x{cnt[(q, n)]:3d}  stream {q}  {n}')
PY
