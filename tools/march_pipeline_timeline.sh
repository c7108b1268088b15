#!/bin/bash
# Kernel trace of bench.py's pipelined marcher loop (3 streams): per frame interval, GPU busy (union), time with 1 / 2 / 3 kernels running, idle.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_mp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_mp -o run -- python $R/bench.py --streams ${STREAMS:-3} --steps 60 --warmup 12 --no-cpu-baseline --no-extras --sr-frames 0 > /tmp/mp.log 2>&1
f=$(find /tmp/prof_mp -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv, json
line = [l for l in open('/tmp/mp.log') if l.startswith('{')][-1]
d = json.loads(line); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'isolated', d['mrays_isolated'])
rows = list(csv.DictReader(open('$f')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:30], r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows
            if ('k4_geom3' in r['Kernel_Name'] and '<0, true' not in r['Kernel_Name']) or 'k4_shade' in r['Kernel_Name'] or 'k4_order' in r['Kernel_Name'])
# the timed region: the 60 consecutive frames (3 kernels each) with the shortest span = the pipelined loop
best = min(range(0, len(ev) - 179), key=lambda i: max(e for s, e, n, q in ev[i:i + 180]) - ev[i][0])
ev = ev[best:best + 180]
t0, t1 = ev[0][0], max(e for s, e, n, q in ev)
pts = sorted([(s, 1) for s, e, n, q in ev] + [(e, -1) for s, e, n, q in ev])
conc, last, hist = 0, t0, {}
for t, dlt in pts:
    hist[conc] = hist.get(conc, 0) + (t - last); last = t; conc += dlt
tot = t1 - t0
print('region %.2f ms for 60 frames = %.3f ms per frame' % (tot / 1e6, tot / 60e6))
for k in sorted(hist): print('  %d kernels running: %5.1f %% of the time' % (k, 100.0 * hist[k] / tot))
import collections
per = collections.defaultdict(list)
for s, e, n, q in ev: per[n].append((e - s) / 1e3)
for n, v in per.items(): print('  %-32s avg %7.1f us (isolated: geom3 ~400, shade ~345)' % (n, sum(v) / len(v)))
PY
