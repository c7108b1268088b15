#!/bin/bash
# Kernel timeline of the single-GPU 4K frame loop (rocprofv3 kernel trace): GPU busy / idle per frame and the largest gaps -- where the frame's time beyond its kernels goes.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${OUT:-r06_four_k_timeline}; mkdir -p $OUT
python $R/${FRAME_SCRIPT:-tools/four_k_timeline.py} 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_4k
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_4k -o run -- python $R/${FRAME_SCRIPT:-tools/four_k_timeline.py} > $OUT/timeline.log 2>&1
f=$(find /tmp/prof_4k -name '*kernel_trace.csv' | head -1)
tail -1 $OUT/timeline.log
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$f')))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '0'))) for r in rows)
ends = [e for s, e, n, q in ev if n.startswith('k4_conv_taps_b6_kernel')]          # conv_last: the decoder's final layer, once per frame
print('frames seen', len(ends))
for a, b in list(zip(ends, ends[1:]))[-4:]:
    cur = [(s, e, n, q) for s, e, n, q in ev if a <= s < b]
    busy, last_end, gaps = 0, a, []
    for s, e, n, q in cur:
        if s > last_end:
            gaps.append((s - last_end, n)); busy += e - s; last_end = e
        elif e > last_end:
            busy += e - last_end; last_end = e
    big = sorted(gaps, reverse=True)[:6]
    k4 = sum(e - s for s, e, n, q in cur if 'k4_conv' in n or 'k4_sft' in n)
    print(f'frame {(b - a) / 1e6:6.2f} ms: GPU busy (union) {busy / 1e6:6.2f} ms, idle {(b - a - busy) / 1e6:5.2f} ms in {len(gaps)} gaps, decoder kernels {k4 / 1e6:6.2f} ms, {len(cur)} kernels; largest gaps (us, before):',
          [(round(g / 1e3), n[:34]) for g, n in big])
a, b = ends[-3], ends[-2]
per = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in ev:
    if a <= s < b and not ('k4_conv' in n or 'k4_sft' in n):
        per[n[:70]] += e - s; cnt[n[:70]] += 1
print('non-decoder kernels of one frame:')
for n, t in per.most_common(14):
    print(f'  {t / 1e3:8.1f} us  x{cnt[n]:4d}  {n}')
PY
