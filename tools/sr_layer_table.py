"""Per-kernel / per-grid table of ONE decoder frame from a rocprofv3 --kernel-trace csv of tools/sr_frame_time.py <mode> (3 frames)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k4_' in r['Kernel_Name']]
n = len(rows) // 3
last = rows[-n:]
agg = collections.OrderedDict()
for r in last:
    k = (r['Kernel_Name'].split('(')[0], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print(f'one frame: {len(last)} launches, {tot / 1e3:.2f} ms of kernel time')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{v[1]:9.1f} us  x{v[0]:3d}  avg {v[1] / v[0]:8.1f}  grid {k[1]:>9s}  {k[0][:80]}')
