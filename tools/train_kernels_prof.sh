#!/bin/bash
# rocprofv3 kernel durations of tools/train_kernels_time.py (single stream: no neighbours).  usage: train_kernels_prof.sh [tag] ; K4_LIB selects a variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-cur}
OUT=$R/gpurun_out/r05_trk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_trk_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trk_$TAG -o run -- python $R/tools/train_kernels_time.py > $OUT/$TAG.log 2>&1
f=$(find /tmp/prof_trk_$TAG -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$f')))
# launches in order; group consecutive runs of the same kernel name (21 launches per case) and print the median of each run
runs = []
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = (int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)))
    key = (n, g)
    if runs and runs[-1][0] == key: runs[-1][1].append(d)
    else: runs.append([key, [d]])
seen = collections.OrderedDict()
for key, ds in runs:
    if not any(k in key[0] for k in ('wgrad', 'b6v2', 'sft_train', 'wg_zero')): continue
    seen.setdefault(key, []).extend(ds)
for (n, g), ds in seen.items():
    ds.sort()
    print(f'{ds[len(ds) // 2]:8.1f} us median  {ds[0]:8.1f} min  n={len(ds):3d}  grid {g:7d}  {n[:70]}')
PY
