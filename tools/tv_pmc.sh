#!/bin/bash
# HBM-side traffic and cache behaviour of the grid-maintenance kernels (tools/tv_time.py) -- separate rocprofv3 passes per counter group, kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/tvpmc
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum"; do
  out=/tmp/tvpmc_$i; rm -rf $out
  timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/tv_time.py > $out.log 2>&1 || echo "group $i failed"
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python - <<PY
import csv, collections
rows = list(csv.DictReader(open('$f')))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    if 'k4_tv' in n or 'k4_adam_vec' in n:
        acc[n[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for n, d in acc.items():
    print(n, {k: (round(sorted(v)[len(v) // 2], 1), len(v)) for k, v in d.items()})
PY
  i=$((i+1))
done
