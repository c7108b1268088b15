#!/bin/bash
# kernel time vs wall time of the heaviest 8-GPU rank's decoder share (tools/sr_rank_share_time.py): are the launches or the gaps the cost?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_rs
TS=${TS:-168} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -o run -- python $R/tools/sr_rank_share_time.py > /tmp/prof_rs.log 2>&1
tail -1 /tmp/prof_rs.log
f=$(find /tmp/prof_rs -name '*kernel_stats.csv' | head -1)
python - <<PY
import csv
rows = [r for r in csv.reader(open('$f')) if r and r[0] != 'Name']
it = 11
ker = [r for r in rows if 'k4_' in r[0] and 'absmax' not in r[0]]
tot = sum(float(r[2]) for r in ker)
print(f'decoder kernels: {tot / it / 1e6:.3f} ms per pass over {sum(int(r[1]) for r in ker) / it:.0f} launches')
for r in sorted(ker, key=lambda q: -float(q[2]))[:10]:
    print(f'{float(r[2]) / it / 1e3:8.1f} us/pass {int(r[1]) / it:5.1f} calls avg {float(r[3]) / 1e3:7.1f} us  {r[0][:80]}')
PY
