#!/bin/bash
# ablations of the shading kernels (WRONG results, timing only): PRE=0|1 bash tools/march_abl.sh [debug bits ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
PRE=${PRE:-1}
for dbg in ${@:-0 1 2 256 258}; do echo "== K4_MARCH_PRE=$PRE K4_DEBUG=$dbg"; bash tools/march_prof.sh abl${PRE}_$dbg K4_MARCH_PRE=$PRE K4_DEBUG=$dbg 2>&1 | grep -E "shade|feat_kernel|geom3|isolated" | awk -F, '{ if (NF>=4) printf "   %-60s %9.1f us\n", substr($1,1,60), $4/1000; else print }'; done
