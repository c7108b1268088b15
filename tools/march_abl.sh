#!/bin/bash
# ablations of the shading kernel (WRONG results, timing only): bash tools/march_abl.sh [K4_DEBUG bits ...]   (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for dbg in ${@:-0 1 2 512}; do echo "== K4_DEBUG=$dbg"; bash tools/march_prof.sh abl_$dbg K4_DEBUG=$dbg 2>&1 | grep -E "shade|geom3|isolated" | awk -F, '{ if (NF>=4) printf "   %-60s %9.1f us\n", substr($1,1,60), $4/1000; else print }'; done
