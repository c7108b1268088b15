#!/bin/bash
# Round 5: A/B of the marcher's shading structure on the bench's LLFF frame (isolated call, HIP events; the output hash must not move).
# usage (GPU box): bash tools/march_variants.sh [outdir]
OUT=${1:-gpurun_out/r05_march}
mkdir -p $OUT
cd $(dirname $0)/..
run() { echo "== $*" | tee -a $OUT/variants.log; env "$@" python tools/march_call_time.py 2>&1 | tail -1 | tee -a $OUT/variants.log; }
run K4_MARCH_PRE=0
run K4_MARCH_PRE=1 K4_FEAT_MINW=4
run K4_MARCH_PRE=1 K4_FEAT_MINW=6
run K4_MARCH_PRE=1 K4_FEAT_MINW=8
run K4_MARCH_PRE=1 K4_FEAT_MINW=4 K4_SHADE_PRE_GRID_WG=3
run K4_MARCH_PRE=0 K4_K0_BRICK=1
run K4_MARCH_PRE=1 K4_K0_BRICK=1
