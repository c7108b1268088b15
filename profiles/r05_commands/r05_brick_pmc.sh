#!/bin/bash
# Round 5, verdict item 1(a): L2 hit rate / L1 pending-miss stall of the shading kernel with k0 in [X][Y][Z][12] rows vs 4x4x4-voxel bricks.
# Separate --pmc passes (--kernel-trace only) of the marcher-only bench command.  usage (GPU box): bash tools/r05_brick_pmc.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for brick in 0 1; do
  i=0
  for g in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    out=$R/gpurun_out/pmc_brick${brick}_$i; rm -rf $out
    K4_K0_BRICK=$brick timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/bench.py --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > $out.log 2>&1 || echo "brick $brick group $i failed"
    i=$((i+1))
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_brick${brick}_* > $R/gpurun_out/r05_pmc_brick${brick}.md 2>&1
  rm -rf $R/gpurun_out/pmc_brick${brick}_[0-9]
done
grep -A12 "k4_shade_kernel" $R/gpurun_out/r05_pmc_brick0.md | head -14; grep -A12 "k4_shade_kernel" $R/gpurun_out/r05_pmc_brick1.md | head -14
