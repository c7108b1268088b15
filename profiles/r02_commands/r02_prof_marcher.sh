#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) and SQ counters of the marcher kernels for a set of env configurations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2c
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1"
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' =' '__')
  rm -rf $R/gpurun_out/prof_tmp
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/bench.py $ARGS > $R/gpurun_out/r2c/stats_$tag.log 2>&1
  f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -8 "$f" > $R/gpurun_out/r2c/stats_$tag.csv
  echo "== $cfg"; cut -d, -f1-4 $R/gpurun_out/r2c/stats_$tag.csv | cut -c1-150
  rm -rf $R/gpurun_out/prof_tmp
  cd $R && env $cfg PMC_GROUPS="${PMC_GROUPS_SEL:-0 1}" tools/pmc_run.sh r2c_$tag --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
  mv $R/gpurun_out/pmc_r2c_${tag}_summary.md $R/gpurun_out/r2c/pmc_$tag.md; rm -f $R/gpurun_out/pmc_r2c_${tag}_*.log
  cd /tmp
done
