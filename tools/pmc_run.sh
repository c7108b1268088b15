#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...>   (on the GPU box; writes gpurun_out/pmc_<tag>_<n>/ and a summary)
# Counter groups are collected in SEPARATE rocprofv3 passes, with --kernel-trace only (never with sys/hip traces).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  if [ -n "$PMC_GROUPS" ] && [[ " $PMC_GROUPS " != *" $i "* ]]; then i=$((i+1)); continue; fi
  out=$R/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/bench.py "$@" > $out.log 2>&1 || echo "group $i failed (see $out.log)"
  i=$((i+1))
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_* > $R/gpurun_out/pmc_${tag}_summary.md
rm -rf $R/gpurun_out/pmc_${tag}_[0-9]      # raw per-dispatch CSVs are tens of MB; the summary is what is kept
cat $R/gpurun_out/pmc_${tag}_summary.md
