#!/bin/bash
# SQ / MFMA counters of the decoder kernels for one arithmetic mode (argument: bf16x6 | bf16x3), separate --pmc passes, --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mode=${1:-bf16x6}
O=$R/gpurun_out/r2p_$mode
mkdir -p $O
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  out=$R/gpurun_out/pmc_srp_$i; rm -rf $out
  timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/sr_frame_time.py $mode > $O/pmc_$i.log 2>&1 || { echo "group $i failed"; tail -5 $O/pmc_$i.log; }
  i=$((i+1))
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_srp_* > $O/sr_pmc.md
rm -rf $R/gpurun_out/pmc_srp_[0-9]
grep -A12 "b6v2" $O/sr_pmc.md | head -40
