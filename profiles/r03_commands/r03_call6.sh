#!/bin/bash
# Round 3, GPU call 6: is the decoder memory-bound?  FETCH / WRITE / L2 counters per decoder kernel for one frame (K4_SR_WLDS=0).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export K4_SR_WLDS=0 K4_SR_NBK=1
groups=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
)
i=0
for g in "${groups[@]}"; do
  out=$R/gpurun_out/pmc_sr_$i; rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/sr_frame_time.py f16x3 > $out.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_by_grid.py $R/gpurun_out/pmc_sr_* > $O/sr_pmc_by_grid.md
rm -rf $R/gpurun_out/pmc_sr_[0-9] $R/gpurun_out/pmc_sr_*.log
cat $O/sr_pmc_by_grid.md | head -120
