#!/bin/bash
# SQ counters of the decoder's convolution kernel on isolated layer launches (tools/conv_layer_time.py <case indices>)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2e
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
)
tag=$1; shift
i=0
for g in "${groups[@]}"; do
  out=$R/gpurun_out/pmc_sr_${tag}_$i
  rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/conv_layer_time.py "$@" > $out.log 2>&1 || echo "group $i failed (see $out.log)"
  i=$((i+1))
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_sr_${tag}_* > $R/gpurun_out/r2e/pmc_sr_${tag}.md
rm -rf $R/gpurun_out/pmc_sr_${tag}_[0-9]
grep -A22 "conv_b6" $R/gpurun_out/r2e/pmc_sr_${tag}.md
