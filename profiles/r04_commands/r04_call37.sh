# counters of the pre-split 3x3 kernel per layer shape (incl. the x2-upsampling layer in its phase-pair form): separate --pmc passes, --kernel-trace only
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c37
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  out=$R/gpurun_out/pmc_r4s_$i; rm -rf $out
  K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/p16_layer_time.py 0 3 4 5 6 > $out.log 2>&1 || echo "decoder pmc group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_by_grid.py $R/gpurun_out/pmc_r4s_* > $O/sr_pmc_by_grid_raw.md 2>&1
rm -rf $R/gpurun_out/pmc_r4s_[0-9]
head -30 $O/sr_pmc_by_grid_raw.md | cut -c1-400
