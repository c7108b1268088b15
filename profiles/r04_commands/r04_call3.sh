cd /root/repo
mkdir -p gpurun_out/r4c3
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "p16 or f16x3p" > gpurun_out/r4c3/tests.log 2>&1; tail -5 gpurun_out/r4c3/tests.log
for dbg in 0 64 128 192 256 512 576; do
  echo "== K4_SR_DEBUG=$dbg" >> gpurun_out/r4c3/ablate.log
  K4_SR_DEBUG=$dbg K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 120 python tools/p16_layer_time.py 0 3 4 >> gpurun_out/r4c3/ablate.log 2>&1
  K4_SR_DEBUG=$dbg K4_TOOL_ONLY=p16 timeout 120 python tools/p16_layer_time.py 3 7 8 >> gpurun_out/r4c3/ablate.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r4c3/ablate.log
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
         "FETCH_SIZE" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  out=$R/gpurun_out/r4c3/pmc_$i
  K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/p16_layer_time.py 3 > $out.log 2>&1 || echo "pmc group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_by_grid.py $R/gpurun_out/r4c3/pmc_* > $R/gpurun_out/r4c3/pmc_summary.md 2>&1
rm -rf $R/gpurun_out/r4c3/pmc_[0-9]
cat $R/gpurun_out/r4c3/pmc_summary.md
