#!/bin/bash
# Round-2 evidence run on the GPU box: rocprofv3 --kernel-trace --stats of the marcher call (isolated launches) and of the decoder,
# PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, SQ counters) of the marcher kernels, SQ counters of the decoder kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
mkdir -p $O
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1"
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/bench.py $ARGS > $O/marcher_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -10 "$f" > $O/marcher_kernel_stats.csv
grep '"metric"' $O/marcher_stats.log | tail -1 > $O/marcher_bench_line.json
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/sr_frame_time.py bf16x6 > $O/sr_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -12 "$f" > $O/sr_kernel_stats.csv
grep "ms/frame" $O/sr_stats.log > $O/sr_line.txt
rm -rf $R/gpurun_out/prof_tmp
cd $R
PMC_GROUPS="0 1 2 3 4" tools/pmc_run.sh r2f --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
mv $R/gpurun_out/pmc_r2f_summary.md $O/marcher_pmc.md; rm -f $R/gpurun_out/pmc_r2f_*.log
# decoder counters on the whole frame (all conv / sft kernels)
cd /tmp
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  out=$R/gpurun_out/pmc_srf_$i; rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/sr_frame_time.py bf16x6 > $out.log 2>&1 || echo "group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_srf_* > $O/sr_pmc.md
rm -rf $R/gpurun_out/pmc_srf_[0-9] $R/gpurun_out/pmc_srf_*.log
ls -la $O
# the opt-in 2-term arithmetic on the default kernel, and the marcher training iteration
cd /tmp
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/sr_frame_time.py bf16x3 > $O/sr2_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -12 "$f" > $O/sr_bf16x3_kernel_stats.csv
grep "ms/frame" $O/sr2_stats.log > $O/sr_bf16x3_line.txt
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/train_step_time.py > $O/train_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -30 "$f" > $O/train_kernel_stats.csv
tail -1 $O/train_stats.log > $O/train_line.txt
rm -rf $R/gpurun_out/prof_tmp
ls -la $O
