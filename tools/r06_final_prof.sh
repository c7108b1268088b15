#!/bin/bash
# Round-6 evidence run on the GPU box: the default bench line, rocprofv3 kernel stats of the marcher and decoder commands, PMC passes of the
# marcher (instruction mix, L1->L2 requests, FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only) and of the decoder frame (matrix / vector
# instruction counts per kernel and grid) -> gpurun_out/r6final/ (copied into profiles/ afterwards).   usage: tools/r06_final_prof.sh <commit>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6final
mkdir -p $O
cd $R
COMMIT=${1:-unknown}
if [ "$SKIP_BENCH" != "1" ]; then timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench_rc=$?"; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default_line.json; python tools/bench_summary.py $O/bench_default_line.json; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_m $R/gpurun_out/prof_s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1 > $O/prof_m.log 2>&1
f=$(find $R/gpurun_out/prof_m -name "*kernel_stats.csv" | head -1); head -14 "$f" > $O/marcher_kernel_stats.csv
grep '"metric"' $O/prof_m.log | tail -1 > $O/marcher_bench_line_under_rocprof.json
rm -rf $R/gpurun_out/prof_m
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s -o run -- python $R/tools/sr_frame_time.py f16x3p > $O/prof_s.log 2>&1
f=$(find $R/gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); head -16 "$f" > $O/sr_kernel_stats.csv
grep "ms/frame" $O/prof_s.log > $O/sr_line.txt
rm -rf $R/gpurun_out/prof_s
i=0
for g in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  out=$R/gpurun_out/pmc_r6m_$i; rm -rf $out
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/bench.py --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > $out.log 2>&1 || echo "marcher pmc group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_r6m_* > $O/marcher_pmc.md 2>&1
rm -rf $R/gpurun_out/pmc_r6m_[0-9]
cd $R && python tools/make_traffic_json.py $O/marcher_pmc.md $O/marcher_traffic.json $COMMIT > /dev/null && echo traffic_json_ok
cd /tmp
i=0
for g in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  out=$R/gpurun_out/pmc_r6s_$i; rm -rf $out
  # (the whole-frame command segfaults inside rocprofv3's counter collection on these boxes -- as the WRITE_SIZE passes did in round 3; the per-layer
  #  tool, four windows per launch = the launches of a 4K frame, does not)
  K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $out -o run -- python $R/tools/p16_layer_time.py 0 3 4 5 6 > $out.log 2>&1 || echo "decoder pmc group $i failed"
  i=$((i+1))
done
python $R/tools/pmc_by_grid.py $R/gpurun_out/pmc_r6s_* > $O/sr_pmc_by_grid_raw.md 2>&1
rm -rf $R/gpurun_out/pmc_r6s_[0-9]
# joint training iteration on the final tree: phase events (untraced), blocks, kernel timeline of one iteration
cd $R
timeout 300 python tools/joint_phase_events.py > $O/joint_phase_events.txt 2>/dev/null
BLOCKS=8 SHOW_BLOCKS=1 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -2 > $O/joint_step_time.txt
# ... and an iteration after tv_before (STEP0 >= 10000: no total variation, k0 stepped from the scatter image)
STEP0=20000 timeout 300 python tools/joint_phase_events.py > $O/joint_phase_events_after_tv_before.txt 2>/dev/null
STEP0=20000 BLOCKS=8 SHOW_BLOCKS=1 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -2 > $O/joint_step_time_after_tv_before.txt
STEP0=20000 OUT=r6final/joint_timeline_after_tv_before timeout 400 bash tools/joint_timeline_detail.sh > $O/joint_timeline_detail_after_tv_before.txt 2>&1
OUT=r6final/joint_timeline timeout 400 bash tools/joint_timeline_detail.sh > $O/joint_timeline_detail.txt 2>&1
ITERS=20 timeout 400 bash tools/joint_prof.sh > $O/joint_kernel_stats.txt 2>&1; cp $R/gpurun_out/r05_joint/kernel_stats.csv $O/joint_iteration_kernel_stats.csv 2>/dev/null
cat $O/joint_phase_events.txt $O/joint_step_time.txt $O/joint_step_time_after_tv_before.txt
ls -la $O
