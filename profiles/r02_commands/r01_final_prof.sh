cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_m $R/gpurun_out/prof_s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1 > $R/gpurun_out/prof_m.log 2>&1
f=$(find $R/gpurun_out/prof_m -name "*kernel_stats.csv" | head -1); head -14 "$f" > $R/gpurun_out/final_marcher_stats.csv
grep '"metric"' $R/gpurun_out/prof_m.log | tail -1 > $R/gpurun_out/final_marcher_bench_line.json
rm -rf $R/gpurun_out/prof_m
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s -o run -- python $R/tools/sr_frame_time.py bf16x6 > $R/gpurun_out/prof_s.log 2>&1
f=$(find $R/gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); head -14 "$f" > $R/gpurun_out/final_sr_stats.csv
grep "ms/frame" $R/gpurun_out/prof_s.log > $R/gpurun_out/final_sr_line.txt
rm -rf $R/gpurun_out/prof_s
cd $R
PMC_GROUPS="3 4" tools/pmc_run.sh final --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
ls $R/gpurun_out
