#!/bin/bash
# Round-3 evidence run on the GPU box: GPU tests, the default bench line, rocprofv3 kernel stats of the marcher and decoder commands,
# FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/r3final/ (copied into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
COMMIT=${1:-unknown}
if [ "$SKIP_TESTS" != "1" ]; then timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests_gpu.log 2>&1; echo "gpu_tests_rc=$?"; tail -3 $O/tests_gpu.log; fi
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench_rc=$?"; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_m $R/gpurun_out/prof_s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m -o run -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1 > $O/prof_m.log 2>&1
f=$(find $R/gpurun_out/prof_m -name "*kernel_stats.csv" | head -1); head -14 "$f" > $O/marcher_kernel_stats.csv
grep '"metric"' $O/prof_m.log | tail -1 > $O/marcher_bench_line_under_rocprof.json
rm -rf $R/gpurun_out/prof_m
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s -o run -- python $R/tools/sr_frame_time.py f16x3 > $O/prof_s.log 2>&1
f=$(find $R/gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); head -14 "$f" > $O/sr_kernel_stats.csv
grep "ms/frame" $O/prof_s.log > $O/sr_line.txt
rm -rf $R/gpurun_out/prof_s
cd $R
PMC_GROUPS="3 4" tools/pmc_run.sh r3final --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
cp $R/gpurun_out/pmc_r3final_summary.md $O/marcher_pmc_fetch_write.md
python tools/make_traffic_json.py $O/marcher_pmc_fetch_write.md $O/marcher_traffic.json $COMMIT > /dev/null && echo traffic_json_ok
ls -la $O
