# final evidence of round 4: full GPU suite, smoke, default bench line, rocprofv3 kernel stats of the decoder frame
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c48
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench_rc=$?"; grep '"metric"' $O/bench_default.log | tail -1 > $O/bench_default_line.json; python tools/bench_summary.py $O/bench_default_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_s
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s -o run -- python $R/tools/sr_frame_time.py f16x3p > $O/prof_s.log 2>&1
f=$(find $R/gpurun_out/prof_s -name "*kernel_stats.csv" | head -1); head -16 "$f" > $O/sr_kernel_stats.csv
grep "ms/frame" $O/prof_s.log > $O/sr_line.txt
rm -rf $R/gpurun_out/prof_s
cat $O/sr_line.txt; cut -c1-150 $O/sr_kernel_stats.csv | head -8
