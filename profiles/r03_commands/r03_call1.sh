#!/bin/bash
# Round 3, GPU call 1: live-mask tests, its A/B on the bench line, kernel stats + PMC passes of the marcher at this tree, the whole
# -m gpu suite, the default bench line.  Writes gpurun_out/r3a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_march_gpu.py -m gpu -x -q > $O/tests_march.log 2>&1; echo "march_tests_rc=$?"
tail -4 $O/tests_march.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --sr-frames 0"
for cfg in "K4_LIVE_MASK=0" "K4_LIVE_MASK=1" "K4_LIVE_MASK=0" "K4_LIVE_MASK=1"; do
  tag=$(echo $cfg | tr ' =' '__')_$RANDOM
  env $cfg timeout 300 $B > $O/ab_$tag.json 2> $O/ab_$tag.err
  python - "$O/ab_$tag.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], 'value', d['value'], 'median_ms', d.get('ms_per_step_median'), 'isolated_ms', d['roofline']['kernel_ms'], 'mrays_iso', d['mrays_isolated'], 'frac', d['roofline']['frac'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
# kernel stats (isolated launches) with and without the live mask
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --sr-frames 0 --streams 1"
for lm in 1 0; do
  rm -rf $R/gpurun_out/prof_tmp
  K4_LIVE_MASK=$lm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/bench.py $ARGS > $O/marcher_stats_lm$lm.log 2>&1
  f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -12 "$f" > $O/marcher_kernel_stats_lm$lm.csv
  grep '"metric"' $O/marcher_stats_lm$lm.log | tail -1 > $O/marcher_bench_line_lm$lm.json
  cat $O/marcher_kernel_stats_lm$lm.csv | cut -c1-160
done
rm -rf $R/gpurun_out/prof_tmp
cd $R
for lm in 1 0; do
  K4_LIVE_MASK=$lm PMC_GROUPS="0 1 3 4" tools/pmc_run.sh r3a_lm$lm --steps 3 --warmup 1 --sr-frames 0 --no-cpu-baseline --no-extras --streams 1 > /dev/null 2>&1
  mv $R/gpurun_out/pmc_r3a_lm${lm}_summary.md $O/marcher_pmc_lm$lm.md; rm -f $R/gpurun_out/pmc_r3a_lm${lm}_*.log
done
grep -A12 "k4_geom3_kernel<0, false" $O/marcher_pmc_lm1.md | head -40
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests_rc=$?"
tail -6 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench_rc=$?"
python tools/bench_summary.py $O/bench.json 2>/dev/null || tail -c 600 $O/bench.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r3a/bench.json'))
    for k in ('reference_pipeline_rocm', 'dvgo_config0', 'joint_train_step', 'own_staged_pipeline'):
        print(k, json.dumps(d.get(k))[:700])
    print('median', d.get('ms_per_step_median'), d.get('ms_per_step_p90'), d['roofline'].get('traffic_source'))
except Exception as e:
    print('bench parse failed', e)
PY
tail -5 $O/bench.err
