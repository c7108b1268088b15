cd /root/repo
mkdir -p gpurun_out/r4c4
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r4c4/gpu_tests.log 2>&1; tail -8 gpurun_out/r4c4/gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c4/smoke.log 2>&1; tail -3 gpurun_out/r4c4/smoke.log
timeout 600 python bench.py > gpurun_out/r4c4/bench.json 2> gpurun_out/r4c4/bench.err; tail -3 gpurun_out/r4c4/bench.err; python tools/bench_summary.py gpurun_out/r4c4/bench.json 2>/dev/null | head -60
