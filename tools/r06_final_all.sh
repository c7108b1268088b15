bash tools/r06_final_prof.sh ${1:-HEAD} 2>&1 | tail -25
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r6final/gpu_tests.log 2>&1; echo rc=$? >> gpurun_out/r6final/gpu_tests.log; tail -3 gpurun_out/r6final/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python tools/rccl_world1_smoke.py > gpurun_out/r6final/rccl_world1.json 2>/dev/null; tail -c 400 gpurun_out/r6final/rccl_world1.json
