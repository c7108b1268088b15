O=gpurun_out/r06_streams; mkdir -p $O
timeout 1500 python -m pytest tests/test_tile_parallel.py tests/test_e2e_gpu.py tests/test_sr_train_gpu.py tests/test_optim_gpu.py tests/test_rccl_gpu.py -q -m gpu 2>&1 | tail -3
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo bench_rc=$?
python tools/bench_summary.py $O/bench_default_line.json 2>/dev/null | head -12
python -c "
import json; d=json.load(open('$O/bench_default_line.json')); j=d['joint_train_step']; print({k: j[k] for k in ('ms_per_iteration','ms_per_iteration_blocks','ms_per_iteration_per_block_graph')}); print(d.get('four_k_horns',{}).get('rank_share_8gpu'))"
