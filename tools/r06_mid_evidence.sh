O=gpurun_out/r06_mid; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/joint_dp_smoke.py > $O/joint_dp_smoke_2rank_gloo.json 2> $O/joint_dp_smoke.err; echo dp_rc=$?
tail -c 1500 $O/joint_dp_smoke_2rank_gloo.json
timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_tile_parallel.py -q -m gpu 2>&1 | tail -3
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo bench_rc=$?
python tools/bench_summary.py $O/bench_default_line.json 2>/dev/null | head -40 || head -c 3000 $O/bench_default_line.json
