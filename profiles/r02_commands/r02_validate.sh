#!/bin/bash
# Round-2 validation on the GPU box: the whole -m gpu suite, the default bench line, and a 2-rank run of the N>1 bench logic on ONE GPU
# (gloo + --same-device: exercises row-band sharding, the async all-gather and the tile-parallel 4K path without a second GPU).
O=gpurun_out/r2v
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests_rc=$?"
tail -6 $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench_rc=$?"
python tools/bench_summary.py $O/bench.json 2>/dev/null || tail -c 600 $O/bench.json
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r2v/bench.json'))
    print('joint_train_step', json.dumps(d.get('joint_train_step'))[:900])
    print('training_step_kernels', json.dumps(d.get('training_step_kernels'))[:600])
except Exception as e:
    print('bench parse failed', e)
PY
if [ -n "$RUN_N2" ]; then
timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 \
  --backend gloo --same-device --no-cpu-baseline --sr-frames 1 > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2_rc=$?"
tail -c 700 $O/bench_n2.json; tail -3 $O/bench_n2.err
fi
