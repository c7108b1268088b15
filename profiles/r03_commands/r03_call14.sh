#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench_rc=$?"
python tools/bench_summary.py $O/bench.json 2>/dev/null || tail -c 600 $O/bench.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3i/bench.json'))
for k in ('four_k_bf16x6', 'four_k_bf16x3', 'four_k_fp32mfma', 'joint_train_step', 'training_step_kernels'):
    print(k, json.dumps(d.get(k))[:600])
PY
tail -3 $O/bench.err
