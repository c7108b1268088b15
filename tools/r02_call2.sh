#!/bin/bash
O=gpurun_out/r2w
mkdir -p $O
python -m pytest tests/test_train_ops_gpu.py tests/test_sr_gpu.py tests/test_e2e_gpu.py tests/test_sr_train_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "tests_rc=$?"
tail -12 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench_rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r2w/bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'iso', d['mrays_isolated'], 'frac', d['roofline']['frac'])
for k in ('four_k', 'four_k_fp32mfma', 'four_k_bf16x3', 'joint_train_step'):
    v = d.get(k) or {}
    print(k, {q: v.get(q) for q in ('ms_per_frame', 'frames_per_s', 'effective_tflops', 'psnr_vs_oracle_db', 'ms_per_iteration', 'breakdown_ms')})
PY
