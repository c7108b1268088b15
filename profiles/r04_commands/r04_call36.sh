cd /root/repo
mkdir -p gpurun_out/r4c36
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r4c36/tests.log
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); fk = d['four_k']
        print({k: fk[k] for k in ('ms_per_frame', 'ms_per_frame_median', 'ms_per_frame_p90', 'ms_per_frame_one_at_a_time', 'frames_per_s')}, fk['sr_roofline']['frac'])
        print({k: v.get('ms_per_frame') for k, v in d.items() if k.startswith('four_k_')})" | tee gpurun_out/r4c36/bench.log
