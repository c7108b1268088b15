cd /root/repo
mkdir -p gpurun_out/r4c31
for g in 2 1 2 1; do echo "K4_SHADE_GRID_WG=$g"; K4_SHADE_GRID_WG=$g timeout 300 python bench.py --no-cpu-baseline --no-extras --sr-frames 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'isolated ms', d['roofline']['kernel_ms'])"; done | tee gpurun_out/r4c31/shade_grid.log
timeout 600 python -m pytest tests/test_sr_gpu.py -x -q -k "deterministic" 2>&1 | tail -2 | tee -a gpurun_out/r4c31/shade_grid.log
