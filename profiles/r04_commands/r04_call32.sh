cd /root/repo
mkdir -p gpurun_out/r4c32
for g in 2 1; do for s in 3 4 6; do echo "K4_SHADE_GRID_WG=$g streams=$s"; K4_SHADE_GRID_WG=$g timeout 300 python bench.py --no-cpu-baseline --no-extras --sr-frames 0 --streams $s 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'median', d.get('ms_per_step_median'))"; done; done | tee gpurun_out/r4c32/streams.log
