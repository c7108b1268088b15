#!/bin/bash
# pipelined marcher rate vs the shading kernel's persistent grid size (tenths of a workgroup per CU), 60 frames each
for t in 20 15 17 13 20; do
  K4_SHADE_GRID_TENTHS=$t python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-extras --sr-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('tenths', $t, 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'iso_ms', d['roofline']['kernel_ms'])"
done
