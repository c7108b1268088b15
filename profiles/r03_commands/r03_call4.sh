#!/bin/bash
# Round 3, GPU call 4: two output blocks per workgroup in the fp16 3x3 convolution (K4_SR_NBK A/B, tests), phase timing of the shading kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_sr_gpu.py -m gpu -q -s > $O/tests_sr.log 2>&1; echo "sr_tests_rc=$?"
grep -E "PSNR|dB|max err|passed|failed|Error|assert" $O/tests_sr.log | head -30
for nbk in 2 1 2 1; do echo "K4_SR_NBK=$nbk"; K4_SR_NBK=$nbk python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; done
for r in 3; do echo "K4_SR_2T_RPW=$r (NBK 2)"; K4_SR_2T_RPW=$r python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame; done
K4_LIB=$R/4k-nerf_amd/lib4k_hip_timing.so python tools/shade_timing.py 2>&1 | grep -v Warning | tail -22
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/sr_frame_time.py f16x3 > $O/sr_trace.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $O/sr_layer_table.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k4_' in r['Kernel_Name']]
# the last frame's launches: take the final third
n = len(rows) // 3
last = rows[-n:]
agg = collections.OrderedDict()
for r in last:
    k = (r['Kernel_Name'].split('(')[0], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print(f'one frame: {len(last)} launches, {tot / 1e3:.2f} ms of kernel time')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{v[1]:9.1f} us  x{v[0]:3d}  avg {v[1] / v[0]:8.1f}  grid {k[1]:>9s}  {k[0][:70]}')
PY
cat $O/sr_layer_table.txt | head -30
rm -rf $R/gpurun_out/prof_tmp
