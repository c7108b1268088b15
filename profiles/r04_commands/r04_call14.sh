cd /root/repo
mkdir -p gpurun_out/r4c14
for b in 1 2 4 8 1 4; do K4_MARCH_BANDS=$b timeout 200 python tools/march_call_time.py 2>&1 | grep K4_MARCH; done | tee gpurun_out/r4c14/bands.log
