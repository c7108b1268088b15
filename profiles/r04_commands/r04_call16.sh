cd /root/repo
mkdir -p gpurun_out/r4c16
for lib in "" 4k-nerf_amd/lib4k_hip_w6.so "" 4k-nerf_amd/lib4k_hip_w6.so; do echo "K4_LIB=$lib"; K4_LIB=${lib:+/root/repo/$lib} K4_MARCH_BANDS=x timeout 200 python tools/march_call_time.py 2>&1 | grep K4_MARCH; done | tee gpurun_out/r4c16/k2_w6.log
K4_LIB=/root/repo/4k-nerf_amd/lib4k_hip_w6.so timeout 600 python -m pytest tests/test_march_gpu.py -x -q 2>&1 | tail -3 | tee -a gpurun_out/r4c16/k2_w6.log
