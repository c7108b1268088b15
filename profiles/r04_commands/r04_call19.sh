cd /root/repo
mkdir -p gpurun_out/r4c19
K4_LIB=/root/repo/4k-nerf_amd/lib4k_hip_p16timing.so timeout 600 python tools/p16_phase_timing.py 2>&1 | grep -v Warn | tee gpurun_out/r4c19/p16_phase_timing.log
