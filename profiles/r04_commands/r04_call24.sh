cd /root/repo
mkdir -p gpurun_out/r4c24
for v in timing0 timing; do echo "== lib4k_hip_p16$v.so (timing0 = four issuing matrix waves, timing = two producer waves)"; K4_LIB=/root/repo/4k-nerf_amd/lib4k_hip_p16$v.so timeout 600 python tools/p16_phase_timing.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done | tee gpurun_out/r4c24/p16_phase_timing_clock.log
