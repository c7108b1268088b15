cd /root/repo
mkdir -p gpurun_out/r4c27
K4_SR_SFT_FUSE=0 K4_LIB=/root/repo/4k-nerf_amd/lib4k_hip_p16timing.so timeout 300 python tools/p16_clock_under_frame.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee gpurun_out/r4c27/clock.log
for f in 1 0 1; do echo "K4_SR_SFT_FUSE=$f"; K4_SR_SFT_FUSE=$f timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c27/frame.log
timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1 | tee gpurun_out/r4c27/hash.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4c27/tests.log
