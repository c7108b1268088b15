cd /root/repo
mkdir -p gpurun_out/r4c21
B=/root/repo/4k-nerf_amd/lib4k_hip_p16burst.so
for lib in "" "" $B $B; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1; done | tee gpurun_out/r4c21/hash.log
for lib in "" $B; do echo "K4_LIB=$lib fuse=0"; K4_SR_SFT_FUSE=0 K4_LIB=$lib timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1; done | tee -a gpurun_out/r4c21/hash.log
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "sft_epilogue or sft_epilogues or p16 or f16x3p" 2>&1 | tail -8 | tee gpurun_out/r4c21/tests.log
