cd /root/repo
mkdir -p gpurun_out/r4c23
B=/root/repo/4k-nerf_amd/lib4k_hip_p16burst.so
for lib in "" "" $B $B; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1; done | tee gpurun_out/r4c23/hash.log
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "sft_epilogue or sft_epilogues or p16 or f16x3p" 2>&1 | tail -8 | tee gpurun_out/r4c23/tests.log
for lib in "" $B "" $B; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c23/frame.log
for lib in "" $B; do echo "K4_LIB=$lib"; K4_LIB=$lib TS=168 timeout 300 python tools/sr_rank_share_time.py 2>&1 | grep rank-0; done | tee gpurun_out/r4c23/rank.log
K4_LIB=/root/repo/4k-nerf_amd/lib4k_hip_p16timing.so timeout 600 python tools/p16_phase_timing.py 2>&1 | grep -v Warn | tee gpurun_out/r4c23/p16_phase_timing_producer.log
