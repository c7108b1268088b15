cd /root/repo
mkdir -p gpurun_out/r4c17
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "sft_epilogue or sft_epilogues or p16 or f16x3p" 2>&1 | tail -8 | tee gpurun_out/r4c17/tests.log
for f in 1 0 1 0; do echo "K4_SR_SFT_FUSE=$f"; K4_SR_SFT_FUSE=$f timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c17/frame.log
for f in 1 0; do echo "K4_SR_SFT_FUSE=$f"; K4_SR_SFT_FUSE=$f TS=168 timeout 300 python tools/sr_rank_share_time.py 2>&1 | grep rank-0; done | tee gpurun_out/r4c17/rank.log
