cd /root/repo
mkdir -p gpurun_out/r4c9
for ts in 168 189; do for s in 1 2 4; do echo "TS=$ts streams=$s"; TS=$ts K4_SR_STREAMS=$s timeout 120 python tools/sr_rank_share_time.py 2>&1 | grep rank; done; done | tee gpurun_out/r4c9/rank_share_streams.log
for s in 1 2; do echo "TS=168 streams=$s mode f16x3"; K4_SR_MODE=f16x3 TS=168 K4_SR_STREAMS=$s timeout 120 python tools/sr_rank_share_time.py 2>&1 | grep rank; done | tee -a gpurun_out/r4c9/rank_share_streams.log
