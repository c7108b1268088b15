cd /root/repo
mkdir -p gpurun_out/r4c42
for d in 0 4096 0 4096; do echo "K4_SR_DEBUG=$d (4096: the workgroup in odd wave slots at s_setprio 3)"; K4_SR_DEBUG=$d K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 5 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c42/prio.log
for d in 0 4096 0 4096; do echo "K4_SR_DEBUG=$d"; K4_SR_DEBUG=$d timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee -a gpurun_out/r4c42/prio.log
