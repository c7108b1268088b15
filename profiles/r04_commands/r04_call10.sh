cd /root/repo
mkdir -p gpurun_out/r4c10
for dbg in 0 1024 0 1024; do echo "K4_SR_DEBUG=$dbg"; K4_SR_DEBUG=$dbg K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 120 python tools/p16_layer_time.py 0 3 4 2>&1 | grep cin | sed 's/f16x3 per-tile.*| p16 in, p16 out/p16out/'; done | tee gpurun_out/r4c10/setprio.log
for dbg in 0 1024; do echo "K4_SR_DEBUG=$dbg"; K4_SR_DEBUG=$dbg timeout 200 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee -a gpurun_out/r4c10/setprio.log
