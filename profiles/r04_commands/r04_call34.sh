cd /root/repo
mkdir -p gpurun_out/r4c34
OLD=/root/repo/4k-nerf_amd/lib4k_hip_p16old.so
for lib in "" $OLD "" $OLD; do echo "K4_LIB=$lib (empty = two column phases per workgroup; old = one phase per workgroup)"; K4_LIB=$lib K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 6 10 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c34/up_layers.log
for lib in "" $OLD "" $OLD; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c34/frame.log
