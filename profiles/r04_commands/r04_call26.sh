cd /root/repo
mkdir -p gpurun_out/r4c26
for lib in "" 4k-nerf_amd/lib4k_hip_p16prod1.so 4k-nerf_amd/lib4k_hip_p16burst.so; do echo "== K4_LIB=$lib (default build: two producer waves)"; K4_LIB=${lib:+/root/repo/$lib} K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 5 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c26/layers.log
