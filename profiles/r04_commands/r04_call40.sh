cd /root/repo
mkdir -p gpurun_out/r4c40
R4=/root/repo/4k-nerf_amd/lib4k_hip_p16ring4.so; R6=/root/repo/4k-nerf_amd/lib4k_hip_p16ring6.so
for lib in "" $R4 $R6; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1; done | tee gpurun_out/r4c40/hash.log
for lib in "" $R4 $R6 "" $R4 $R6; do echo "K4_LIB=$lib (empty = ring 3: fragments of sub-stage u + 2 read under u; ring4 / ring6 = u + 3 / u + 5)"; K4_LIB=$lib K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 5 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c40/layers.log
