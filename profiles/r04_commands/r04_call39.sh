cd /root/repo
mkdir -p gpurun_out/r4c39
SEQ=/root/repo/4k-nerf_amd/lib4k_hip_p16seq.so
for lib in "" $SEQ; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1; done | tee gpurun_out/r4c39/hash.log
for lib in "" $SEQ "" $SEQ; do echo "K4_LIB=$lib (empty = both rows of a tap interleaved; seq = one row's three products back to back)"; K4_LIB=$lib K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 5 7 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c39/layers.log
for lib in "" $SEQ "" $SEQ; do echo "K4_LIB=$lib"; K4_LIB=$lib timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c39/frame.log
