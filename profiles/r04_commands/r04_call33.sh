cd /root/repo
mkdir -p gpurun_out/r4c33
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "p16 or f16x3p or grouping" 2>&1 | tail -3 | tee gpurun_out/r4c33/tests.log
K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 6 10 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/' | tee gpurun_out/r4c33/up_layers.log
for i in 1 2; do timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c33/frame.log
timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1 | tee gpurun_out/r4c33/hash.log
