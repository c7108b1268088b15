cd /root/repo
mkdir -p gpurun_out/r4c38
timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1 | tee gpurun_out/r4c38/hash.log
K4_SR_DEBUG=1024 timeout 300 python tools/sr_frame_hash.py f16x3p 2>&1 | grep sha1 | tee -a gpurun_out/r4c38/hash.log
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "p16 or f16x3p or grouping or deterministic" 2>&1 | tail -3 | tee gpurun_out/r4c38/tests.log
for d in 0 1024 0 1024; do echo "K4_SR_DEBUG=$d (1024 = one workgroup per tile)"; K4_SR_DEBUG=$d K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 5 7 2>&1 | grep "^cin" | sed 's/f16x3 per-tile.*| p16 in, p16 out/| p16 in, p16 out/'; done | tee gpurun_out/r4c38/layers.log
for d in 0 1024 0 1024; do echo "K4_SR_DEBUG=$d"; K4_SR_DEBUG=$d timeout 300 python tools/sr_frame_time.py f16x3p f16x3p 2>&1 | grep ms/frame; done | tee gpurun_out/r4c38/frame.log
