cd /root/repo
mkdir -p gpurun_out/r4c2
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "p16 or f16x3p" > gpurun_out/r4c2/tests.log 2>&1; tail -25 gpurun_out/r4c2/tests.log
timeout 300 python tools/p16_layer_time.py > gpurun_out/r4c2/layers1.log 2>&1; cat gpurun_out/r4c2/layers1.log
K4_TOOL_WINDOWS=4 timeout 300 python tools/p16_layer_time.py 0 3 4 5 > gpurun_out/r4c2/layers4.log 2>&1; cat gpurun_out/r4c2/layers4.log
timeout 300 python tools/sr_frame_time.py f16x3 f16x3p f16x3 f16x3p > gpurun_out/r4c2/frame.log 2>&1; cat gpurun_out/r4c2/frame.log
