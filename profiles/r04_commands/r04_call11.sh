cd /root/repo
mkdir -p gpurun_out/r4c11
timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "p16 or f16x3p or grouping" > gpurun_out/r4c11/tests.log 2>&1; tail -6 gpurun_out/r4c11/tests.log
timeout 200 python tools/p16_layer_time.py 6 7 10 2>&1 | grep cin | tee gpurun_out/r4c11/layers_up.log
K4_TOOL_WINDOWS=4 timeout 200 python tools/p16_layer_time.py 6 2>&1 | grep cin | tee -a gpurun_out/r4c11/layers_up.log
timeout 300 python tools/sr_frame_time.py f16x3p f16x3p f16x3 2>&1 | grep ms/frame | tee gpurun_out/r4c11/frame.log
