cd /root/repo
mkdir -p gpurun_out/r4c7
K4_P16_V2=1 timeout 900 python -m pytest tests/test_sr_gpu.py -x -q -k "p16 or f16x3p or grouping" > gpurun_out/r4c7/tests_v2.log 2>&1; tail -6 gpurun_out/r4c7/tests_v2.log
K4_P16_V2=1 K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 200 python tools/p16_layer_time.py 0 3 4 5 > gpurun_out/r4c7/layers4_v2.log 2>&1; grep -v amdgpu gpurun_out/r4c7/layers4_v2.log
K4_P16_V2=0 K4_TOOL_ONLY=p16 K4_TOOL_WINDOWS=4 timeout 200 python tools/p16_layer_time.py 0 3 4 5 > gpurun_out/r4c7/layers4_v1.log 2>&1; grep -v amdgpu gpurun_out/r4c7/layers4_v1.log
K4_P16_V2=1 K4_TOOL_ONLY=p16 timeout 200 python tools/p16_layer_time.py 3 6 7 8 9 > gpurun_out/r4c7/layers1_v2.log 2>&1; grep -v amdgpu gpurun_out/r4c7/layers1_v2.log
K4_P16_V2=0 K4_TOOL_ONLY=p16 timeout 200 python tools/p16_layer_time.py 3 6 7 8 9 > gpurun_out/r4c7/layers1_v1.log 2>&1; grep -v amdgpu gpurun_out/r4c7/layers1_v1.log
K4_P16_V2=1 timeout 300 python tools/sr_frame_time.py f16x3p f16x3p > gpurun_out/r4c7/frame_v2.log 2>&1; grep -v amdgpu gpurun_out/r4c7/frame_v2.log
K4_P16_V2=0 timeout 300 python tools/sr_frame_time.py f16x3p f16x3p > gpurun_out/r4c7/frame_v1.log 2>&1; grep -v amdgpu gpurun_out/r4c7/frame_v1.log
