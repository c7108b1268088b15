cd /root/repo
mkdir -p gpurun_out/r4c12
timeout 900 python -m pytest tests/test_sr_gpu.py tests/test_e2e_gpu.py -x -q > gpurun_out/r4c12/tests.log 2>&1; tail -4 gpurun_out/r4c12/tests.log
timeout 300 python tools/sr_frame_time.py f16x3p f16x3p f16x3 2>&1 | grep ms/frame | tee gpurun_out/r4c12/frame.log
