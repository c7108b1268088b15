#!/bin/bash
# Round 3, GPU call 2: f16x3 decoder arithmetic -- accuracy tests, frame times per arithmetic and tile height, 3 workgroups per CU variant,
# kernel stats; marcher tests again (live-mask adversarial scene fixed).  Writes gpurun_out/r3b/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_march_gpu.py -m gpu -x -q -k "live_mask" > $O/tests_live.log 2>&1; echo "live_tests_rc=$?"; tail -3 $O/tests_live.log
timeout 900 python -m pytest tests/test_sr_gpu.py -m gpu -q -s > $O/tests_sr.log 2>&1; echo "sr_tests_rc=$?"
grep -E "PSNR|dB|max err|passed|failed|Error|assert" $O/tests_sr.log | head -40
python tools/sr_frame_time.py bf16x6 f16x3 bf16x3 f16x3 bf16x6 2>&1 | grep ms/frame
for r in 2 3 4; do echo "K4_SR_2T_RPW=$r"; K4_SR_2T_RPW=$r python tools/sr_frame_time.py f16x3 2>&1 | grep ms/frame; done
echo "3 workgroups per CU variant (RPW 2)"; K4_LIB=$R/4k-nerf_amd/lib4k_hip_f16wg3.so python tools/sr_frame_time.py f16x3 f16x3 2>&1 | grep ms/frame
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tmp -o run -- python $R/tools/sr_frame_time.py f16x3 > $O/sr_f16_stats.log 2>&1
f=$(find $R/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1); head -14 "$f" > $O/sr_f16x3_kernel_stats.csv; cut -c1-150 $O/sr_f16x3_kernel_stats.csv
rm -rf $R/gpurun_out/prof_tmp
cd $R
timeout 600 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -s -k "full_4k_frame or tile_parallel_hip" > $O/tests_e2e.log 2>&1; echo "e2e_rc=$?"; grep -E "PSNR|passed|failed" $O/tests_e2e.log | tail -5
