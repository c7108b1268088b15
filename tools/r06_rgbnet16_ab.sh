#!/bin/bash
# rgbnet training kernels at 16 waves per tile (width 64): the marcher's training tests + the joint iteration, both phases
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_train_ops_gpu.py tests/test_march_gpu.py -x -q -m gpu -k "rgbnet or train or grad or joint" > $O/tests_rgbnet16.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_rgbnet16.log
for rep in 1 2; do for S0 in 0 20000; do
  echo "STEP0=$S0: $(STEP0=$S0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done | tee $O/rgbnet16.txt
STEP0=20000 timeout 300 python tools/joint_phase_events.py 2>/dev/null | tail -7 | tee -a $O/rgbnet16.txt
