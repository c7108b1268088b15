#!/bin/bash
# A/B on one box: conv1's weight gradient of every dense block on the third stream against all five on the weight gradients' stream
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -x -q -m gpu > $O/tests_auxw.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_auxw.log
for rep in 1 2; do for a in 1 0; do for S0 in 0 20000; do
  echo "aux_wgrad=$a STEP0=$S0: $(TOOL_AUX_WGRAD=$a STEP0=$S0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done; done | tee $O/auxw.txt
