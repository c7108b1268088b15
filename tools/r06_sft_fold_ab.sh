#!/bin/bash
# tests of the decoder's training path + joint iteration time of the tree (both phases), two repetitions
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py tests/test_abi.py -x -q -m gpu > $O/tests_fold.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_fold.log
for rep in 1 2; do for S0 in 0 20000; do
  echo "STEP0=$S0: $(STEP0=$S0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done | tee $O/fold.txt
