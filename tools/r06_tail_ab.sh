#!/bin/bash
# A/B on one box: the tail of the decoder's backward pass (conv_first's weight gradient behind the last block's fork, the CondNet's dealt to both side streams)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -x -q -m gpu > $O/tests_tail.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_tail.log
for rep in 1 2 3; do for a in 1 0; do for S0 in 0 20000; do
  echo "tail_split=$a STEP0=$S0: $(TOOL_TAIL_SPLIT=$a STEP0=$S0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done; done | tee $O/tail.txt
