#!/bin/bash
# k0's optimizer step of the dense-TV iterations in two exact parts: tests + A/B of the joint iteration (first 10,000 iterations) on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_optim_gpu.py tests/test_train_ops_gpu.py -x -q -m gpu > $O/tests_splitstep.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests_splitstep.log
for rep in 1 2 3; do for a in 1 0; do
  echo "split_step=$a STEP0=0: $(TOOL_SPLIT_STEP=$a STEP0=0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done | tee $O/splitstep.txt
for a in 1 0; do echo "split_step=$a"; TOOL_SPLIT_STEP=$a STEP0=0 timeout 300 python tools/joint_phase_events.py 2>/dev/null | tail -7; done | tee -a $O/splitstep.txt
