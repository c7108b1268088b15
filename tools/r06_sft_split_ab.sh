#!/bin/bash
# A/B on one box: the SFT layers' backward as gx (chain) + rest (third stream) against the one-launch form; both phases of fern_lg_joint_l1.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py tests/test_abi.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for s in 1 0; do for S0 in 0 20000; do
  echo "split=$s STEP0=$S0: $(TOOL_SFT_SPLIT=$s STEP0=$S0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done; done | tee $O/ab.txt
for s in 1 0; do for S0 in 0 20000; do echo "split=$s STEP0=$S0"; TOOL_SFT_SPLIT=$s STEP0=$S0 timeout 300 python tools/joint_phase_events.py 2>/dev/null | tail -8; done; done | tee $O/phases.txt
