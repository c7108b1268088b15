#!/bin/bash
# A/B on one box: the decoder's side streams (weight gradients, SFT rest) at the lowest stream priority against the default.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sftsplit; mkdir -p $O; cd $R
for rep in 1 2; do for low in 0 1; do for S0 in 20000 0; do
  echo "low=$low STEP0=$S0: $(TOOL_SIDE_LOW=$low STEP0=$S0 BLOCKS=6 timeout 300 python tools/joint_step_time.py 2>/dev/null | tail -1 | cut -c1-150)"
done; done; done | tee $O/prio_ab.txt
