cd /root/repo
mkdir -p gpurun_out/r4c30
K4_TRAIN_WGRAD_STREAM=0 ITERS=12 PROFILE_HOST=1 timeout 300 python tools/joint_step_time.py 2>/dev/null > gpurun_out/r4c30/joint_host_profile.log
head -50 gpurun_out/r4c30/joint_host_profile.log | cut -c1-180
