cd /root/repo
mkdir -p gpurun_out/r4c46
ITERS=12 PROFILE_HOST=1 timeout 300 python tools/joint_step_time.py 2>/dev/null > gpurun_out/r4c46/joint_host_profile.log
head -44 gpurun_out/r4c46/joint_host_profile.log | cut -c1-150
