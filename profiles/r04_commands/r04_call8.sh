cd /root/repo
mkdir -p gpurun_out/r4c8
for g in 1 2; do K4_SR_GROUP=$g timeout 200 python tools/sr_frame_time.py f16x3p f16x3p > gpurun_out/r4c8/frame_group$g.log 2>&1; echo "group $g"; grep -v amdgpu gpurun_out/r4c8/frame_group$g.log; done
bash tools/r04_final_prof.sh $1 2>&1 | tail -40
