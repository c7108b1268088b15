O=gpurun_out/r06_dbg; mkdir -p $O
cat > /tmp/dbg.py <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
import nerf4k_amd
from nerf4k_amd import scene
from nerf4k_amd.lib import dvgo
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
(H, W), K = scene.LLFF_HW, scene.LLFF_K
poses = scene.llff_spiral_poses()
with torch.no_grad():
    rays = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(poses[0]).to(dev), True, False, False, False)]
out = bench.joint_train_step(ck, rays, H, W, dev)
print(json.dumps({k: out[k] for k in ('ms_per_iteration', 'ms_per_iteration_blocks', 'ms_per_iteration_per_block_graph', 'shaded_samples', 'breakdown_ms')}))
PY
python /tmp/dbg.py 2>/dev/null | tail -1
