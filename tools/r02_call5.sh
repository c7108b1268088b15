#!/bin/bash
O=gpurun_out/r2y
mkdir -p $O
python -m pytest tests/test_sr_train_gpu.py tests/test_train_ops_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "tests_rc=$?"; tail -5 $O/tests.log
python - <<'PY' 2>/dev/null
import sys, json, torch
sys.path.insert(0, '.')
import bench, nerf4k_amd
from nerf4k_amd import scene
from nerf4k_amd.lib import dvgo
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
H, W = scene.LLFF_HW
ro, rd, vd = dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(scene.llff_spiral_poses()[0]).to(dev), True, False, False, False)
rays = tuple(x.reshape(-1, 3).contiguous() for x in (ro, rd, vd))
print(json.dumps(bench.joint_train_step(ck, rays, H, W, dev)))
PY
