"""bench.py's data-parallel joint iteration (configs[4], N > 1) on ONE GPU: `torchrun --nproc-per-node 2 tools/joint_dp_smoke.py`
(gloo, both ranks on cuda:0) -- a logic check of the N > 1 leg that the 1-GPU box cannot run over RCCL."""
import json, os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
dist.init_process_group('gloo')
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import dvgo
ck = scene.make_llff_checkpoint()
(H, W), K = scene.LLFF_HW, scene.LLFF_K
poses = scene.llff_spiral_poses()
with torch.no_grad():
    rays = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(poses[0]).to(dev), True, False, False, False)]
out = bench.joint_train_step(ck, rays, H, W, dev, 2, world, rank)
if rank == 0:
    print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
