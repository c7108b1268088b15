"""Marcher-only training iteration (4096-ray patch, full LLFF scene): forward + backward time, colour MLP on k4_rgbnet_* vs on the
nn.Sequential.  Usage: python tools/train_step_time.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo

dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
model = utils.model_from_checkpoint_dict(ck).to(dev).train()
H, W = scene.LLFF_HW
ro, rd, vd = dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(scene.llff_spiral_poses()[0]).to(dev), True, False, False, False)
rk = dict(ck['render_kwargs'], render_depth=True)
g = torch.Generator(device=dev).manual_seed(1)


def it(i):
    r0, c0 = (37 * i) % (H - 64), (101 * i) % (W - 64)
    rays = [x[r0:r0 + 64, c0:c0 + 64].reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    tgt = torch.rand([4096, 3], device=dev, generator=g)
    model.zero_grad(set_to_none=True)
    with torch.enable_grad():
        out = model(*rays, global_step=i, **rk)
        loss = torch.nn.functional.l1_loss(out['rgb_feature'], tgt)
        loss.backward()
    return out['weights'].numel()


for i in range(3):
    it(i)
torch.cuda.synchronize()
t = time.perf_counter()
n = 0
for i in range(10):
    n += it(3 + i)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t) / 10 * 1e3:.2f} ms per marcher train iteration (fwd+bwd), {n // 10} shaded samples")
