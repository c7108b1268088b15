"""Where do the marcher variants differ on the full LLFF frame?  (GPU box; debugging aid of round 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
rk = dict(ck['render_kwargs'], render_depth=True)
H, W = scene.LLFF_HW
poses = scene.llff_spiral_poses()
def run(pre, brick, frame=3):
    dvgo._MARCH_PRE, dvgo._K0_BRICK = pre, brick
    with torch.no_grad():
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(poses[frame]).to(dev), True, False, False, False)]
        out = model(ro, rd, vd, k4_img_w=W, **rk)
        torch.cuda.synchronize()
    return {k: out[k].clone() for k in ('rgb_marched', 'depth', 'alphainv_last')}
def diff(a, b, name):
    for k in a:
        d = (a[k] - b[k]).abs()
        if d.dim() > 1: d = d.amax(-1)
        n = int((d > 0).sum())
        if n:
            idx = (d > 0).nonzero().squeeze(1)
            ys, xs = (idx // W).cpu().numpy(), (idx % W).cpu().numpy()
            print(f'{name}: {k}: {n} rays differ, max {float(d.max()):.3e}; first rays (y,x): {list(zip(ys[:12], xs[:12]))}; tiles(8x8) touched {len(set(zip(ys // 8, xs // 8)))}')
        else:
            print(f'{name}: {k}: identical')
v = {}
for name, (pre, brick) in dict(a0=(False, False), a1=(False, False), b0=(True, False), b1=(True, False), c0=(False, True), c1=(False, True), d0=(True, True)).items():
    v[name] = run(pre, brick)
diff(v['a0'], v['a1'], 'old/cl run-to-run')
diff(v['b0'], v['b1'], 'pre/cl run-to-run')
diff(v['c0'], v['c1'], 'old/brick run-to-run')
diff(v['a0'], v['b0'], 'old/cl vs pre/cl')
diff(v['a0'], v['c0'], 'old/cl vs old/brick')
diff(v['b0'], v['d0'], 'pre/cl vs pre/brick')
diff(v['c0'], v['d0'], 'old/brick vs pre/brick')
