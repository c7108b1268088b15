"""Run-to-run determinism of the marcher on the full LLFF frame, for the package tree given as argv[1] (default: this repo) under the
current environment.  GPU box; debugging aid of round 5."""
import os, sys
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
rk = dict(ck['render_kwargs'], render_depth=True)
H, W = scene.LLFF_HW
poses = scene.llff_spiral_poses()
def run(frame=3):
    with torch.no_grad():
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(poses[frame]).to(dev), True, False, False, False)]
        out = model(ro, rd, vd, k4_img_w=W, **rk)
        torch.cuda.synchronize()
    return out['rgb_marched'].clone()
ref = run()
bad = []
for i in range(6):
    d = (run() - ref).abs().amax(-1)
    bad.append((int((d > 0).sum()), float(d.max())))
print(f'{root} K4_LIB={os.environ.get("K4_LIB", "-")}: 6 repeats vs the first run: (rays differing, max abs) = {bad}')
