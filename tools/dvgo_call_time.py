"""configs[0] at 800x800 (DirectVoxGO 160^3, rgbnet 39->128->128->3): isolated fused call per K4_MLP value -- `K4_MLP=fp32 python tools/dvgo_call_time.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
dev = torch.device('cuda', 0)
ck = scene.make_lego_checkpoint()
model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
rk = ck['render_kwargs']
H = 800
K = scene.lego_K(H, H)
pose = scene.lego_pose()
with torch.no_grad():
    ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in
                  dvgo.get_rays_of_a_view(H, H, K, torch.from_numpy(pose[:3, :4].astype(np.float32)).to(dev), False, False, False, False)]
    ms = []
    for i in range(12):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = model(ro, rd, vd, k4_img_w=H, **rk); b.record()
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(a.elapsed_time(b))
print(f"K4_MLP={os.environ.get('K4_MLP', 'default(b2)')}: DVGO 800x800 isolated call median {np.median(ms):.3f} ms = {H * H / np.median(ms) / 1e3:.1f} Mrays/s; mean rgb {float(out['rgb_marched'].mean()):.6f}")
