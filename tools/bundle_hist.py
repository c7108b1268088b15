"""Distribution of the shading queue's jobs on the bench frame: records per 64-ray bundle (read from the marcher workspace after a call).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
kw = dict(seed=781, opaque=True) if 'opaque' in sys.argv else {}
ck = scene.make_llff_checkpoint(**kw)
model = utils.model_from_checkpoint_dict(ck).cuda().eval()
H, W = scene.LLFF_HW
with torch.no_grad():
    for f in (0, 7):
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(scene.llff_spiral_poses()[f]).cuda(), True, False, False, False)]
        model(ro, rd, vd, k4_img_w=W, **ck['render_kwargs'])
        torch.cuda.synchronize()
        ws = model._k4_cache()[('workspace', 0)]
        nb = ((W + 15) // 16) * ((H + 15) // 16) * 4
        off = nb * 64 * 256 * 8
        cnt = ws[off:off + nb * 4].view(torch.int32).cpu().numpy()
        bat = (cnt + 63) // 64
        print(f'frame {f}: {nb} bundles, {cnt.sum()} records; empty {np.mean(cnt == 0):.3f}, 1 batch {np.mean(bat == 1):.3f}, 2-4 {np.mean((bat >= 2) & (bat <= 4)):.3f}, '
              f'5-16 {np.mean((bat >= 5) & (bat <= 16)):.3f}, >16 {np.mean(bat > 16):.3f}; batches {bat.sum()} (ideal {cnt.sum() / 64:.0f}); max {bat.max()}')
