"""Isolated marcher call on the bench's LLFF frame (stream-synchronised, HIP events, median of 20 after 3 warm-ups) + a hash of its outputs:
`K4_MARCH_BANDS=1|2|4|8 python tools/march_call_time.py` -- the hash must not depend on the banding.  GPU box."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
h = hashlib.sha1()
ms = []
with torch.no_grad():
    for i in range(23):
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(poses[i % 20]).to(dev), True, False, False, False)]
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = model(ro, rd, vd, k4_img_w=W, **rk); b.record()
        torch.cuda.synchronize()
        if i >= 3:
            ms.append(a.elapsed_time(b))
        if i < 4:
            for k in ('rgb_marched', 'depth', 'alphainv_last'):
                h.update(out[k].cpu().numpy().tobytes())
m = float(np.median(ms))
print(f"K4_MARCH_BANDS={os.environ.get('K4_MARCH_BANDS', 'default')}: isolated call median {m:.4f} ms = {H * W / m / 1e3:.1f} Mrays/s (p10 {np.percentile(ms, 10):.4f}, p90 {np.percentile(ms, 90):.4f}); outputs sha1 {h.hexdigest()[:16]}")
