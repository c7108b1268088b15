"""sha1 of the fused marcher's outputs on a mid-size synthetic LLFF frame, under the CURRENT environment (the library reads its K4_* knobs
once while it loads, so variants -- K4_MLP, K4_GEOM_SKIP, K4_LIB builds -- are compared across processes)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
dev = torch.device('cuda', 0)
# K4_HASH_SCENE: mpi64 (default: the LLFF shape), mpi32 (rgbnet width 32), mpi_d2 (no hidden layer), dvgo64 (bounded scene, 12 k0 channels + view direction)
which = os.environ.get('K4_HASH_SCENE', 'mpi64')
h = hashlib.sha1()
with torch.no_grad():
    if which.startswith('mpi'):
        kw = {'mpi64': {}, 'mpi32': dict(rgbnet_width=32), 'mpi_d2': dict(rgbnet_depth=2)}[which]
        ck = scene.make_llff_checkpoint(seed=61, num_voxels=128 * 128 * 96, mpi_depth=96, n_blobs=40, **kw)
        model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
        rk = dict(ck['render_kwargs'], render_depth=True)
        H, W = 378, 504
        K = scene.LLFF_K.copy()
        K[:2] *= W / scene.LLFF_HW[1]
        views = [dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(scene.llff_spiral_poses()[f]).to(dev), True, False, False, False) for f in (2, 11)]
    else:
        import numpy as np
        ck = scene.make_lego_checkpoint(seed=55, num_voxels=96 ** 3, rgbnet_dim=12, rgbnet_width=64, viewbase_pe=0)
        model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
        rk = dict(ck['render_kwargs'], render_depth=True)
        H = W = 200
        K = scene.lego_K(H, W)
        views = [dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(scene.lego_pose(theta_deg=t)[:3, :4].astype(np.float32)).to(dev), False, False, False, False) for t in (30., 140.)]
    for v in views:
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in v]
        out = model(ro, rd, vd, k4_img_w=W, **rk)
        torch.cuda.synchronize()
        for k in ('rgb_marched', 'depth', 'alphainv_last'):
            h.update(out[k].cpu().numpy().tobytes())
        assert float(out['rgb_marched'].abs().sum()) > 0
print('MARCH_HASH', which, h.hexdigest())
