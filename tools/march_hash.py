"""sha1 of the fused marcher's outputs on a mid-size synthetic LLFF frame, under the CURRENT environment (the library reads its K4_* knobs
once while it loads, so variants -- K4_MLP, K4_GEOM_SKIP, K4_LIB builds -- are compared across processes)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint(seed=61, num_voxels=128 * 128 * 96, mpi_depth=96, n_blobs=40)
model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
rk = dict(ck['render_kwargs'], render_depth=True)
H, W = 378, 504
K = scene.LLFF_K.copy()
K[:2] *= W / scene.LLFF_HW[1]
h = hashlib.sha1()
with torch.no_grad():
    for f in (2, 11):
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(scene.llff_spiral_poses()[f]).to(dev), True, False, False, False)]
        out = model(ro, rd, vd, k4_img_w=W, **rk)
        torch.cuda.synchronize()
        for k in ('rgb_marched', 'depth', 'alphainv_last'):
            h.update(out[k].cpu().numpy().tobytes())
        assert float(out['rgb_marched'].abs().sum()) > 0
print('MARCH_HASH', h.hexdigest())
