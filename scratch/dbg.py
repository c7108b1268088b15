import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import nerf4k_amd
from nerf4k_amd import scene, render
from nerf4k_amd.lib import utils
from oracle import marcher
cfg = dict(seed=31, num_voxels=64 * 64 * 48, mpi_depth=48)
ck = scene.make_llff_checkpoint(**cfg)
model = utils.model_from_checkpoint_dict(ck).cuda().eval()
H, W = 90, 120
K = scene.LLFF_K.copy(); K[:2] *= W / scene.LLFF_HW[1]
pose = scene.llff_spiral_poses()[7]
rays = marcher.get_rays_of_a_view(H, W, K, pose, ndc=True)
ro, rd, vd = [x.reshape(-1, 3) for x in rays]
want = marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], ro, rd, vd, **ck['render_kwargs'])
with torch.no_grad():
    res = render.render_frame(model, H, W, K, pose, True, dict(ck['render_kwargs']), rays=[x.cuda() for x in rays])
    lin = model(ro.cuda(), rd.cuda(), vd.cuda(), **ck['render_kwargs'])
    lin2 = model(ro.cuda(), rd.cuda(), vd.cuda(), **ck['render_kwargs'])
a = res['alphainv_last'].reshape(-1).cpu(); b = lin['alphainv_last'].cpu(); c = want['alphainv_last']; b2 = lin2['alphainv_last'].cpu()
print('res vs lin mismatches', int((a != b).sum()), 'max', float((a - b).abs().max()))
print('lin vs lin2 mismatches', int((b != b2).sum()))
print('res vs oracle max', float((a - c).abs().max()), 'n>1e-6', int(((a - c).abs() > 1e-6).sum()))
print('lin vs oracle max', float((b - c).abs().max()), 'n>1e-6', int(((b - c).abs() > 1e-6).sum()))
idx = torch.nonzero(a != b).flatten()[:10]
for i in idx.tolist():
    print(i, i // W, i % W, float(a[i]), float(b[i]), float(c[i]))
