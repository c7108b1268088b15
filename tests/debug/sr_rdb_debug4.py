"""LeakyReLU masks of the decoder's 4x path in both training graphs: elements whose sign differs, and their magnitudes.  GPU box."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, torch.nn.functional as F
import nerf4k_amd  # noqa
from nerf4k_amd.lib import sr_esrnet, sr_train
from oracle import sr as osr
z = np.load(os.path.join(R, 'tests', 'golden', 'grad_sr.npz'))
nb = int(z['num_block'])
sd = osr.make_state_dict(seed=int(z['seed']), num_block=nb)
rec = {}
for mode in ('convs', 'fused'):
    os.environ['K4_TRAIN_SFT'] = mode
    net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=nb, num_grow_ch=32, num_cond=1)
    net.load_state_dict(sd); net = net.cuda().train()
    cur = rec[mode] = {}
    def tap(blk, t, c, o, cur=cur):
        if isinstance(blk, str):
            cur[blk] = o.detach().clone()
    sr_train._TAP = tap
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True); cond = torch.from_numpy(z['cond']).cuda().requires_grad_(True)
    out = net(x, cond)
for k in rec['convs']:
    a, b = rec['convs'][k], rec['fused'][k]
    flip = (a > 0) != (b > 0)
    idx = torch.nonzero(flip)
    print(k, tuple(a.shape), 'mask flips', int(flip.sum()), [(tuple(int(v) for v in i), float(a[tuple(i)]), float(b[tuple(i)])) for i in idx[:6]],
          'max |diff|', float((a - b).abs().max()))
