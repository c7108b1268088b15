"""Per dense block of the golden network: gradient entering (d out) and leaving (d input) in both training graphs.  GPU box."""
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
    names = {m: n for n, m in net.named_modules()}
    cur = rec[mode] = {}
    def tap(blk, t, c, o, cur=cur, names=names):
        n = names[blk]
        cur[n + '.in'] = t.detach().clone(); cur[n + '.out'] = o.detach().clone()
        o.register_hook(lambda g, n=n: cur.__setitem__(n + '.d_out', g.detach().clone()))
        t.register_hook(lambda g, n=n: cur.__setitem__(n + '.d_in_total', g.detach().clone()))
    sr_train._TAP = tap
    x = torch.from_numpy(z['x']).cuda().requires_grad_(True); cond = torch.from_numpy(z['cond']).cuda().requires_grad_(True)
    out = net(x, cond)
    F.l1_loss(out, torch.from_numpy(z['target']).cuda()).backward()
for k in sorted(rec['convs']):
    a, b = rec['convs'][k], rec['fused'][k]
    d = (a - b).abs()
    bad = d.sum(2) > 1e-5 * float(a.abs().max())
    print(f'{k:28s} max diff {float(d.max()):.3e} of {float(a.abs().max()):.3e}   bad pixels {int(bad.sum())}')
    if int(bad.sum()) and int(bad.sum()) < 60:
        ys, xs = torch.nonzero(bad, as_tuple=True)
        print('      rows', int(ys.min()), int(ys.max()), 'cols', int(xs.min()), int(xs.max()))
