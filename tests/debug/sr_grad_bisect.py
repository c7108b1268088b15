"""Which gradients of the golden SFTNet backward (tests/golden/grad_sr.npz) differ, per parameter in network order.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch, torch.nn.functional as F
import nerf4k_amd  # noqa
from nerf4k_amd.lib import sr_esrnet
from oracle import sr as osr
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'grad_sr.npz'))
nb = int(z['num_block'])
sd = osr.make_state_dict(seed=int(z['seed']), num_block=nb)
def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=nb, num_grow_ch=32, num_cond=1)
net.load_state_dict(sd); net = net.cuda().train()
x = torch.from_numpy(z['x']).cuda().requires_grad_(True); cond = torch.from_numpy(z['cond']).cuda().requires_grad_(True)
out = net(x, cond)
F.l1_loss(out, torch.from_numpy(z['target']).cuda()).backward()
print('mode', os.environ.get('K4_TRAIN_SFT', 'fused'), 'out', rel(out, torch.from_numpy(z['out'])), 'gx', rel(x.grad, torch.from_numpy(z['grad_x'])),
      'gcond', rel(cond.grad, torch.from_numpy(z['grad_cond'])))
named = dict(net.named_parameters())
stats = z['stats']
for i, n in enumerate([str(n) for n in z['names']]):
    gn = float(named[n].grad.double().norm())
    e = abs(gn - stats[i, 1]) / (stats[i, 1] + 1e-30)
    k = 'grad/' + n
    r = rel(named[n].grad, torch.from_numpy(z[k])) if k in z.files else float('nan')
    if e > 2e-5 or r > 2e-5:
        print(f'  {n:45s} norm rel err {e:.2e}  max rel {r:.2e}')
from nerf4k_amd.lib.sr_esrnet import _Packed
cache = net._k4['train_cache']
for name in ('conv_first', 'CondNet.0', 'conv_hr'):
    m = dict(net.named_modules())[name]
    hit = cache._c.get(('b', m.weight.data_ptr()))
    fresh = _Packed.native(m.weight, None, dgrad=True)
    print(name, 'cached dgrad operand == fresh:', None if hit is None else (torch.equal(hit[1].w, fresh.w), torch.equal(hit[1].b, fresh.b), hit[1].flags_extra == fresh.flags_extra))
