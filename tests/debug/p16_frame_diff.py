"""Debug: 'f16x3p' against 'f16x3' window by window (GPU box).  python tests/debug/p16_frame_diff.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nerf4k_amd  # noqa: F401
from nerf4k_amd.lib import sr_esrnet


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 200.0 if mse == 0 else -10.0 * math.log10(mse)


torch.manual_seed(777)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).cuda().eval()
g = torch.Generator().manual_seed(9)
kind = sys.argv[1] if len(sys.argv) > 1 else 'rand'
x = torch.rand([1, 3, 756, 1008], generator=g).cuda()
c = torch.rand([1, 756, 1008], generator=g).cuda()
if kind == 'smooth':
    u = torch.linspace(0, 1, 1008).view(1, -1).expand(756, 1008); v = torch.linspace(0, 1, 756).view(-1, 1).expand(756, 1008)
    x = torch.stack([0.5 + 0.5 * torch.sin(7 * u + 3 * v), u * v, 0.3 + 0.2 * torch.cos(11 * v)], 0).unsqueeze(0).cuda()
    c = (0.5 + 0.4 * torch.sin(5 * u - 2 * v)).unsqueeze(0).cuda()
with torch.no_grad():
    for grp in ('8', '1'):
        sr_esrnet.SR_GROUP = int(grp)
        net.k4_mode = 'f16x3'
        a = net.tile_process_device(x, c, 510, 10).clone()
        net.k4_mode = 'f16x3p'
        net._k4.pop('p16_reruns', None)
        b = net.tile_process_device(x, c, 510, 10).clone()
        print(f'group {grp}: frame psnr {psnr(a, b):.1f} dB  max|d| {float((a - b).abs().max()):.3e}  reruns {net._k4.get("p16_reruns", 0)}')
        for (y0, y1, x0, x1, *_r) in net.tile_geometry(756, 1008, 510, 10):
            wa, wb = a[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1], b[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]
            d = (wa - wb).abs()
            iy, ix = divmod(int(d.amax(1)[0].argmax()), d.shape[3])
            print(f'   window y {y0}-{y1} x {x0}-{x1}: {psnr(wa, wb):.1f} dB  max|d| {float(d.max()):.3e} at HR ({iy},{ix})  frac>1e-4 {float((d > 1e-4).float().mean()):.2e}')
    E = net._k4['p16']['E']; am = net._k4['p16']['amax']
    print('E range', min(E.values()), max(E.values()), ' amax range', min(am.values()), max(am.values()))
    print({k: (E[k], round(am[k], 4)) for k in list(E)[:12]})
