"""Fused SFT layer (k4_sft_nhwc_multi, bf16x6) against the module graph, error localised by tile half / channel block / pixel range.
K4_SR_DEBUG=32 selects the unpipelined kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd import _native as N
from nerf4k_amd.lib import sr_esrnet
for C in (64, 32):
    for with_res in (False, True):
        torch.manual_seed(C)
        layer = sr_esrnet.SFTLayer(C, 32).cuda()
        for p in layer.parameters():
            p.data.normal_(0, 0.3)
        n = 1000
        cond = torch.randn([n, 32]).cuda(); x = torch.randn([n, C]).cuda(); res = torch.randn([n, C]).cuda(); y = torch.zeros([n, C]).cuda()
        with torch.no_grad():
            want = layer(x.t().reshape(1, C, 1, n), cond.t().reshape(1, 32, 1, n))[0, :, 0].t()
            if with_res:
                want = want * 0.2 + res
        wp = sr_esrnet.pack_sft(layer)
        job = (N.SftJob * 1)()
        job[0].cond, job[0].x, job[0].y, job[0].res, job[0].n_pix = cond.data_ptr(), x.data_ptr(), y.data_ptr(), (res.data_ptr() if with_res else 0), n
        N.check(N.lib().k4_sft_nhwc_multi(job, 1, 32, N.f32(wp), C, C, C, 0.2, C if with_res else 0, 0.2, 1, N.stream()), 'sft_multi')
        torch.cuda.synchronize()
        d = (y - want).abs()
        print(f'C={C} res={with_res} max {float(d.max()):.3e}; first-half tiles {float(d[(torch.arange(n) % 64 < 32).cuda()].max()):.3e} second-half '
              f'{float(d[(torch.arange(n) % 64 >= 32).cuda()].max()):.3e}; per 32-channel block {[round(float(d[:, b:b + 32].max()), 6) for b in range(0, C, 32)]}; '
              f'last 40 pixels {float(d[-40:].max()):.3e}; bad pixels {int((d.max(1).values > 1e-4).sum())} first bad {(d.max(1).values > 1e-4).nonzero()[:6].flatten().tolist()}')
