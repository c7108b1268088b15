"""Phase breakdown of the shading kernel (k4_shade_kernel) on the bench frame: run with K4_LIB=<library built with -DK4_SHADE_TIMING>
(profiles/r03_commands/r03_call4.sh).  The kernel adds per-wave s_memtime sums to out_counters[8..15]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf4k_amd
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
ck = scene.make_llff_checkpoint()
model = utils.model_from_checkpoint_dict(ck).cuda().eval()
H, W = scene.LLFF_HW
names = ['record unpack + corner setup', '24 corner fetches + interpolation', 'PE / viewdir features -> LDS', 'layer 1 (split + 12 MFMA per tile)',
         'layer 2 (split + 48 MFMA per tile)', 'output layer + shuffles', 'sigmoid + blend + per-ray sums', 'between batches (ticket, ray setup, outputs)']
with torch.no_grad():
    for f in (0, 7):
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(scene.llff_spiral_poses()[f]).cuda(), True, False, False, False)]
        cnt = torch.zeros(24, dtype=torch.int64, device='cuda'); cnt[20] = 2 ** 62
        model(ro, rd, vd, k4_img_w=W, k4_counters=cnt, **ck['render_kwargs'])
        torch.cuda.synchronize()
        c = cnt.cpu().numpy().astype(np.float64)
        nb = c[3] / 64.0
        tot = c[8:16].sum()
        print(f'frame {f}: shaded {int(c[3])} records ~ {nb:.0f} batches; s_memtime ticks summed over waves {tot:.3e} = {tot / max(nb, 1):.0f} per batch (100 MHz ticks x ? -- relative shares matter)')
        if c[18] > 0:
            print(f'  waves {int(c[18])}: mean lifetime {c[16] / c[18]:.0f} ticks, longest {c[17]:.0f} ticks (mean / longest = {c[16] / c[18] / c[17]:.2f}); phase ticks / lifetime ticks = {tot / c[16]:.2f}')
        if c[18] > 0 and c[21] > 0:
            r0, r1, nw = float(cnt[20]), float(cnt[21]), c[18]
            ms, me = float(cnt[22]) / nw, float(cnt[23]) / nw
            print(f'  split-path shading kernel: span {(r1 - r0) / 100:.1f} us (100 MHz clock); mean wave start +{(ms - r0) / 100:.1f} us, mean wave exit +{(me - r0) / 100:.1f} us; '
                  f'weight staging {c[19] / nw:.0f} memtime ticks per wave; memtime ticks per us ~ {c[17] / max((r1 - r0) / 100, 1e-9):.0f}')
        for i, n in enumerate(names):
            print(f'  {n:48s} {100 * c[8 + i] / tot:5.1f} %   {c[8 + i] / max(nb, 1):8.1f} ticks/batch')
