"""Stage breakdown of the geometry kernel (k4_geom3_kernel) on the bench frame: run with K4_LIB=<library built with -DK4_GEOM_TIMING>
(bash tools/build_variant.sh geomtiming -DK4_GEOM_TIMING).  The kernel adds per-wave s_memtime sums to out_counters[8..16]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
kw = dict(seed=781, opaque=True) if 'opaque' in sys.argv else {}
ck = scene.make_llff_checkpoint(**kw)
model = utils.model_from_checkpoint_dict(ck).cuda().eval()
H, W = scene.LLFF_HW
names = ['prologue + ray setup', 'stage P: probe of the skip groups', 'entry list', 'stage A: indices + byte fetch issue', 'stage A: wait bytes, ballots, ring',
         'stage B: retire / issue density batch', 'drain', 'arrival + (last wave) scan C + compaction D']
with torch.no_grad():
    for f in (0, 7):
        ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(scene.llff_spiral_poses()[f]).cuda(), True, False, False, False)]
        cnt = torch.zeros(24, dtype=torch.int64, device='cuda')
        # NOTE: with a counter buffer the product library runs the COUNTING instantiation; the timing build runs the render instantiation on the plain mask
        model(ro, rd, vd, k4_img_w=W, k4_counters=cnt, k4_live_mask='force', **ck['render_kwargs'])
        torch.cuda.synchronize()
        c = cnt.cpu().numpy().astype(np.float64)
        tot, nw = c[8:16].sum(), c[16]
        print(f'frame {f}: {int(nw)} waves, mean life {tot / max(nw, 1):.0f} s_memtime ticks')
        for i, n in enumerate(names):
            print(f'  {n:48s} {100 * c[8 + i] / max(tot, 1):5.1f} %')
