"""Where an UNTRACED joint training iteration spends its time: HIP events on the main stream at the phase boundaries of JointTrainer.step (GPU
intervals) beside the host's clock at the same points (when the host had ISSUED everything up to there).  A phase whose GPU interval is close to
the host interval is paced by the host; one whose GPU interval is longer is paced by the kernels.  GPU box."""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene, joint_train
from nerf4k_amd.lib import dvgo, sr_esrnet, sr_train, utils
if os.environ.get('TOOL_SPLIT_STEP') == '0':                             # A/B: k0's step of the dense-TV iterations in one pass after the backward pass
    joint_train._SPLIT_GRID_STEP = False
if os.environ.get('TOOL_SFT_SPLIT') == '0':                               # A/B: the SFT layers' whole backward on the chain (one launch each)
    sr_train._SFT_SPLIT = False
if os.environ.get('SIDE_PRIO') is not None and hasattr(sr_train, '_SIDE_LOW_PRIORITY'):
    sr_train._SIDE_LOW_PRIORITY = os.environ['SIDE_PRIO'] != '0'
dev = torch.device('cuda', 0)
S0 = int(os.environ.get('STEP0', '0'))              # >= 10000: the iterations after tv_before (no TV, sparse grid gradients)
ck = scene.make_llff_checkpoint()
H, W = scene.LLFF_HW
ro, rd, vd = dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(scene.llff_spiral_poses()[0]).to(dev), True, False, False, False)
model = utils.model_from_checkpoint_dict(ck).to(dev).train()
torch.manual_seed(778)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).to(dev).train()
cfg = joint_train.JointCfg.fern_lg_joint_l1()
with contextlib.redirect_stdout(sys.stderr):
    tr = joint_train.JointTrainer(model, net, cfg, dict(ck['render_kwargs'], render_depth=True, rand_bkgd=True), n_train_images=17)
g = torch.Generator(device=dev).manual_seed(5)


def batch(i):
    r0, c0 = (37 * i) % (H - 64), (101 * i) % (W - 64)
    rays = [x[r0:r0 + 64, c0:c0 + 64].reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    return rays + [torch.rand([4096, 3], device=dev, generator=g), torch.rand([65536, 3], device=dev, generator=g), 64, 64]


marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))


def wrap(obj, attr, before=None, after=None):
    fn = getattr(obj, attr)

    def inner(*a, **k):
        if before:
            mark(before)
        r = fn(*a, **k)
        if after:
            mark(after)
        return r
    setattr(obj, attr, inner)


wrap(tr, '_decoder', 'marcher forward done', 'decoder forward issued')
wrap(tr, 'losses', None, 'losses issued')
wrap(joint_train, 'exchange_gradients', 'backward issued', None)
for i in range(4):
    tr.step(*batch(i), global_step=S0 + 1 + i)
torch.cuda.synchronize()
rows = []
n = int(os.environ.get('ITERS', '24'))
t_all = time.perf_counter()
for i in range(n):
    marks.clear()
    mark('start')
    tr.step(*batch(4 + i), global_step=S0 + 5 + i)
    mark('end')
    rows.append(list(marks))
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / n * 1e3
names = [m[0] for m in rows[0]]
gpu = np.array([[r[0][1].elapsed_time(m[1]) for m in r] for r in rows])          # ms since the iteration's start event, GPU timeline
host = np.array([[(m[2] - r[0][2]) * 1e3 for m in r] for r in rows])
print(f'joint iteration {wall:.2f} ms (wall / iteration over {n} iterations, events on)')
print(f'{"boundary":28s} {"GPU reached (ms)":>17s} {"host issued (ms)":>17s} {"GPU interval":>13s} {"host interval":>14s}')
gm, hm = np.median(gpu, 0), np.median(host, 0)
for k, nm in enumerate(names):
    print(f'{nm:28s} {gm[k]:17.2f} {hm[k]:17.2f} {gm[k] - (gm[k - 1] if k else 0):13.2f} {hm[k] - (hm[k - 1] if k else 0):14.2f}')
