"""ms per joint training iteration (configs[4] on one GPU), a few iterations after warm-up.  GPU box."""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene, joint_train
from nerf4k_amd.lib import dvgo, sr_esrnet, utils
if os.environ.get('TOOL_AUX_WGRAD') == '0':                              # A/B: all weight gradients on their own stream
    from nerf4k_amd.lib import sr_train as _T3
    _T3._AUX_WGRAD = False
if os.environ.get('TOOL_TAIL_SPLIT') == '0':                             # A/B: the tail's weight gradients in one queue
    from nerf4k_amd.lib import sr_train as _T4
    _T4._TAIL_SPLIT = False
if os.environ.get('TOOL_SIDE_LOW') == '1':                                # A/B: the decoder's side streams at the device's lowest priority
    from nerf4k_amd.lib import sr_train as _T2
    _T2._SIDE_LOW_PRIORITY = True
if os.environ.get('TOOL_EARLY_WGS'):
    from nerf4k_amd.lib import masked_adam as _MA2
    _MA2._EARLY_WORKGROUPS = int(os.environ['TOOL_EARLY_WGS'])
if os.environ.get('TOOL_ADAM_LOW') == '1':
    from nerf4k_amd.lib import masked_adam as _MA
    _MA._SIDE_LOW_PRIORITY = True
if os.environ.get('TOOL_SPLIT_STEP') == '0':                             # A/B: k0's step of the dense-TV iterations in one pass after the backward pass
    joint_train._SPLIT_GRID_STEP = False
if os.environ.get('TOOL_SFT_SPLIT') == '0':                               # A/B: the SFT layers' whole backward on the chain (one launch each)
    from nerf4k_amd.lib import sr_train as _T
    _T._SFT_SPLIT = False
dev = torch.device('cuda', 0)
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
losses = []
S0 = int(os.environ.get('STEP0', '0'))              # 0: the first 10,000 iterations of fern_lg_joint_l1 (dense TV on both grids, every voxel's Adam state moves); >= 10000: the other 290,000 (no TV, MaskedAdam skips voxels without gradient)
for i in range(3):
    losses.append(float(tr.step(*batch(i), global_step=S0 + 1 + i)['total']))
torch.cuda.synchronize()
n = int(os.environ.get('ITERS', '6'))
if os.environ.get('K4_TOOL_NOGC') == '1':                               # diagnosis: are the slow blocks Python's cyclic garbage collector?
    import gc
    gc.collect(); gc.disable()
blocks = []
for b in range(int(os.environ.get('BLOCKS', '5'))):                  # the iteration is paced by the host: one block of 6 iterations is noisy from box to box
    t = time.perf_counter()
    for i in range(n):
        tr.step(*batch(3 + b * n + i), global_step=S0 + 4 + b * n + i)
    t_host = time.perf_counter() - t              # the host has issued everything (a step that reads a loss back synchronises inside: then host == wall)
    torch.cuda.synchronize()
    blocks.append(((time.perf_counter() - t) / n * 1e3, t_host / n * 1e3))
if os.environ.get('SHOW_BLOCKS') == '1':
    print('blocks (wall, host) ms:', [(round(a, 2), round(b, 2)) for a, b in blocks])
blocks.sort()
med = blocks[len(blocks) // 2]
print('joint iteration ms', round(med[0], 2), '(host returned after', round(med[1], 2), 'ms per iteration; median of', len(blocks), 'blocks of', n,
      'iterations, fastest', round(blocks[0][0], 2), 'slowest', round(blocks[-1][0], 2), ') first losses', [round(v, 5) for v in losses])
if os.environ.get('PROFILE_HOST') == '1':
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        tr.step(*batch(20 + i), global_step=S0 + 30 + i)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(45)
