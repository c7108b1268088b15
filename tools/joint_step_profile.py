"""Host-side profile (cProfile) of the joint training iteration: where the wall time of `JointTrainer.step` goes.  GPU box."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene, joint_train
from nerf4k_amd.lib import dvgo, sr_esrnet, utils
import contextlib

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


for i in range(2):
    tr.step(*batch(i), global_step=S0 + 1 + i)
torch.cuda.synchronize()
# K4_PROF_BWD=1: the backward pass on the calling thread (no engine worker threads), so that the Function.backward bodies show up below
ctx = torch.autograd.set_multithreading_enabled(False) if os.environ.get('K4_PROF_BWD', '0') == '1' else contextlib.nullcontext()
pr = cProfile.Profile()
with ctx:
    for i in range(2):
        tr.step(*batch(2 + i), global_step=S0 + 3 + i)
    torch.cuda.synchronize()
    pr.enable()
    for i in range(3):
        tr.step(*batch(4 + i), global_step=S0 + 5 + i)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(int(os.environ.get('K4_PROF_ROWS', 28)))
if os.environ.get('K4_PROF_BWD', '0') == '1':
    st.sort_stats('cumulative').print_stats(40)
