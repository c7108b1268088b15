"""cProfile of the host side of the single-GPU 4K frame loop (tile_parallel.render_frame_tiles at test_tile=510): what the ~1 ms between two frames' kernels is."""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd import scene, tile_parallel as tp
from nerf4k_amd.lib import sr_esrnet, utils, dvgo
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
torch.manual_seed(777)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).to(dev).eval()
H, W = scene.LLFF_HW
poses = scene.llff_spiral_poses()
march_fn, sr_fn = tp.hip_march_fn(model, dict(ck['render_kwargs'])), tp.hip_sr_fn(net)
with torch.no_grad():
    frames = [dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(poses[i]).to(dev), True, False, False, False) for i in range(3)]
    hr = tp.render_frame_tiles(frames[0], H, W, march_fn, sr_fn, 510)
    hr = tp.render_frame_tiles(frames[1], H, W, march_fn, sr_fn, 510, out=hr)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(6):
        hr = tp.render_frame_tiles(frames[i % 3], H, W, march_fn, sr_fn, 510, out=hr)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats('tottime').print_stats(22)
st.sort_stats('cumulative').print_stats(30)
