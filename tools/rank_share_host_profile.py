import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
import bench
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
    tl = tp.tile_geometry(H, W, 168, 10)
    owned = tp.assign_tiles(tl, 8)
    area = [sum((tl[i][5] - tl[i][4]) * (tl[i][7] - tl[i][6]) for i in o) for o in owned]
    sub = bench._SubsetGeometry(tl, owned[max(range(8), key=lambda q: area[q])])
    for q in range(3):
        sub.render(frames[q % 3], march_fn, sr_fn)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for q in range(8):
        sub.render(frames[q % 3], march_fn, sr_fn)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats('tottime').print_stats(18)
st.sort_stats('cumulative').print_stats(26)
