"""The heaviest rank's share of the 8-GPU 4K job (tile 168) as bench.py times it -- serial, then with the next frame's march issued before the decode -- for a
rocprofv3 kernel trace: FRAME_SCRIPT=tools/rank_share_timeline.py tools/four_k_timeline.sh."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
TS = int(os.environ.get('TS', '168'))
with torch.no_grad():
    frames = [dvgo.get_rays_of_a_view(H, W, scene.LLFF_K, torch.from_numpy(poses[i]).to(dev), True, False, False, False) for i in range(3)]
    tl = tp.tile_geometry(H, W, TS, 10)
    owned = tp.assign_tiles(tl, 8)
    area = [sum((tl[i][5] - tl[i][4]) * (tl[i][7] - tl[i][6]) for i in o) for o in owned]
    sub = bench._SubsetGeometry(tl, owned[max(range(8), key=lambda q: area[q])])
    for q in range(3):
        sub.render(frames[q % 3], march_fn, sr_fn)
    torch.cuda.synchronize()
    n = int(os.environ.get('FRAMES', '8'))
    t = time.perf_counter()
    if os.environ.get('PIPELINED', '1') == '1':
        st_next = sub.march(frames[0], march_fn)
        for q in range(n):
            st_cur, st_next = st_next, sub.march(frames[(q + 1) % 3], march_fn)
            sub.decode(st_cur, sr_fn)
    else:
        for q in range(n):
            sub.render(frames[q % 3], march_fn, sr_fn)
    torch.cuda.synchronize()
    print('ms per rank-share frame', round((time.perf_counter() - t) / n * 1e3, 2))
