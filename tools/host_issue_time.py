"""Host-side cost of issuing one marcher call (Python + ctypes + 2 kernel launches), measured while the GPU queue is not
the limiter: time to ENQUEUE n calls vs time until they finish."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd import scene
from nerf4k_amd.lib import utils, dvgo
ck = scene.make_llff_checkpoint(num_voxels=96 * 96 * 64, mpi_depth=64)          # small scene: GPU time per call is tiny
model = utils.model_from_checkpoint_dict(ck).cuda().eval()
H, W = 64, 96
K = scene.LLFF_K.copy(); K[:2] *= W / scene.LLFF_HW[1]
ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(scene.llff_spiral_poses()[0]).cuda(), True, False, False, False)]
rk = ck['render_kwargs']
out = (torch.empty([H * W, 3], device='cuda'), torch.empty([H * W], device='cuda'), torch.empty([H * W], device='cuda'))
with torch.no_grad():
    for _ in range(20):
        model(ro, rd, vd, k4_img_w=W, k4_out=out, **rk)
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        model(ro, rd, vd, k4_img_w=W, k4_out=out, **rk)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'enqueue {1e6 * (t1 - t0) / n:.1f} us/call, complete {1e6 * (t2 - t0) / n:.1f} us/call ({H * W} rays)')
