"""Production entry points, decoder only: SFTNet._forward_hip_multi (one grouped launch per layer) against SFTNet._forward_hip_streams (one window group per
verified worker stream, launch sequences interleaved) on the windows of the heaviest rank of 8; wall time.  GPU box; needs profiles/r06_interleaved_stream_decode_tapes_neutral.patch applied (the streams form is not in the product)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd.lib import sr_esrnet
from nerf4k_amd import tile_parallel as tp
torch.manual_seed(777)
dev = torch.device('cuda', 0)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).cuda().eval()
x = torch.rand([1, 3, 756, 1008]).cuda(); c = torch.rand([1, 1, 756, 1008]).cuda()
TS = int(os.environ.get('TS', '168'))
tiles = tp.tile_geometry(756, 1008, TS, 10)
owned = tp.assign_tiles(tiles, 8)
rk = max(range(8), key=lambda q: sum((tiles[i][5] - tiles[i][4]) * (tiles[i][7] - tiles[i][6]) for i in owned[q]))
mine = [tiles[i] for i in owned[rk]]
xs = [x[:, :, t[4]:t[5], t[6]:t[7]] for t in mine]
cs = [c[:, :, t[4]:t[5], t[6]:t[7]] for t in mine]


def timeit(fn, n=12):
    fn(); fn(); torch.cuda.synchronize()
    w = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); w.append((time.perf_counter() - t) * 1e3)
    w.sort()
    return w[len(w) // 2]


with torch.no_grad():
    print('TS', TS, 'windows', [(int(a.shape[2]), int(a.shape[3])) for a in xs])
    print('grouped           ', round(timeit(lambda: net._forward_hip_multi(xs, cs)), 2), 'ms')
    for S in (2, 3, 4):
        if S <= len(xs):
            pool = tp._stream_pool(dev, S)
            print(f'{S} streams interleaved', round(timeit(lambda: net._forward_hip_streams(xs, cs, pool)), 2), 'ms')
