"""Time ONE rank's share (decoder only) of the 8-GPU 4K job on a single GPU: the tiles of the heaviest rank (RANK_OF_8 = another) at tile size TS."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd.lib import sr_esrnet
from nerf4k_amd import tile_parallel as tp
torch.manual_seed(777)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).cuda().eval()
x = torch.rand([1, 3, 756, 1008]).cuda(); c = torch.rand([1, 756, 1008]).cuda()
TS = int(os.environ.get("TS", "189"))
tiles = tp.tile_geometry(756, 1008, TS, 10)
owned = tp.assign_tiles(tiles, 8)
rk = int(os.environ.get('RANK_OF_8', '-1'))
if rk < 0:          # the heaviest rank by padded area (what bench.py's projection times)
    rk = max(range(8), key=lambda q: sum((tiles[i][5] - tiles[i][4]) * (tiles[i][7] - tiles[i][6]) for i in owned[q]))
mine = [tiles[i] for i in owned[rk]]
with torch.no_grad():
    out = net.tile_process_device(x, c, TS, 10, tiles=mine)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        out = net.tile_process_device(x, c, TS, 10, tiles=mine, out=out)
    torch.cuda.synchronize()
print('rank-%d share of 8, tile_size %d, %d tiles:' % (rk, TS, len(mine)), ': ms', round((time.perf_counter() - t) / 10 * 1e3, 2), [(t_[5] - t_[4], t_[7] - t_[6]) for t_ in mine])
