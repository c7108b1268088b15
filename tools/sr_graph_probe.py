"""Does replaying the decoder pass of one rank's share (3 windows of 209x209, 111 launches) as a hipGraph beat issuing the launches?  GPU box."""
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
mine = [tiles[i] for i in tp.assign_tiles(tiles, 8)[0]] if TS != 510 else tiles
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    out = net.tile_process_device(x, c, TS, 10, tiles=mine)
    eager = timeit(lambda: net.tile_process_device(x, c, TS, 10, tiles=mine, out=out))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): net.tile_process_device(x, c, TS, 10, tiles=mine, out=out)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            net.tile_process_device(x, c, TS, 10, tiles=mine, out=out)
        out.zero_(); g.replay(); torch.cuda.synchronize()
        same = torch.equal(out, ref)
        graph = timeit(lambda: g.replay())
        print(f'tile {TS}, {len(mine)} windows: eager {eager:.2f} ms, hipGraph replay {graph:.2f} ms, identical output {same}')
    except Exception as e:
        print(f'tile {TS}: eager {eager:.2f} ms; capture failed: {type(e).__name__}: {str(e)[:300]}')
