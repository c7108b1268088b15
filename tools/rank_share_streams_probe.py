"""8-GPU rank share (decoder only, tile 168: 4 windows of the heaviest rank) with the windows dealt to 1 / 2 / 4 streams that are VERIFIED to overlap
(_native.overlapping_stream), their launch sequences issued interleaved (layer k of every group before layer k + 1 of any): does a second chain fill the
dispatch gaps and the partly filled last round of the first?  Round 4 measured streams neutral / slower -- with PyTorch pool streams (possibly sharing a
hardware queue) and one group issued after the other.  GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd import _native as N
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
if os.environ.get('ALL') == '1':          # every window of the frame (the single-GPU 4K frame at TS=510: four windows)
    mine = list(tiles)
wins = [(x[:, :, t[4]:t[5], t[6]:t[7]], c[:, :, t[4]:t[5], t[6]:t[7]]) for t in mine]
print('windows', [(int(a.shape[2]), int(a.shape[3])) for a, _ in wins])
with torch.no_grad():
    net._forward_hip_multi([w[0] for w in wins], [w[1] for w in wins])          # calibration, packing
    st = net._p16_state()

    def plan_of(idx, slot0):
        Bs, hws = [], []
        for j, i in enumerate(idx):
            a, b = wins[i]
            h, w = int(a.shape[2]), int(a.shape[3])
            B = net._k4_buffers(h, w, dev, slot0 + j)
            B['xin'].copy_(a[0].permute(1, 2, 0)); B['cnd'].copy_(b[0].permute(1, 2, 0))
            Bs.append(B); hws.append((h, w))
        ovf = torch.zeros([N.K4_MAX_JOBS], dtype=torch.int32, device=dev)
        plan = []
        net._record_hip(net._packed(), Bs, hws, plan, p16={'E': st['E'], 'pk': st['pk'], 'ovf': ovf})
        return plan, Bs, ovf

    def run(groups, streams):
        plans = [plan_of(g, 16 + 8 * k) for k, g in enumerate(groups)]
        torch.cuda.synchronize()
        raws = [N.C.c_void_p(s.cuda_stream) for s in streams]
        cur = torch.cuda.current_stream()

        def once():
            for s in streams:
                s.wait_stream(cur)
            n = max(len(p[0]) for p in plans)
            for k in range(n):
                for (plan, _, ovf), s, raw in zip(plans, streams, raws):
                    if k < len(plan):
                        fn, args, what = plan[k]
                        if fn is None:
                            with torch.cuda.stream(s):
                                args[0].zero_()
                        else:
                            N.check(fn(*args, raw), what)
            for s in streams:
                cur.wait_stream(s)
        once(); torch.cuda.synchronize()
        ts = []
        for _ in range(12):
            torch.cuda.synchronize(); t = time.perf_counter()
            once()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
        ts.sort()
        outs = [B['out'].clone() for p in plans for B in p[1]]
        return ts[len(ts) // 2], ts[0], outs

    main = torch.cuda.current_stream()
    n = len(wins)
    res = {}
    res['1 group of %d, current stream' % n] = run([list(range(n))], [main])
    s2 = [N.overlapping_stream(dev, f'probe {i}', group='probe', beside_main=False) for i in range(4)]
    half = (n + 1) // 2
    res['2 groups, 2 verified streams'] = run([list(range(half)), list(range(half, n))], s2[:2])
    res['%d groups of 1, %d verified streams' % (n, min(n, 4))] = run([[i] for i in range(n)][:4], s2[:min(n, 4)]) if n <= 4 else None
    pool = [torch.cuda.Stream() for _ in range(2)]
    res['2 groups, 2 PyTorch pool streams'] = run([list(range(half)), list(range(half, n))], pool)
    ref = res['1 group of %d, current stream' % n][2]
    for k, v in res.items():
        if v is None:
            continue
        same = all(torch.equal(a, b) for a, b in zip(ref, v[2]))
        print(f'{k:45s} median {v[0]:6.2f} ms  best {v[1]:6.2f} ms   pixels identical to the single group: {same}')
