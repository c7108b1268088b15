"""Phase breakdown of k4_conv_p16_kernel (csrc/k4_sr_p16.hip) per layer shape, four 520x520 windows per launch as a 4K frame issues them:
run with K4_LIB=<library built with -DK4_P16_TIMING>.  Ticks are s_memtime counts summed over waves (every wave: 2 rows x 32 columns x 32
output channels, 54 MFMAs = 1728 matrix-pipe cycles per input chunk)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import _native as N
from nerf4k_amd.lib.sr_esrnet import _PackedP16, _PackedSfe, SFTNet, SFTLayer, EPI_LRELU
names = ['prologue (to the first barrier)', 'barrier waits (DMA landed, all waves)', 'DMA issue', 'MFMA sub-stages + fragment reads', 'epilogue', 'SFT epilogue',
         'SFT epilogue: its barrier', '-']
L = N.lib()
fn = L.k4_debug_p16_timing
fn.argtypes = [C.c_void_p, C.c_int]
import numpy as np
raw = np.zeros([65536, 8], dtype=np.uint64)
buf = raw.ctypes.data_as(C.c_void_p)
torch.manual_seed(0)
nwin = 4
for cin, cout, sft in ((64, 32, False), (160, 32, False), (160, 32, True), (192, 64, False), (192, 64, True)):
    H = W = 520
    xps = [torch.randint(-2 ** 31, 2 ** 31 - 1, [H, W, 192], device='cuda', dtype=torch.int64).to(torch.int32) & 0x3bff3bff for _ in range(nwin)]
    ys = [torch.zeros([H, W, 64], device='cuda') for _ in range(nwin)]
    y2s = [torch.zeros([H, W, 192], device='cuda', dtype=torch.int32) for _ in range(nwin)]
    conds = [torch.rand([H, W, 32], device='cuda') for _ in range(nwin)]
    w = torch.randn([cout, cin, 3, 3], device='cuda') / (cin * 9) ** 0.5
    b = torch.randn([cout], device='cuda')
    pkp = _PackedP16(w, b, [0] * (cin // 16))
    sfe = _PackedSfe(SFTLayer(cout, 32).cuda(), 9)
    net = SFTNet.__new__(SFTNet)
    torch.nn.Module.__init__(net)
    net.num_grow_ch = 32
    ovf = torch.zeros([8], dtype=torch.int32, device='cuda')
    Bs = [{'xp': xp, 'y': y, 'y2': y2, 'cond': c} for xp, y, y2, c in zip(xps, ys, y2s, conds)]
    hws = [(H, W)] * nwin
    if sft:
        run = lambda: SFTNet._conv_p16_sft_multi(net, pkp, sfe, Bs, hws, 'xp', 0, 192, ('y', 0, 64) if cout == 64 else None, EPI_LRELU, None, 'y2', 0, 192, 0, ovf, [])
    else:
        run = lambda: SFTNet._conv_p16_multi(net, pkp, Bs, hws, 'xp', 0, 192, 'y', 0, 64, 1, EPI_LRELU, None, 0 if cout == 32 else None, ovf, [])
    run(); torch.cuda.synchronize(); fn(buf, 1)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    e.record(); torch.cuda.synchronize()
    fn(buf, 1)
    t = [float(v) for v in raw.sum(0)]
    if not sft:
        # slot 5: the wave's life in s_memtime << 32 | in s_memrealtime (100 MHz) ticks, summed over 5 launches; slot 6 (last launch only is meaningful:
        # '+=' of 5 launches -- so one more launch after a reset): start << 24 | XCC << 16 | HW_ID[15:0] (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
        live = raw[:, 7] > 0
        mt, rt = (raw[live, 5] >> np.uint64(32)).astype(np.float64), (raw[live, 5] & np.uint64(0xffffffff)).astype(np.float64)
        print(f'   s_memtime runs at {mt.sum() / rt.sum() * 100:.0f} MHz during this launch')
        fn(buf, 1); run(); torch.cuda.synchronize(); fn(buf, 0)
        live = raw[:, 7] > 0
        start = (raw[live, 6] >> np.uint64(24)).astype(np.int64)
        life = (raw[live, 5] & np.uint64(0xffffffff)).astype(np.int64)
        cu = ((raw[live, 6] >> np.uint64(4)) & np.uint64(0xfff)).astype(np.int64) >> 4 | (((raw[live, 6] >> np.uint64(16)) & np.uint64(15)).astype(np.int64) << 8)      # XCC | SE SH CU
        simd = ((raw[live, 6] >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
        conc = []
        for c in np.unique(cu)[:64]:
            m = cu == c
            ev = sorted([(s_, 1) for s_ in start[m]] + [(s_ + l_, -1) for s_, l_ in zip(start[m], life[m])])
            cur = best = 0
            for _, d in ev:
                cur += d; best = max(best, cur)
            conc.append(best)
        print(f'   {len(np.unique(cu))} CUs seen; matrix waves resident at once per CU: median {int(np.median(conc))}, max {max(conc)}; waves per SIMD id: {np.bincount(simd, minlength=4).tolist()}')
        fn(buf, 1)
        t[5] = t[6] = 0.0
        raw[:, 5] = 0; raw[:, 6] = 0
    waves, tot = t[7], sum(t[:7])
    per = raw[raw[:, 7] > 0, :7].sum(1).astype(np.float64) / 5
    print(f'   ticks per wave pass: min {per.min():.0f}  p10 {np.percentile(per, 10):.0f}  median {np.median(per):.0f}  p90 {np.percentile(per, 90):.0f}  max {per.max():.0f}')
    us = a.elapsed_time(e) / 5 * 1e3
    print(f'cin {cin} cout {cout} sft {int(sft)} 4 x {H}x{W}: {us:.1f} us per launch; {waves / 5:.0f} waves per launch, {tot / waves:.0f} ticks per wave '
          f'({cin // 16} chunks x 1728 = {cin // 16 * 1728} matrix cycles); wave-ticks / (1024 SIMDs x 2 slots) = {tot / 5 / 2048:.0f} ticks per slot per launch')
    for i, n in enumerate(names[:7]):
        print(f'   {n:44s} {100 * t[i] / tot:5.1f} %  {t[i] / waves:9.0f} ticks/wave')
