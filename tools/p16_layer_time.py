"""Per-layer timing (HIP events, median of 10 isolated launches) of the 3x3 convolution on PRE-SPLIT activations (k4_conv3x3_p16_multi,
csrc/k4_sr_p16.hip) beside the per-tile 'f16x3' kernel on fp32 activations, for the layer shapes of SFTNet: us, fp32-equivalent TFLOP/s,
fraction of the 3-product fp16 matrix floor (2.5 PFLOP/s / 3).  K4_TOOL_WINDOWS=4: four windows per launch (what a 4K frame issues).
Usage: python tools/p16_layer_time.py [case indices]   (GPU box)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd  # noqa: F401
from nerf4k_amd import _native as N
from nerf4k_amd.lib.sr_esrnet import _Packed, _PackedP16, _PackedP16Up, SFTNet, EPI_LRELU, PRE_UP2X
torch.manual_seed(0)
nwin = int(os.environ.get('K4_TOOL_WINDOWS', '1'))
cases = [(64, 32, 520, 520, 0), (96, 32, 520, 520, 0), (128, 32, 520, 520, 0), (160, 32, 520, 520, 0), (192, 64, 520, 520, 0),
         (64, 64, 520, 520, 0), (64, 64, 1040, 1040, PRE_UP2X), (64, 64, 2080, 2080, 0), (160, 32, 209, 209, 0), (192, 64, 209, 209, 0), (64, 64, 2080, 2080, PRE_UP2X)]
only = os.environ.get('K4_TOOL_ONLY', '')          # 'p16': time only the pre-split kernel (profiling runs)
if len(sys.argv) > 1:
    cases = [cases[int(a)] for a in sys.argv[1:]]


def med(run):
    run(); torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); e.record(); ev.append((a, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[len(ev) // 2] * 1e3


for cin, cout, H, W, fl in cases:
    sh, sw = (H // 2, W // 2) if fl & PRE_UP2X else (H, W)
    up = 2 if fl & PRE_UP2X else 1
    xst = 64 if cin == 64 else 192          # pixel stride of the input image (the 4x layers' images are 64 wide: 32-bit buffer offsets)
    xs = [torch.randn([sh, sw, xst], device='cuda') for _ in range(nwin)]
    xps = [torch.randint(-2 ** 31, 2 ** 31 - 1, [sh, sw, xst], device='cuda', dtype=torch.int64).to(torch.int32) & 0x3bff3bff for _ in range(nwin)]   # finite fp16 pairs
    ys = [torch.zeros([H, W, 64], device='cuda') for _ in range(nwin)]
    w = (torch.randn([cout, cin, 3, 3], device='cuda') / (cin * 9) ** 0.5)
    b = torch.randn([cout], device='cuda')
    pk, pkp = _Packed(w, b, 'f16x3'), (_PackedP16Up if fl & PRE_UP2X else _PackedP16)(w, b, [0] * (cin // 16))
    net = SFTNet.__new__(SFTNet)
    ovf = torch.zeros([8], dtype=torch.int32, device='cuda')
    Bs = [{'x': x, 'xp': xp, 'y': y} for x, xp, y in zip(xs, xps, ys)]
    hws = [(sh, sw)] * nwin
    t_tile = 1.0 if only == 'p16' else med(lambda: SFTNet._conv_multi(net, pk, Bs, hws, 'x', 0, xst, 'y', 0, 64, cout, up, EPI_LRELU | fl, plan=[]))
    t_p32 = 1.0 if only == 'p16' else med(lambda: SFTNet._conv_p16_multi(net, pkp, Bs, hws, 'xp', 0, xst, 'y', 0, 64, up, EPI_LRELU | fl, None, None, ovf, []))
    t_p16 = med(lambda: SFTNet._conv_p16_multi(net, pkp, Bs, hws, 'xp', 0, xst, 'y', 0, 64, up, EPI_LRELU | fl, None, 0, ovf, []))
    flop = 2.0 * 9 * cin * cout * H * W * nwin
    f = lambda us: f'{us:8.1f} us {flop / us / 1e6:6.1f} TF frac {flop / us / 1e6 / 833.3:.3f}'
    print(f'cin {cin:3d} cout {cout:2d} {H}x{W} x{nwin} flags {fl:2d}: f16x3 per-tile {f(t_tile)} | p16 in, fp32 out {f(t_p32)} | p16 in, p16 out {f(t_p16)}', flush=True)
