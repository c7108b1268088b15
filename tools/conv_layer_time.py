"""Per-layer timing of the decoder's convolution kernel (HIP events, isolated launches): us, fp32-equivalent TFLOP/s and the
fraction of the bf16x6 matrix floor (2.5 PFLOP/s / 6) for the layer shapes of SFTNet at a 520x520 window."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd.lib.sr_esrnet import _Packed, SFTNet, EPI_LRELU, PRE_UP2X
torch.manual_seed(0)
mode = os.environ.get('K4_SR_MODE', 'bf16x6')
cases = [(64, 32, 520, 520, 0), (96, 32, 520, 520, 0), (128, 32, 520, 520, 0), (160, 32, 520, 520, 0), (192, 64, 520, 520, 0),
         (64, 64, 520, 520, 0), (64, 64, 1040, 1040, PRE_UP2X), (64, 64, 2080, 2080, 0), (64, 3, 2080, 2080, 0), (160, 32, 209, 209, 0)]
if len(sys.argv) > 1:
    cases = [cases[int(a)] for a in sys.argv[1:]]
for cin, cout, H, W, fl in cases:
    sh, sw = (H // 2, W // 2) if fl & PRE_UP2X else (H, W)
    x = torch.randn([sh, sw, 192], device='cuda')
    w = (torch.randn([cout, cin, 3, 3], device='cuda') / (cin * 9) ** 0.5)
    b = torch.randn([cout], device='cuda')
    y = torch.zeros([H, W, 64], device='cuda')
    pk = _Packed(w, b, mode)
    if os.environ.get('K4_TOOL_MULTI', '0') == '1':       # grouped entry point with a ticket queue (the path SFTNet takes)
        net = SFTNet.__new__(SFTNet)
        up = 2 if fl & PRE_UP2X else 1
        plan = []
        run = lambda: SFTNet._conv_multi(net, pk, [{'x': x, 'y': y}], [(sh, sw)], 'x', 0, 192, 'y', 0, 64, cout, up, EPI_LRELU | fl, plan=plan)
    else:
        run = lambda: SFTNet._conv(pk, x, 0, 192, y, 0, 64, cout, H, W, EPI_LRELU | fl)
    run(); torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); e.record(); ev.append((a, e))
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(e) for a, e in ev)[len(ev) // 2] * 1e3
    flop = 2.0 * 9 * cin * cout * H * W
    tf = flop / us / 1e6
    print(f'cin {cin:3d} cout {cout:2d} {H}x{W} flags {fl:2d}: {us:8.1f} us  {tf:6.1f} TFLOP/s  frac {tf / 416.7:.3f}')
