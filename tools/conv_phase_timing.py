"""Phase breakdown of k4_conv_b6v2_kernel per layer shape: run with K4_LIB=<library built with -DK4_SR_TIMING>."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf4k_amd
from nerf4k_amd import _native as N
from nerf4k_amd.lib.sr_esrnet import _Packed, SFTNet, EPI_LRELU
mode = os.environ.get('K4_SR_MODE', 'f16x3')
names = ['prologue', 'wait raw + scale + split + LDS stores', 'barrier A', 'issue next loads + first fragments', 'MFMA phase',
         'next chunk maximum (waits raw)', 'barrier B', 'epilogue']
L = N.lib()
fn = L.k4_debug_sr_timing
fn.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
for cin, cout, H, W in ((160, 32, 520, 520), (192, 64, 520, 520), (64, 64, 2080, 2080)):
    x = torch.randn([H, W, 192], device='cuda'); w = torch.randn([cout, cin, 3, 3], device='cuda') / (cin * 9) ** 0.5
    b = torch.randn([cout], device='cuda'); y = torch.zeros([H, W, 64], device='cuda')
    pk = _Packed(w, b, mode)
    run = lambda: SFTNet._conv(pk, x, 0, 192, y, 0, 64, cout, H, W, EPI_LRELU)
    run(); torch.cuda.synchronize(); fn(buf, 1)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    e.record(); torch.cuda.synchronize()
    fn(buf, 1)
    t = [float(v) for v in buf]
    waves, tot = t[8], sum(t[:8])
    print(f'cin {cin} cout {cout} {H}x{W} ({mode}): {a.elapsed_time(e) / 5 * 1e3:.1f} us per launch; {waves / 5:.0f} waves per launch, {tot / waves:.0f} ticks per wave, chunks {-(-cin // 16)}')
    for i, n in enumerate(names):
        print(f'   {n:44s} {100 * t[i] / tot:5.1f} %  {t[i] / waves:9.0f} ticks/wave')
