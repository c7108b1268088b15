"""Isolated timing (HIP events, one launch at a time, median of 20) of the decoder's TRAINING kernels at the joint loop's sizes:
a 64x64 patch (configs[4]) -- convolution forward / dgrad (k4_conv2d_nhwc_bf16x6), weight gradient (k4_conv2d_wgrad_dbias_bf16x6) and
the fused SFT layer forward / backward.  The joint iteration's rocprof table mixes two streams (durations include waiting for CUs):
this is the per-kernel number without neighbours."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd import _native as N
from nerf4k_amd.lib.sr_esrnet import _Packed, SFTNet, EPI_LRELU, CONV_SMALL
from nerf4k_amd.lib import sr_train

torch.manual_seed(0)
H = W = int(os.environ.get('PATCH', 64))
dev = 'cuda'


def med(run, n=20):
    run(); torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); e.record(); ev.append((a, e))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2] * 1e3


for cin, cout in [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)]:
    x = torch.randn([H, W, 192], device=dev)
    w = torch.randn([cout, cin, 3, 3], device=dev) / (cin * 9) ** 0.5
    b = torch.randn([cout], device=dev)
    y = torch.zeros([H, W, 64], device=dev)
    gy = torch.randn([H, W, 64], device=dev)
    pk = _Packed(w, b, 'bf16x6')
    t_f = med(lambda: SFTNet._conv(pk, x, 0, 192, y, 0, 64, cout, H, W, EPI_LRELU | CONV_SMALL))      # K4_SR_DEBUG=2048: the row kernel
    t_w = med(lambda: sr_train._wgrad(x, 0, cin, 192, gy, 0, cout, 64, 3, H, W, w.shape, True))
    print(f'conv {cin:3d}->{cout:2d} {H}x{W}: forward {t_f:6.1f} us   wgrad+dbias (zero-fill + kernel) {t_w:6.1f} us')

for C in (32, 64):
    x = torch.randn([H * W, C], device=dev)
    cond = torch.randn([H * W, 32], device=dev)
    gy = torch.randn([H * W, C], device=dev)
    ws = [torch.randn(s, device=dev) * 0.1 for s in ([32, 32], [32], [C, 32], [C], [32, 32], [32], [C, 32], [C])]
    y = torch.empty_like(x)
    t_f = med(lambda: N.check(N.lib().k4_sft_train_fwd(N.f32(x), C, N.f32(cond), 32, H * W, C, *[N.f32(t) for t in ws], 0.2, N.f32(y), C, N.stream()), 'fwd'))
    t_b = med(lambda: sr_train._sft_bwd(x, C, C, cond, gy, 0, C, H * W, ws))
    print(f'sft C={C} {H}x{W}: forward {t_f:6.1f} us   backward (+ reduce) {t_b:6.1f} us')
