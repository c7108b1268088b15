"""The shader clock the 3x3 kernels actually run at while a 4K frame is decoded: every wave of k4_conv_p16_kernel records its life in s_memtime
(shader clock) and s_memrealtime (100 MHz) ticks; the ratio over all waves of three frames is the sustained clock.  Run with
K4_LIB=<library built with -DK4_P16_TIMING> and K4_SR_SFT_FUSE=0 (the SFT-epilogue instantiation uses the slot for a phase time)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nerf4k_amd  # noqa: F401
from nerf4k_amd import _native as N
from nerf4k_amd.lib import sr_esrnet
assert os.environ.get('K4_SR_SFT_FUSE') == '0'
fn = N.lib().k4_debug_p16_timing
fn.argtypes = [C.c_void_p, C.c_int]
raw = np.zeros([65536, 8], dtype=np.uint64)
buf = raw.ctypes.data_as(C.c_void_p)
torch.manual_seed(777)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).cuda().eval()
x = torch.rand([1, 3, 756, 1008]).cuda(); c = torch.rand([1, 756, 1008]).cuda()
with torch.no_grad():
    for _ in range(2):
        out = net.tile_process_device(x, c, 510, 10)
    torch.cuda.synchronize(); fn(buf, 1)
    t = time.perf_counter()
    for _ in range(3):
        out = net.tile_process_device(x, c, 510, 10, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 3 * 1e3
fn(buf, 0)
live = raw[:, 7] > 0
mt = (raw[live, 5] >> np.uint64(32)).astype(np.float64).sum()
rt = (raw[live, 5] & np.uint64(0xffffffff)).astype(np.float64).sum()
print(f'4K frame (instrumented build) {ms:.1f} ms; k4_conv_p16_kernel waves ran at {mt / rt * 100:.0f} MHz on average (s_memtime / s_memrealtime over {int(raw[live, 7].sum())} wave passes)')
print(f'   => dense fp16 matrix peak at that clock: {2500 * mt / rt * 100 / 2400:.0f} TFLOP/s (2500 at 2400 MHz); three-product floor {2500 * mt / rt * 100 / 2400 / 3:.0f} TFLOP/s fp32-equivalent')
