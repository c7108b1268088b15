"""Dense Adam step over the LLFF feature grid (417 x 353 x 256 x 9 = 339 M elements, 28 bytes each): median of 20 launches.  K4_LIB selects a variant library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nerf4k_amd
from nerf4k_amd import _native as N
n = 417 * 353 * 256 * 9
p, g, m, v = (torch.rand([n], device='cuda') for _ in range(4))
L = N.lib()
def run():
    N.check(L.k4_adam_upd(N.f32(p), N.f32(g), N.f32(m), N.f32(v), n, 3, 0.9, 0.99, 0.1, 1e-15, N.stream()), 'adam')
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort()
med = ts[10]
print(os.environ.get('K4_LIB', 'default'), 'adam 339 M elements: median %.3f ms = %.0f GB/s = %.3f of 8 TB/s' % (med, n * 28 / med / 1e6, n * 28 / med / 1e6 / 8000))
