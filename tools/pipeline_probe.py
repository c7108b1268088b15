"""Timeline of bench.py's pipelined marcher loop: per step the host's issue time and the frame's start / end on its stream (HIP events against one start event).
`python tools/pipeline_probe.py [streams]` -- run a few times: the loop is bimodal on some boxes.  GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import nerf4k_amd  # noqa: F401
from nerf4k_amd import scene
from nerf4k_amd.lib import utils
dev = torch.device('cuda', 0)
ck = scene.make_llff_checkpoint()
model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
H, W = scene.LLFF_HW
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
run = bench.MarcherRun(model, scene.llff_spiral_poses(), ck['render_kwargs'], H, W, scene.LLFF_K, dev, 1, 0, False, S)
steps = 40
with torch.no_grad():
    for rep in range(3):
        for i in range(8):
            run.step(i)
        run.sync()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(); b.record()
        e0 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        host = []
        for i in range(steps):
            run.step(i, timed=ev[i])
            host.append((time.perf_counter() - t0) * 1e3)
        t_issue = (time.perf_counter() - t0) * 1e3
        run.sync()
        t_all = (time.perf_counter() - t0) * 1e3
        st = [e0.elapsed_time(a) for a, _ in ev]
        en = [e0.elapsed_time(b) for _, b in ev]
        print(f'rep {rep}: streams {S}: issue {t_issue:.2f} ms, all {t_all:.2f} ms for {steps} steps ({t_all / steps:.3f} per step)')
        print('   host issue done at  :', ' '.join(f'{v:.2f}' for v in host[:12]), '...', f'{host[-1]:.2f}')
        print('   frame start on GPU  :', ' '.join(f'{v:.2f}' for v in st[:12]), '...', f'{st[-1]:.2f}')
        print('   frame end on GPU    :', ' '.join(f'{v:.2f}' for v in en[:12]), '...', f'{en[-1]:.2f}')
