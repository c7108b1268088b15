O=gpurun_out/r06_dbg3; mkdir -p $O
cat > /tmp/dbg2.py <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
import nerf4k_amd
from nerf4k_amd import scene
from nerf4k_amd.lib import dvgo, sr_train
dev = torch.device('cuda', 0)
n_pre = int(os.environ.get('PRE_STREAMS', '0'))
pre = [torch.cuda.Stream() for _ in range(n_pre)]
for s in pre:
    with torch.cuda.stream(s):
        torch.zeros([1024], device=dev).add_(1)
torch.cuda.synchronize()
if os.environ.get('SIDE_PRIO') == '0':
    sr_train._SIDE_LOW_PRIORITY = False
ck = scene.make_llff_checkpoint()
(H, W), K = scene.LLFF_HW, scene.LLFF_K
poses = scene.llff_spiral_poses()
with torch.no_grad():
    rays = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(poses[0]).to(dev), True, False, False, False)]
out = bench.joint_train_step(ck, rays, H, W, dev)
print(os.environ.get('PRE_STREAMS'), os.environ.get('SIDE_PRIO'), json.dumps({k: out[k] for k in ('ms_per_iteration', 'ms_per_iteration_per_block_graph')}))
PY
timeout 600 python -m pytest tests/test_sr_train_gpu.py tests/test_optim_gpu.py -q -m gpu -k "overlap or side or stream or tape" 2>&1 | tail -3
for v in "PRE_STREAMS=0" "PRE_STREAMS=6" "PRE_STREAMS=12" "PRE_STREAMS=3" "PRE_STREAMS=9"; do
  env $v python /tmp/dbg2.py 2>/dev/null | tail -1
done
timeout 1500 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; echo bench_rc=$?
python tools/bench_summary.py $O/bench_default_line.json 2>/dev/null | tail -3
python -c "
import json; d=json.load(open('$O/bench_default_line.json'))['joint_train_step']; print({k: d[k] for k in ('ms_per_iteration','ms_per_iteration_blocks','ms_per_iteration_per_block_graph','breakdown_ms')})"
