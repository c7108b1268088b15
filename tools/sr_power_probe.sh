#!/bin/bash
# Is the decoder power / clock limited?  Runs the 4K decoder in a loop and samples rocm-smi (power, sclk, mclk) while it runs.
mkdir -p gpurun_out/r2h
python - <<'PY' &
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import nerf4k_amd
from nerf4k_amd.lib import sr_esrnet
torch.manual_seed(777)
net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).cuda().eval()
x = torch.rand([1, 3, 756, 1008]).cuda(); c = torch.rand([1, 756, 1008]).cuda()
with torch.no_grad():
    net.tile_process_device(x, c, 510, 10); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 12:
        net.tile_process_device(x, c, 510, 10); n += 1
    torch.cuda.synchronize()
print('frames', n, 'ms/frame', (time.time() - t0) / n * 1e3)
PY
PID=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)" | head -8
  echo ---
  sleep 1
done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
