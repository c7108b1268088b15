"""The grid-maintenance kernels of the training step alone (bench.training_step_kernels without the scatter cases): ms and fraction of HBM peak."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
os.environ.setdefault('K4_TOOL_NO_SCATTER', '1')
out = bench.training_step_kernels(torch.device('cuda', 0))
for k, v in out.items():
    if isinstance(v, dict) and ('adam' in k or 'tv' in k):
        print(f"{k:28s} {v['ms']:7.3f} ms  {v['B_per_voxel']:5.2f} B/voxel  {v['GBs']:7.1f} GB/s  frac {v['frac_hbm']}")
