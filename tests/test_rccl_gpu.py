"""RCCL on the single leased GPU (verdict r05 item 4a): a world-size-1 `nccl` process group that runs the production collective calls of
the tile-parallel renderer and of the joint step's gradient exchange on device buffers.  In a subprocess: the process group must not leak
into the other tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world_size_one_runs_the_production_collectives():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_world1_smoke.py')], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('RCCL_WORLD1 ')][-1]
    res = json.loads(line[len('RCCL_WORLD1 '):])
    assert res['ok'] and res['backend'] == 'nccl' and res['world_size'] == 1
    assert res['tile_all_gather']['fp32_equal'] and res['tile_all_gather']['uint8_equal']
    assert res['gradient_exchange']['gradients_equal'] and res['gradient_exchange']['sparse_bytes_gathered'] > 0
