"""Multi-process (world_size 2 and 3, gloo, CPU) test of the tile sharding / all-gather / assembly logic of
4k-nerf_amd/tile_parallel.py, with the CPU oracle injected in place of the HIP kernels.  The assembled frame must be
bit-identical on every rank to the single-process ``tile_process`` of the oracle with the same tile geometry."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nerf4k_amd  # noqa: F401
from nerf4k_amd import tile_parallel as tp, scene
from oracle import marcher, sr as osr


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    ck = scene.make_llff_checkpoint(seed=9, num_voxels=20 * 20 * 16, mpi_depth=16)
    H, W = 22, 30
    K = scene.LLFF_K.copy()
    K[:2] *= W / scene.LLFF_HW[1]
    rays = marcher.get_rays_of_a_view(H, W, K, scene.llff_spiral_poses()[6], ndc=True)
    sd = osr.make_state_dict(seed=21, num_block=1)
    return ck, H, W, rays, sd


def _fns(ck, sd):
    def march_fn(ro, rd, vd, window_w):
        o = marcher.mpi_forward(ck['model_kwargs'], ck['model_state_dict'], ro, rd, vd, **ck['render_kwargs'])
        return o['rgb_feature'], o['depth']

    def sr_fn(img, cond):
        return osr.sftnet_forward(sd, img, cond)
    return march_fn, sr_fn


def _reference_frame(ck, H, W, rays, sd, tile):
    march_fn, _ = _fns(ck, sd)
    rgb, depth = march_fn(*[r.reshape(-1, 3) for r in rays], W)
    return osr.tile_process(sd, rgb.reshape(H, W, 3).permute(2, 0, 1).unsqueeze(0), depth.reshape(1, H, W), tile)


def _worker(rank, world, port, tile, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        ck, H, W, rays, sd = _scene()
        march_fn, sr_fn = _fns(ck, sd)
        out = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size=tile)
        out8 = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size=tile, out_dtype=torch.uint8)      # 8-bit pixels cross the wire
        q.put((rank, (out.numpy(), out8.numpy())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,tile', [(2, 12), (3, 8)])
def test_tile_sharded_frame_equals_single_process(world, tile):
    ck, H, W, rays, sd = _scene()
    want = _reference_frame(ck, H, W, rays, sd, tile).numpy()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tile, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        # tile windows are marched separately: identical samples, identical SR tiles -> identical pixels
        assert np.array_equal(got[r][0], want), (r, float(np.abs(got[r][0] - want).max()))
        # quantised before the exchange with the reference's to8b rule: byte-identical to to8b of the fp32 frame
        assert got[r][1].dtype == np.uint8 and np.array_equal(got[r][1], (255 * np.clip(want, 0, 1)).astype(np.uint8))


def test_assignment_covers_all_tiles_and_balances():
    tiles = tp.tile_geometry(756, 1008, 189, 10)
    assert len(tiles) == 24                                           # 6x4 tiles, 3 per rank at P=8
    owned = tp.assign_tiles(tiles, 8)
    assert sorted(i for o in owned for i in o) == list(range(24)) and all(len(o) == 3 for o in owned)
    area = lambda i: (tiles[i][5] - tiles[i][4]) * (tiles[i][7] - tiles[i][6])
    loads = [sum(area(i) for i in o) for o in owned]
    assert max(loads) / (sum(loads) / 8) < 1.15          # tile areas are discrete (interior 209^2, edges smaller)
    # reference geometry at test_tile=510 and single-rank degenerate case
    assert [(t[7] - t[6], t[5] - t[4]) for t in tp.tile_geometry(756, 1008, 510)] == [(520, 520), (508, 520), (520, 256), (508, 256)]
    assert tp.assign_tiles(tiles, 1) == [list(range(24))]
    # ragged: more ranks than tiles leaves some ranks empty but still covers everything
    owned = tp.assign_tiles(tp.tile_geometry(20, 20, 16), 8)
    assert sorted(i for o in owned for i in o) == [0, 1, 2, 3]
    assert tp.shard_rows(756, 8, 7) == (672, 756, 96) and tp.shard_rows(756, 1, 0) == (0, 756, 760)


def test_single_process_path_without_process_group():
    ck, H, W, rays, sd = _scene()
    march_fn, sr_fn = _fns(ck, sd)
    out = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size=12)
    want = _reference_frame(ck, H, W, rays, sd, 12)
    assert torch.equal(out, want)
    out8 = tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size=12, out_dtype=torch.uint8)
    assert out8.dtype == torch.uint8 and torch.equal(out8, (255.0 * want.clamp(0, 1)).to(torch.uint8))
    with pytest.raises(ValueError):
        tp.render_frame_tiles(rays, H, W, march_fn, sr_fn, tile_size=12, out_dtype=torch.float16)
