#!/usr/bin/env python
"""Benchmark of the 4K-NeRF rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Metric (BASELINE.json): Mrays/s on LLFF-fern ``render_test``.  A *step* is one full 1008x756 frame
(762,048 rays x 256 samples) of BASELINE configs[1] -- "LLFF fern_lg_pretrain render_test at 1008x756,
1xMI355X, HIP ray-marcher only (no SR)" -- marched by the fused HIP kernel on the seeded synthetic
LLFF scene (no datasets / checkpoints exist offline), camera poses cycling through the 20-pose spiral.
Rays are resident in HBM before the timed region (the reference's own timer starts after
``get_rays_of_a_view`` too, run_sr.py:104-111).

The timed region pipelines the frames over 3 HIP streams (the geometry kernel of frame i+1 overlaps the shading kernel of
frame i); ``roofline`` is measured on isolated launches of the same call (one stream, HIP events on the launch stream).

N>1: the frames of the pose sequence are sharded over the ranks (independent units, full model replica per GPU, no data-path
collective; ``scaling: "weak"``, value = rays of all ranks / max-over-ranks time).  ``--shard rows`` selects the
one-frame-split-N-ways form instead (row bands + one asynchronous ``all_gather_into_tensor`` of 5 floats per ray, strong).
The 4K pipeline (``four_k``, BASELINE configs[2]/[3]) always shards the SR tiles of ONE frame over the ranks with one
all-gather of the final HR pixels.

Extra objects on the JSON line (rank 0):
  roofline        -- dominant kernels (fused marcher call), HBM bound: algorithmic bytes per launch (device counters,
                     SURVEY.md 8d formula) / mean isolated launch time; ``traffic`` = fabric bytes per launch from the
                     committed rocprofv3 PMC passes (profiles/r01_marcher_traffic.json).
  cpu_baseline    -- the CPU oracle (oracle/marcher.py, kind "port") timed on this host's cores on one full frame.
  parity_vs_oracle-- the HIP marcher's output on that frame against the oracle's (PSNR, max errors): the oracle as checker.
  four_k / four_k_fp32mfma / four_k_bf16x3 -- march + SFTNet x4 (tile 510) to 4032x3024 per decoder arithmetic.
  reference_pipeline_baseline -- the reference's op-per-launch sequence (staged kernels, 8192-ray chunks) on this GPU.
  training_step_kernels       -- Adam / masked Adam / TV / grid-sample backward streaming kernels vs the HBM roof.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=80)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-stride', type=int, default=1, help='CPU baseline marches every k-th row and column')
    ap.add_argument('--small', action='store_true', help='reduced scene (debug only; never a reported number)')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams frames alternate on (2: the geometry kernel of frame '
                    'i+1 overlaps the matrix-core shading kernel of frame i)')
    ap.add_argument('--shard', choices=['frames', 'rows'], default='frames',
                    help='N>1, marcher metric: "frames" = every rank renders its own frames of the pose sequence (independent '
                         'units, no collective, weak scaling); "rows" = every frame split in row bands + all_gather of the final '
                         'pixels (strong scaling; the 4K pipeline always shards tiles of one frame, see four_k)')
    ap.add_argument('--no-extras', action='store_true', help='marcher line only: skip the reference-pipeline and training-step '
                    'side measurements (profiling runs)')
    ap.add_argument('--backend', default='nccl', help=argparse.SUPPRESS)          # gloo + --same-device: 1-GPU logic smoke test
    ap.add_argument('--same-device', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--sr-frames', type=int, default=3, help='4K frames (march + SFTNet x4, test_tile=510) timed for the '
                    'secondary frames/s figure (N=1 only; 0 disables)')
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)         # nccl IS RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import nerf4k_amd  # noqa: F401
    from nerf4k_amd import scene, _native
    from nerf4k_amd.lib import utils, dvgo

    _native.lib()
    t0 = time.time()
    if args.small:
        ck = scene.make_llff_checkpoint(num_voxels=96 * 96 * 64, mpi_depth=64)
    else:
        ck = scene.make_llff_checkpoint()
    model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
    rk = ck['render_kwargs']
    H, W = scene.LLFF_HW
    K = scene.LLFF_K
    poses = scene.llff_spiral_poses()
    if rank == 0:
        print(f'[bench] scene ready in {time.time() - t0:.1f}s: world_size={model.world_size.tolist()} '
              f'k0_ch={model.k0_dim} rays/frame={H * W}', file=sys.stderr)

    # band of pixel rows owned by this rank (multiples of 8 rows -> whole 8x8 wave tiles)
    from nerf4k_amd import tile_parallel as tp
    by_rows = world > 1 and args.shard == 'rows'
    r0, r1, rows_per = tp.shard_rows(H, world, rank) if by_rows else (0, H, H)
    if world > 1 and not by_rows:          # rank r renders frames r, r+N, r+2N, ... of the pose sequence
        poses = [poses[(rank + i * world) % len(poses)] for i in range(len(poses))]
    rays = []
    with torch.no_grad():
        for p in poses:
            ro, rd, vd = dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(p).to(dev), True, False, False, False)
            rays.append(tuple(x[r0:r1].reshape(-1, 3).contiguous() for x in (ro, rd, vd)))
    n_band = (r1 - r0) * W
    slot = rows_per * W
    # all-gather buffers (double buffered): [rgb n x 3 | depth n | alphainv n] -- the marcher writes straight into them
    send = [torch.zeros([5 * slot], dtype=torch.float32, device=dev) for _ in range(2)]
    recv = [torch.empty([world * 5 * slot], dtype=torch.float32, device=dev) for _ in range(2)] if by_rows else None
    outs = [(b[:3 * n_band].view(n_band, 3), b[3 * slot:3 * slot + n_band], b[4 * slot:4 * slot + n_band]) for b in send]
    works = [None, None]

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())

    def step(i, counters=None, timed=None):
        b = i & 1
        st = streams[i % len(streams)]
        with torch.cuda.stream(st):
            if works[b] is not None:
                works[b].wait()                  # the collective that last read send[b] has finished
                works[b] = None
            ro, rd, vd = rays[i % len(rays)]
            if timed is not None:
                timed[0].record(st)
            out = model(ro, rd, vd, k4_img_w=W, k4_counters=counters, k4_out=outs[b], k4_ws_slot=i % len(streams), **rk)
            if timed is not None:
                timed[1].record(st)
            if by_rows:                          # final pixels only; asynchronous, overlaps the next frame's march
                works[b] = dist.all_gather_into_tensor(recv[b], send[b], async_op=True)
        return out

    def sync():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()             # all streams

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        sync()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t_start = time.perf_counter()
        for i in range(args.steps):
            step(i, timed=ev[i])
        sync()
        elapsed = time.perf_counter() - t_start
        kern_ms = [a.elapsed_time(b) for a, b in ev]

        # untimed: (a) algorithmic bytes per launch from device counters, (b) ISOLATED launch duration (one stream,
        # HIP events around each marcher call on the launch stream) -- the figure the rocprofv3 --stats summary
        # of `bench.py --streams 1` must agree with (geom + shade kernel averages)
        cnt = torch.zeros(4, dtype=torch.int64, device=dev)
        nf = min(args.steps, len(rays))
        iso = []
        st0 = streams[0]
        for i in range(nf):
            with torch.cuda.stream(st0):
                ro, rd, vd = rays[i % len(rays)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st0)
                model(ro, rd, vd, k4_img_w=W, k4_counters=None, k4_out=outs[0], k4_ws_slot=0, **rk)
                e1.record(st0)
                iso.append((e0, e1))
                model(ro, rd, vd, k4_img_w=W, k4_counters=cnt, k4_out=outs[0], k4_ws_slot=0, **rk)
        sync()
        iso_ms = float(np.mean([a.elapsed_time(b) for a, b in iso]))
        n_inb, n_mask, n_alpha, n_shade = [c / nf for c in cnt.cpu().tolist()]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        rays_per_step = H * W * (1 if (world == 1 or by_rows) else world)     # frames mode: every rank renders a frame per step
        value = rays_per_step * args.steps / elapsed / 1e6
        eff_ms = elapsed / args.steps * 1e3            # per-frame time of the timed region (frames overlap on the streams)
        b_alg = n_band * 56 + n_inb * 1 + n_mask * 32 + n_shade * 8 * model.k0_dim * 4
        achieved = b_alg / (iso_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None            # HBM-side bytes per launch: PMC passes of this command, committed under profiles/
        tp_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_marcher_traffic.json')
        if os.path.exists(tp_) and not args.small:
            tj = json.load(open(tp_))
            traffic, traffic_src = int(tj['fabric_bytes_per_launch']), tj['source']
        res = {
            'metric': 'Mrays/s, LLFF-fern render_test (HIP ray-marcher, 1008x756 frames)',
            'value': round(value, 3), 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(eff_ms, 4), 'higher_is_better': True,
            'scaling': 'strong' if by_rows or world == 1 else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: LLFF fern_lg_pretrain render_test 1008x756, DirectMPIGO '
                                   '417x353x256 grid, 256 samples/ray, rgbnet 15->64->64->3, marcher only (no SR)'
                                   + (' [REDUCED --small scene]' if args.small else ''),
                       'rays_per_frame': H * W, 'frames': args.steps * (1 if (world == 1 or by_rows) else world),
                       'streams': len(streams),
                       'parallelism': ('single GPU' if world == 1 else
                                       f'row-bands x{world} + all_gather of final pixels' if by_rows else
                                       f'frames of the pose sequence sharded over {world} GPUs (full model replica each, no collective)')},
            'frames_per_s_lr': round(rays_per_step * args.steps / (H * W) / elapsed, 2),
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                         'kernel': 'marcher call = k4_geom2_kernel<MPI> + k4_shade_kernel<MPI,64,1>',
                         'kernel_ms': round(iso_ms, 4),
                         'kernel_ms_note': 'isolated launch duration (1 stream, HIP events on the launch stream); in the '
                                           'timed region frames overlap on %d streams: %.4f ms/frame effective = %.1f GB/s '
                                           'algorithmic' % (len(streams), eff_ms, b_alg / (eff_ms * 1e-3) / 1e9),
                         'overlapped_launch_ms': round(float(np.mean(kern_ms)), 4),
                         'algorithmic_bytes_per_launch': int(b_alg),
                         'samples_per_launch': {'in_bbox': int(n_inb), 'mask': int(n_mask), 'alpha': int(n_alpha),
                                                'shaded': int(n_shade)}},
        }
    four_k = four_k_fp32 = four_k_fast = None
    if args.sr_frames > 0 and not args.small:
        four_k = four_k_frames(model, poses, rk, H, W, K, dev, args.sr_frames, world, mode='bf16x6')
        four_k_fp32 = four_k_frames(model, poses, rk, H, W, K, dev, args.sr_frames, world, mode='fp32')
        four_k_fast = four_k_frames(model, poses, rk, H, W, K, dev, args.sr_frames, world, mode='bf16x3')
    if rank == 0:
        if four_k is not None:
            res['four_k'] = four_k
            res['four_k_fp32mfma'] = four_k_fp32
            res['four_k_bf16x3'] = four_k_fast
        if world == 1 and not args.small and not args.no_extras:
            res['reference_pipeline_baseline'] = reference_pipeline_baseline(model, rays[0], rk)
            res['reference_pipeline_baseline']['speedup_of_value'] = round(
                value / res['reference_pipeline_baseline']['value'], 1)
        if world == 1 and not args.small and not args.no_extras:
            res['training_step_kernels'] = training_step_kernels(dev, rays[0], model)
        if not args.no_cpu_baseline:
            res['cpu_baseline'], parity = cpu_baseline(ck, poses[0], args.cpu_stride, model)
            if parity is not None:
                res['parity_vs_oracle'] = parity
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def four_k_frames(model, poses, rk, H, W, K, dev, n_frames, world, mode='bf16x6'):
    """BASELINE configs[2] (N=1) / configs[3] (N>1): LLFF 4K render_test = march 1008x756 + SFTNet x4 to 4032x3024,
    reference tile geometry (test_tile=510, tile_pad=10; 189 when more than 4 ranks need tiles), tiles sharded over
    the ranks, ONE all-gather of the final HR pixels per frame.  SFTNet weights: seeded default init."""
    from nerf4k_amd.lib import sr_esrnet, dvgo
    from nerf4k_amd import tile_parallel as tp
    torch.manual_seed(777)
    net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).to(dev).eval()
    net.k4_mode = mode
    tile = {1: 510, 2: 510, 4: 252}.get(world, 189)       # balanced tile counts: 4 / 4 / 12 / 24 tiles
    flop_per_px = 10377728
    px = sum((t[5] - t[4]) * (t[7] - t[6]) for t in tp.tile_geometry(H, W, tile, 10))
    march_fn, sr_fn = tp.hip_march_fn(model, rk), tp.hip_sr_fn(net)
    frames = []
    with torch.no_grad():
        for p in poses[:max(2, min(n_frames, len(poses)))]:
            frames.append(dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(p).to(dev), True, False, False, False))
        hr = tp.render_frame_tiles(frames[0], H, W, march_fn, sr_fn, tile)           # warm-up (buffers, packing)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n_frames):
            hr = tp.render_frame_tiles(frames[i % len(frames)], H, W, march_fn, sr_fn, tile, out=hr)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n_frames
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    tflops = flop_per_px * px / dt / 1e12
    base = {'frames_per_s': round(1.0 / dt, 3), 'ms_per_frame': round(dt * 1e3, 2), 'n_gpus': world, 'test_tile': tile,
            'effective_tflops': round(tflops, 2)}
    if mode == 'bf16x3':
        base['arithmetic'] = ('SR convs: 2-term bf16 splits, 3 products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; opt-in '
                              '(K4_SR_MODE=bf16x3), >= 75 dB vs the fp32 oracle (tests/test_sr_gpu.py)')
        return base
    if mode == 'fp32':
        base['arithmetic'] = 'SR convs on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains; K4_SR_MODE=fp32); peak 157.3 TFLOP/s'
        base['frac_of_fp32_mfma_peak'] = round(tflops / (157.3 * world), 4)
        return base
    peak = 2500.0 / 6 * world
    base.update({
        'output': list(hr.shape),
        'workload': ('configs[2]' if world == 1 else 'configs[3]') + ': march 1008x756 + SFTNet x4 tile_process('
                    f'{tile}, pad 10) -> 4032x3024' + (f', tiles sharded over {world} GPUs + all-gather of HR pixels' if world > 1 else ''),
        'arithmetic': 'marcher fp32; SR convs: exact 3-term bf16 splits, 6 of 9 partial products on v_mfma_f32_32x32x16_bf16, '
                      'fp32 accumulation = fp32-equivalent (dropped terms <= 2^-23 per product; >= 115 dB vs the fp32 oracle)',
        'sr_roofline': {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s (fp32-equivalent)',
                        'frac': round(tflops / peak, 4), 'flop_per_frame': flop_per_px * px,
                        'note': 'peak = 2.5 PFLOP/s dense bf16 MFMA / 6 matrix instructions per fp32-equivalent product, x n_gpus; '
                                'time includes the marcher, layout copies and the all-gather'}})
    return base


def training_step_kernels(dev, frame_rays=None, model=None, reps=5):
    """MaskedAdam + total_variation_add_grad on the LLFF k0 grid (12 x 417 x 353 x 256 fp32 =
    1.81 GB per tensor, far beyond the 256 MB infinity cache), HIP-event timed.  Algorithmic bytes per voxel:
    dense Adam 28 (param/exp_avg/exp_avg_sq read+write, grad read), TV 12 (param read once, grad read+write),
    masked Adam 4 + 24 x touched fraction (here 1 % of the voxels, in 64-voxel runs as a ray batch leaves them)."""
    from nerf4k_amd.lib import masked_adam as MA, grid as G
    shape = (1, 12, 417, 353, 256)
    n = int(np.prod(shape))
    gen = torch.Generator(device=dev).manual_seed(0)
    p = torch.randn(shape, device=dev, generator=gen)
    g = torch.randn(shape, device=dev, generator=gen)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    gs = torch.zeros(n // 64, 64, device=dev)
    touched = torch.rand(n // 64, device=dev, generator=gen) < 0.01
    gs[touched] = 1.0
    frac_touched = float(touched.float().mean())
    gs = gs.reshape(shape)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev]))

    out = {'grid': list(shape), 'voxels': n}
    for name, fn, bpe in (
            ('adam_upd', lambda: MA.adam_upd(p, g, m, v, 3, 0.9, 0.99, 1e-3, 1e-8), 28.0),
            ('masked_adam_upd_1pct', lambda: MA.masked_adam_upd(p, gs, m, v, 3, 0.9, 0.99, 1e-3, 1e-8), 4.0 + 24.0 * frac_touched),
            ('total_variation_add_grad_dense', lambda: G.total_variation_add_grad(p, g, 1e-3, 1e-3, 1e-3, True), 12.0),
            ('total_variation_add_grad_sparse_1pct', lambda: G.total_variation_add_grad(p, gs, 1e-3, 1e-3, 1e-3, False), 4.0 + 8.0 * frac_touched)):
        ms = timed(fn)
        gbs = n * bpe / (ms * 1e-3) / 1e9
        out[name] = {'ms': round(ms, 3), 'algorithmic_bytes_per_voxel': round(bpe, 2), 'achieved_GBs': round(gbs, 1),
                     'frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 4)}
    # marcher backward scatter (8f rank 1): d(DenseGrid lookup)/d(grid) for one training batch of 8192 rays x 256 samples,
    # 12 channels, fp32 hardware atomics into the same grid; bytes = grad_out + xyz read + 8 corners x 12 ch x 4 B RMW
    npts = 8192 * 256
    pts = torch.rand([npts, 3], device=dev, generator=gen) * 2 - 1
    gout = torch.randn([npts, 12], device=dev, generator=gen)
    mn, mx = torch.tensor([-1., -1., -1.], device=dev), torch.tensor([1., 1., 1.], device=dev)
    from nerf4k_amd import _native as N_
    ms = timed(lambda: N_.check(N_.lib().k4_grid_sample_3d_backward(N_.f32(gout), 12, 417, 353, 256, N_.f32(pts), N_.f32(mn), N_.f32(mx),
                                                                    npts, N_.f32(g), N_.stream()), 'grid_sample_3d_backward'))
    bpp = 12 * 4 + 12 + 8 * 12 * 4 * 2
    out['grid_sample_3d_backward_2M_random_points'] = {'ms': round(ms, 3), 'Mpoints_per_s': round(npts / ms / 1e3, 1),
                                                       'algorithmic_bytes_per_point': bpp,
                                                       'achieved_GBs': round(npts * bpp / (ms * 1e-3) / 1e9, 1)}
    if frame_rays is not None and model is not None:
        # the coherent case: 8192 rays of the frame x 256 NDC samples each (what a training batch scatters)
        ro, rd = frame_rays[0][:8192], frame_rays[1][:8192]
        t = torch.linspace(0, 1, 256, device=dev)
        pts = (ro[:, None, :] + rd[:, None, :] * t[None, :, None]).reshape(-1, 3).contiguous()
        mn, mx = model.xyz_min.float().contiguous(), model.xyz_max.float().contiguous()
        ms = timed(lambda: N_.check(N_.lib().k4_grid_sample_3d_backward(N_.f32(gout), 12, 417, 353, 256, N_.f32(pts), N_.f32(mn), N_.f32(mx),
                                                                        npts, N_.f32(g), N_.stream()), 'grid_sample_3d_backward'))
        out['grid_sample_3d_backward_8192_rays_x_256'] = {'ms': round(ms, 3), 'Mpoints_per_s': round(npts / ms / 1e3, 1),
                                                          'achieved_GBs': round(npts * bpp / (ms * 1e-3) / 1e9, 1)}
    return out


def reference_pipeline_baseline(model, rays, rk, chunk=8192, frames=2):
    """the REFERENCE's pipeline structure on this MI355X -- 8192-ray chunks (run_sr.py:121-124), one
    launch per op (sampler, maskcache_lookup, grid_sample, raw2alpha, alpha2weight, segment sum: the staged gfx950
    kernels of this package; rgbnet on rocBLAS), boolean-mask compactions with their host syncs, exactly the op
    sequence of lib/dmpigo.py:300-427.  It is what `k4_staged=True` runs; same outputs as the fused path."""
    ro, rd, vd = rays
    def frame():
        outs = [model(a, b, c, k4_staged=True, **rk) for a, b, c in zip(ro.split(chunk, 0), rd.split(chunk, 0), vd.split(chunk, 0))]
        return torch.cat([o['rgb_marched'] for o in outs])
    with torch.no_grad():
        frame()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(frames):
            frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / frames
    return {'value': round(ro.shape[0] / dt / 1e6, 3), 'unit': 'Mrays/s', 'ms_per_frame': round(dt * 1e3, 2),
            'what': 'reference op sequence, per-op launches, 8192-ray chunks, on the same GPU (k4_staged=True)'}


def cpu_baseline(ck, pose, stride, model=None):
    """The CPU oracle ("port": oracle/marcher.py on torch CPU kernels) on a bounded sample of the same frame."""
    from oracle import marcher
    from nerf4k_amd import scene
    H, W = scene.LLFF_HW
    cores = min(os.cpu_count() or 1, 32)       # torch CPU kernels stop scaling (and thrash) far below 256 threads
    torch.set_num_threads(cores)
    ro, rd, vd = marcher.get_rays_of_a_view(H, W, scene.LLFF_K, pose, ndc=True)
    sel = (slice(None, None, stride), slice(None, None, stride))
    ro, rd, vd = [x[sel].reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], ro[:8192], rd[:8192], vd[:8192],
                    **ck['render_kwargs'])                                  # warm-up
    t = time.perf_counter()
    want = marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], ro, rd, vd, **ck['render_kwargs'])
    dt = time.perf_counter() - t
    base = {'value': round(len(ro) / dt / 1e6, 5), 'unit': 'Mrays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{len(ro)} rays = every {stride}th row and column of one 1008x756 frame, 8192-ray chunks '
                      f'as run_sr.py:121-124, {dt:.1f}s of CPU work, torch {torch.__version__} CPU kernels'}
    parity = None
    if model is not None:
        # the oracle's output is at hand: use it as the CHECKER of the HIP path on the same (BASELINE-size) rays
        dev = next(model.parameters()).device
        with torch.no_grad():
            got = model(ro.to(dev), rd.to(dev), vd.to(dev), k4_img_w=(W if stride == 1 else 0),
                        **dict(ck['render_kwargs'], render_depth=True))
        d = (got['rgb_marched'].cpu().double() - want['rgb_marched'].double())
        mse = float((d ** 2).mean())
        parity = {'rays': len(ro), 'psnr_rgb_db': round(200.0 if mse == 0 else -10.0 * float(np.log10(mse)), 1),
                  'max_abs_rgb': float(d.abs().max()),
                  'max_abs_depth': float((got['depth'].cpu().double() - want['depth'].double()).abs().max()),
                  'max_abs_alphainv': float((got['alphainv_last'].cpu().double() - want['alphainv_last'].double()).abs().max()),
                  'what': 'HIP fused marcher vs the CPU oracle on the same rays (the frame timed for cpu_baseline)'}
    return base, parity


if __name__ == '__main__':
    main()
