#!/usr/bin/env python
"""Benchmark of the 4K-NeRF rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Metric (BASELINE.json): Mrays/s on LLFF-fern ``render_test``.  A *step* is one full 1008x756 frame
(762,048 rays x 256 samples) of BASELINE configs[1] -- "LLFF fern_lg_pretrain render_test at 1008x756,
1xMI355X, HIP ray-marcher only (no SR)" -- marched by the fused HIP kernels on the seeded synthetic
LLFF scene (no datasets / checkpoints exist offline), camera poses cycling through the 20-pose spiral.
Rays are resident in HBM before the timed region (the reference's own timer starts after
``get_rays_of_a_view`` too, run_sr.py:104-111).

``value`` is whole-job throughput: the frames of the timed region alternate over 3 HIP streams (the geometry kernel of frame
i+1 overlaps the shading kernel of frame i).  ``mrays_isolated`` is the stream-synchronised per-call rate (one stream, HIP
events around each call): ``roofline`` is computed on THAT duration, never on the overlapped one.

N>1 (strong scaling, the default): every frame is split in N row bands, rank r marches band r of EVERY frame and one
asynchronous ``all_gather_into_tensor`` (RCCL) per frame moves the final 5 floats per ray -- ``scaling: "strong"``,
``value`` = rays of one frame x frames / max-over-ranks time.  ``frames_sharded`` (secondary field) is the no-collective
form: rank r renders frames r, r+N, ... (weak).  The 4K pipeline (``four_k``, BASELINE configs[2]/[3]) shards the SR tiles of
ONE frame over the ranks with one all-gather of the final HR pixels (strong).

Objects on the JSON line (rank 0):
  roofline         -- dominant kernels (fused marcher call), HBM bound: algorithmic bytes per launch (device counters,
                      SURVEY.md 8d formula) / mean ISOLATED launch time; ``traffic`` = fabric bytes per launch from the committed
                      rocprofv3 PMC passes of this command (profiles/*_marcher_traffic.json, the newest).  It also carries the scalars of
                      the rest of BASELINE's metric, because the driver's record keeps this object verbatim: ``four_k_ms`` / ``four_k_fps``,
                      ``sr_frac`` + ``sr_kernel`` (decoder roofline, dominant kernel from the committed stats), ``mrays_isolated``,
                      ``rank_share_8gpu_ms`` (projection, unmeasured on hardware), ``reference_pipeline_mrays``, ``joint_iteration_ms``,
                      ``four_k_horns_ms`` (configs[3]'s scene).
  ``value``        -- rays per frame / ``ms_per_step`` (the driver can check it against its own clock); ``value_median_interval`` is the
                      round-4 statistic.
  cpu_baseline     -- the CPU oracle (oracle/marcher.py, kind "port") timed on this host's cores on one full frame.
  parity_vs_oracle -- the HIP marcher's output on that frame against the oracle's.
  four_k           -- march + SFTNet x4 (tile 510) to 4032x3024: frames/s, MFMA roofline of the decoder, PSNR of the HR
                      pixels against the oracle (march + SFTNet on the CPU) on a bounded window, and that window's CPU time
                      as ``cpu_baseline``; ``four_k_fp32mfma`` / ``four_k_bf16x3`` = the other decoder arithmetics.
  own_staged_pipeline   -- the reference's op-per-launch sequence on THIS package's staged kernels (8192-ray chunks).
  training_step_kernels -- Adam / masked Adam / TV / grid-sample backward streaming kernels vs the HBM roof.
  joint_train_step      -- BASELINE configs[4] on one GPU: ms per iteration of the joint marcher + decoder training loop.
"""
import argparse
import glob
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
CPU_THREAD_CAP = 32            # torch CPU kernels stop scaling far below a 256-thread host (measured: 64 threads run the oracle 1.9x SLOWER than 32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=80)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-stride', type=int, default=1, help='CPU baseline marches every k-th row and column')
    ap.add_argument('--small', action='store_true', help='reduced scene (debug only; never a reported number)')
    ap.add_argument('--streams', type=int, default=3, help='HIP streams the frames alternate on')
    ap.add_argument('--shard', choices=['rows', 'frames'], default='rows',
                    help='N>1 headline: "rows" = every frame split in N row bands + all_gather of the final pixels (strong '
                         'scaling, default); "frames" = rank r renders frames r, r+N, ... (no collective, weak scaling)')
    ap.add_argument('--no-extras', action='store_true', help='marcher line only: skip the side measurements (profiling runs)')
    ap.add_argument('--backend', default='nccl', help=argparse.SUPPRESS)          # gloo + --same-device: 1-GPU logic smoke test
    ap.add_argument('--same-device', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--sr-frames', type=int, default=12, help='4K frames (march + SFTNet x4, test_tile=510) timed for the '
                    'secondary frames/s figure (0 disables)')
    return ap.parse_args()


class MarcherRun:
    """The timed marcher loop for one sharding mode."""

    def __init__(self, model, poses, rk, H, W, K, dev, world, rank, by_rows, n_streams):
        from nerf4k_amd import tile_parallel as tp
        from nerf4k_amd.lib import dvgo
        self.model, self.rk, self.W, self.dev, self.world, self.by_rows = model, rk, W, dev, world, by_rows
        r0, r1, rows_per = tp.shard_rows(H, world, rank) if by_rows else (0, H, H)
        if world > 1 and not by_rows:          # rank r renders frames r, r+N, r+2N, ... of the pose sequence
            poses = [poses[(rank + i * world) % len(poses)] for i in range(len(poses))]
        self.rays = []
        with torch.no_grad():
            for p in poses:
                ro, rd, vd = dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(p).to(dev), True, False, False, False)
                self.rays.append(tuple(x[r0:r1].reshape(-1, 3).contiguous() for x in (ro, rd, vd)))
        self.n_band = n_band = (r1 - r0) * W
        slot = rows_per * W
        # all-gather buffers (double buffered): [rgb n x 3 | depth n | alphainv n] -- the marcher writes straight into them
        self.send = [torch.zeros([5 * slot], dtype=torch.float32, device=dev) for _ in range(2)]
        self.recv = [torch.empty([world * 5 * slot], dtype=torch.float32, device=dev) for _ in range(2)] if by_rows else None
        self.outs = [(b[:3 * n_band].view(n_band, 3), b[3 * slot:3 * slot + n_band], b[4 * slot:4 * slot + n_band]) for b in self.send]
        self.works = [None, None]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, n_streams))]
        model.k4_warm(stepsize=rk.get('stepsize'))      # load-time caches on the current stream, before the side streams fork
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream())

    def step(self, i, timed=None):
        b = i & 1
        st = self.streams[i % len(self.streams)]
        with torch.cuda.stream(st):
            if self.works[b] is not None:
                self.works[b].wait()                  # the collective that last read send[b] has finished
                self.works[b] = None
            ro, rd, vd = self.rays[i % len(self.rays)]
            if timed is not None:
                timed[0].record(st)
            out = self.model(ro, rd, vd, k4_img_w=self.W, k4_out=self.outs[b], k4_ws_slot=i % len(self.streams), **self.rk)
            if timed is not None:
                timed[1].record(st)
            if self.by_rows and self.world > 1:       # final pixels only; asynchronous, overlaps the next frame's march
                self.works[b] = dist.all_gather_into_tensor(self.recv[b], self.send[b], async_op=True)
        return out

    def sync(self):
        for b in range(2):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()             # all streams

    def run(self, steps, warmup):
        """-> (elapsed seconds max over ranks, mean launch-to-completion ms of a call inside the overlapped region)"""
        with torch.no_grad():
            for i in range(warmup):
                self.step(i)
            self.sync()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            for a, b in ev:                            # torch creates the HIP event at its FIRST record: outside the timed region (2 x steps creations,
                a.record(); b.record()                 # ~0.1-0.3 ms each on a slow host, made the 3-stream loop host-bound on some boxes)
            torch.cuda.synchronize()
            t_start = time.perf_counter()
            for i in range(steps):
                self.step(i, timed=ev[i])
            self.host_issue_ms_per_step = (time.perf_counter() - t_start) / max(steps, 1) * 1e3      # the host's share: issue only
            self.sync()
            elapsed = time.perf_counter() - t_start
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # per-frame completion intervals inside the overlapped region.  The S streams finish their frames in bursts, so the interval is
        # taken over one round of the streams, (end of frame i - end of frame i-S) / S: a slow round shows in the spread, the median is
        # what a long run converges to
        S = len(self.streams)
        self.frame_intervals_ms = [ev[i - S][1].elapsed_time(ev[i][1]) / S for i in range(S, steps)]
        return float(t.item()), float(np.mean([a.elapsed_time(b) for a, b in ev]))

    def isolated(self, n):
        """ISOLATED launch duration (one stream, HIP events around each call on the launch stream) and the device sample
        counters -- the duration the rocprofv3 --stats summary of `bench.py --streams 1` must agree with."""
        cnt = torch.zeros(8, dtype=torch.int64, device=self.dev)
        st0 = self.streams[0]
        iso = []
        with torch.no_grad():
            for i in range(n):
                with torch.cuda.stream(st0):
                    ro, rd, vd = self.rays[i % len(self.rays)]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st0)
                    self.model(ro, rd, vd, k4_img_w=self.W, k4_out=self.outs[0], k4_ws_slot=0, **self.rk)
                    e1.record(st0)
                    iso.append((e0, e1))
                    self.model(ro, rd, vd, k4_img_w=self.W, k4_counters=cnt, k4_out=self.outs[0], k4_ws_slot=0, **self.rk)
            self.sync()
        self.iso_ms_all = [a.elapsed_time(b) for a, b in iso]
        return float(np.mean(self.iso_ms_all)), [c / n for c in cnt.cpu().tolist()]


def marcher_source_sha1():
    """Fingerprint of the marcher's kernel source: profiles/*_marcher_traffic.json records the one it was measured on."""
    import hashlib, re
    h = hashlib.sha1()
    for f in ('k4_march.hip', 'k4_common.h'):
        src = open(os.path.join(ROOT, '4k-nerf_amd', 'csrc', f), 'r').read()
        # k4_march_mlp.h is a textual part of k4_march.hip (split off in round 6 for readability): hashed in place
        src = re.sub(r'#include "k4_march_mlp\.h"[^\n]*', lambda m: open(os.path.join(ROOT, '4k-nerf_amd', 'csrc', 'k4_march_mlp.h')).read(), src)
        src = re.sub(r'//[^\n]*', '', src)                         # the CODE: comment edits do not make a measurement stale
        h.update('\n'.join(l.strip() for l in src.splitlines() if l.strip()).encode())
    return h.hexdigest()[:12]


def newest_traffic_profile():
    """HBM-side bytes per launch: PMC passes of this command, committed under profiles/ (the newest round's file).  The file names
    the kernel source it was measured on; a mismatch with the tree being benchmarked is stamped, not hidden."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_marcher_traffic.json')))
    if not files:
        return None, None, None
    tj = json.load(open(files[-1]))
    src = os.path.relpath(files[-1], ROOT) + (' @ ' + tj['commit'] if 'commit' in tj else '')
    have = marcher_source_sha1()
    if tj.get('kernel_source_sha1') == have:
        src += ' (kernel source matches this tree)'
    else:
        src += f' (STALE: measured on kernel source {tj.get("kernel_source_sha1", "unrecorded")}, this tree is {have})'
    return int(tj['fabric_bytes_per_launch']), src, second_roof(tj)


def second_roof(tj):
    """The roof the marcher actually leans on (its HBM-side traffic is ~0.2x the algorithmic bytes): instruction issue.  From the
    committed PMC passes of this command (profiles/rNN_marcher_traffic.json, components.*.SQ_INSTS_*), two floors under separate names:
      issue_floor_ms           -- the HARDWARE bound: vector instructions x 2 clk, matrix instructions x 32 clk (v_mfma_f32_32x32x16_bf16),
                                  at 2.4 GHz over the chip's 1024 SIMDs; a kernel's floor is the larger of the two (comparable with round 4);
      issue_floor_measured_ms  -- vector instructions priced at the BEST rate a gfx950 SIMD was measured at, 1.34 ns per wave-instruction
                                  (v_fma_f32 at 4 waves per SIMD, 1.88 cycles at the 1.40 GHz the chip holds there; the pk / cvt / max / shift
                                  classes the shading kernel is made of: 1.96-2.06 ns; profiles/r05_valu_issue_rate.md) -- a microbenchmark
                                  figure, not a hardware limit (round 5 reported only this one).
    l2_bytes = L1 -> L2 read + write requests x 64 B."""
    comp = tj.get('components', {})
    if not any('SQ_INSTS_VALU' in v for v in comp.values()):
        return None
    out, tot, tot_hw = {}, 0.0, 0.0
    for k, v in comp.items():
        valu = v.get('SQ_INSTS_VALU', 0.0) * 1.34e-9 / 1024 * 1e3
        valu_hw = v.get('SQ_INSTS_VALU', 0.0) * 2 / 1024 / 2.4e9 * 1e3
        mfma = v.get('SQ_INSTS_MFMA', 0.0) * 32 / 1024 / 2.4e9 * 1e3
        out[k] = {'valu_ms_measured_rate': round(valu, 4), 'valu_ms_2clk_2.4GHz': round(valu_hw, 4), 'mfma_ms': round(mfma, 4)}
        tot += max(valu, mfma)
        tot_hw += max(valu_hw, mfma)
    l2 = sum((v.get('TCP_TCC_READ_REQ_sum', 0.0) + v.get('TCP_TCC_WRITE_REQ_sum', 0.0)) * 64 for v in comp.values())
    return {'issue_floor_ms': round(tot_hw, 4), 'issue_floor_measured_ms': round(tot, 4), 'per_kernel': out, 'l2_bytes': int(l2) if l2 else None,
            'note': 'issue_floor_ms = sum over the call\'s kernels of max(vector instr x 2 clk, matrix instr x 32 clk) / 2.4 GHz / 1024 SIMDs (hardware bound); '
                    'issue_floor_measured_ms prices vector instructions at the measured best case 1.34 ns (profiles/r05_valu_issue_rate.md)'}


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
    from nerf4k_amd.lib import utils

    _native.lib()
    t0 = time.time()
    ck = scene.make_llff_checkpoint(num_voxels=96 * 96 * 64, mpi_depth=64) if args.small else scene.make_llff_checkpoint()
    model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
    rk = ck['render_kwargs']
    H, W = scene.LLFF_HW
    K = scene.LLFF_K
    poses = scene.llff_spiral_poses()
    if rank == 0:
        print(f'[bench] scene ready in {time.time() - t0:.1f}s: world_size={model.world_size.tolist()} '
              f'k0_ch={model.k0_dim} rays/frame={H * W} dist_world={dist.get_world_size() if world > 1 else 1}', file=sys.stderr)

    by_rows = world > 1 and args.shard == 'rows'
    run = MarcherRun(model, poses, rk, H, W, K, dev, world, rank, by_rows, args.streams)
    elapsed, overlapped_ms = run.run(args.steps, args.warmup)
    nf = min(args.steps, len(run.rays))
    iso_ms, counts = run.isolated(nf)
    n_inb, n_mask, n_alpha, n_shade, n_behind = counts[:5]
    n_band = run.n_band
    secondary = None
    if world > 1 and not args.no_extras:       # the other sharding mode, as a secondary field
        other = MarcherRun(model, poses, rk, H, W, K, dev, world, rank, not by_rows, args.streams)
        e2, _ = other.run(args.steps, args.warmup)
        rays2 = H * W * (1 if not by_rows else world)
        secondary = {'mode': 'rows+all_gather (strong)' if not by_rows else 'frames of the pose sequence per rank, no collective (weak)',
                     'value': round(rays2 * args.steps / e2 / 1e6, 3), 'unit': 'Mrays/s', 'ms_per_step': round(e2 / args.steps * 1e3, 4)}
        del other

    if rank == 0:
        rays_per_step = H * W * (1 if (world == 1 or by_rows) else world)     # frames mode: every rank renders a frame per step
        # value: whole-job throughput of the timed region = rays / ms_per_step, the figure the driver's own clock can check (round 4 led with
        # the median frame-completion interval; it stays as a side key: value_median_interval)
        med_ms = float(np.median(run.frame_intervals_ms)) if run.frame_intervals_ms else elapsed / args.steps * 1e3
        eff_ms = elapsed / args.steps * 1e3            # per-frame time of the timed region (frames overlap on the streams)
        value = rays_per_step / (eff_ms * 1e-3) / 1e6
        b_alg = n_band * 56 + n_inb * 1 + n_mask * 32 + n_shade * 8 * model.k0_dim * 4
        achieved = b_alg / (iso_ms * 1e-3) / 1e9
        traffic, traffic_src, roof2 = (None, None, None) if args.small else newest_traffic_profile()
        res = {
            'metric': 'Mrays/s, LLFF-fern render_test (HIP ray-marcher, 1008x756 frames)',
            'value': round(value, 3), 'value_basis': 'rays per frame / ms_per_step (mean over the timed region; frames pipelined on 3 HIP streams)',
            'value_median_interval': round(rays_per_step / (med_ms * 1e-3) / 1e6, 3),
            'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(eff_ms, 4), 'ms_per_step_median': round(float(np.median(run.frame_intervals_ms)), 4) if run.frame_intervals_ms else None,
            'ms_per_step_p90': round(float(np.percentile(run.frame_intervals_ms, 90)), 4) if run.frame_intervals_ms else None,
            'higher_is_better': True,
            'scaling': 'strong' if by_rows or world == 1 else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: LLFF fern_lg_pretrain render_test 1008x756, DirectMPIGO '
                                   '417x353x256 grid, 256 samples/ray, rgbnet 15->64->64->3, marcher only (no SR)'
                                   + (' [REDUCED --small scene]' if args.small else ''),
                       'rays_per_frame': H * W, 'frames': args.steps * (1 if (world == 1 or by_rows) else world),
                       'streams': len(run.streams), 'dist_world_size': dist.get_world_size() if world > 1 else 1,
                       'parallelism': ('single GPU' if world == 1 else
                                       f'ONE frame split in {world} row bands + RCCL all_gather of the final pixels' if by_rows else
                                       f'frames of the pose sequence sharded over {world} GPUs (full model replica each, no collective)')},
            'mrays_isolated': round(n_band / (iso_ms * 1e-3) / 1e6, 3),
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'kernel': 'k4_geom3 + k4_order + k4_shade<MPI,64,1,b2,FAST>, isolated call (HIP events)',
                         'kernel_ms': round(iso_ms, 4)},
            # what `roofline` is made of (the driver's record keeps ~20 scalar entries of `roofline`: everything nested lives here)
            'roofline_detail': {'kernel_ms_median': round(float(np.median(run.iso_ms_all)), 4), 'overlapped_launch_ms': round(overlapped_ms, 4),
                                'host_issue_ms_per_step': round(getattr(run, 'host_issue_ms_per_step', 0.0), 4),
                                'traffic_source': traffic_src,
                                'second_roof': None if roof2 is None else dict(roof2, frac_of_issue_floor=round(roof2['issue_floor_ms'] / iso_ms, 4),
                                                                               frac_of_measured_floor=round(roof2['issue_floor_measured_ms'] / iso_ms, 4)),
                                'algorithmic_bytes_per_launch': int(b_alg),
                                'samples_per_launch': {'in_bbox': int(n_inb), 'mask': int(n_mask), 'alpha': int(n_alpha), 'shaded': int(n_shade),
                                                       'alpha_behind_stop': int(n_behind)},
                                'behind_stop_fraction_of_alpha': round(n_behind / max(n_alpha, 1), 4)},
        }
        if secondary is not None:
            res['frames_sharded' if by_rows else 'rows_sharded'] = secondary
    four_k = four_k_fp32 = four_k_fast = four_k_alt = four_k_b6 = None
    if args.sr_frames > 0 and not args.small:
        keep = {}
        from nerf4k_amd.lib import sr_esrnet as _sr
        default_mode = _sr.DEFAULT_MODE
        other_mode = 'f16x3'                  # the same products with every consumer splitting fp32 activations per tile: what 'f16x3p' falls back to
        four_k = four_k_frames(model, ck, poses, rk, H, W, K, dev, args.sr_frames, world, rank, mode=default_mode,
                               check=not args.no_cpu_baseline and world == 1, keep=keep)
        if not args.no_extras:
            few = max(3, args.sr_frames // 4)         # the side arithmetics: a few frames each (they are context, not the headline)
            four_k_alt = four_k_frames(model, ck, poses, rk, H, W, K, dev, few, world, rank, mode=other_mode, keep=keep)
            four_k_b6 = four_k_frames(model, ck, poses, rk, H, W, K, dev, few, world, rank, mode='bf16x6', keep=keep)
            four_k_fp32 = four_k_frames(model, ck, poses, rk, H, W, K, dev, 3, world, rank, mode='fp32', keep=keep)
            four_k_fast = four_k_frames(model, ck, poses, rk, H, W, K, dev, few, world, rank, mode='bf16x3', keep=keep)
        keep.clear()
    joint_dp = None
    if world > 1 and not args.small and not args.no_extras:
        # BASELINE configs[4]: one 64x64 patch per rank, gradients exchanged over RCCL (every rank runs it; rank 0 reports)
        full_rays = None
        with torch.no_grad():
            from nerf4k_amd.lib import dvgo as _dv
            full_rays = [x.reshape(-1, 3).contiguous() for x in
                         _dv.get_rays_of_a_view(H, W, K, torch.from_numpy(poses[0]).to(dev), True, False, False, False)]
        joint_dp = _side(joint_train_step, ck, full_rays, H, W, dev, 8, world, rank)
    if rank == 0:
        if joint_dp is not None:
            res['joint_train_step'] = joint_dp
        if four_k is not None:
            res['four_k'] = four_k
        if four_k_fp32 is not None:
            res['four_k_fp32mfma'], res['four_k_bf16x3'], res['four_k_' + other_mode], res['four_k_bf16x6'] = four_k_fp32, four_k_fast, four_k_alt, four_k_b6
        if world == 1 and not args.small and not args.no_extras:
            # side measurements (single process, no collectives): a failure in one of them must not take the headline line with it
            res['own_staged_pipeline'] = _side(own_staged_pipeline, model, run.rays[0], rk)
            res['training_step_kernels'] = _side(training_step_kernels, dev, run.rays[0], model)
            res['joint_train_step'] = _side(joint_train_step, ck, run.rays[0], H, W, dev)
            res['reference_pipeline_rocm'] = _side(reference_pipeline_rocm, ck, run.rays[0], dev)
            res['dvgo_config0'] = _side(dvgo_config0, dev, not args.no_cpu_baseline)
            res['scene_sweep'] = _side(scene_sweep, dev, H, W, K, poses)
            if args.sr_frames > 0:
                res['four_k_horns'] = _side(four_k_horns, dev, poses, H, W, K, world, rank)
        # the driver's parsed record keeps `roofline` and `config` verbatim and only the NAMES of the other keys: the scalars of the other half of
        # BASELINE's metric (4K frames/s), of the projections and of the second roof ride inside `roofline` -- at most 20 scalar entries, the
        # most important first (round 5's line lost its tail: the record keeps ~24 entries)
        rl = res['roofline']
        rl['mrays_isolated'] = res['mrays_isolated']
        if isinstance(four_k, dict):
            rl['four_k_ms'] = four_k.get('ms_per_frame')
            rl['sr_frac'] = (four_k.get('sr_roofline') or {}).get('frac')
            if 'psnr_vs_oracle_db' in four_k:
                rl['four_k_psnr_vs_oracle_db'] = four_k['psnr_vs_oracle_db']
            if isinstance(four_k.get('rank_share_8gpu'), dict):
                rl['rank_share_8gpu_ms'] = four_k['rank_share_8gpu'].get('ms')
                res['roofline_detail']['rank_share_8gpu_pipelined_ms'] = four_k['rank_share_8gpu'].get('pipelined_ms')
            res['roofline_detail']['sr_kernel'] = sr_kernel_from_profiles()
            res['roofline_detail']['four_k_arith'] = four_k.get('arith') or default_mode
        if isinstance(res.get('joint_train_step'), dict):
            rl['joint_iteration_ms'] = res['joint_train_step'].get('ms_per_iteration')
            res['roofline_detail']['joint_iteration_after_tv_before_ms'] = res['joint_train_step'].get('ms_per_iteration_after_tv_before')
        if roof2 is not None:
            rl['issue_floor_ms'] = roof2['issue_floor_ms']
            rl['frac_of_issue_floor'] = round(roof2['issue_floor_ms'] / iso_ms, 4)
        srk = res['roofline_detail'].get('sr_kernel')
        if isinstance(srk, dict):
            rl['sr_kernel_avg_us'] = srk.get('avg_us')
        if isinstance(res.get('dvgo_config0'), dict) and isinstance(res['dvgo_config0'].get('800x800'), dict):
            rl['dvgo_800_ms'] = res['dvgo_config0']['800x800'].get('ms')
        if isinstance(res.get('reference_pipeline_rocm'), dict):
            rl['reference_pipeline_mrays'] = res['reference_pipeline_rocm'].get('value')
        if isinstance(res.get('four_k_horns'), dict):
            res['roofline_detail']['four_k_horns_ms'] = res['four_k_horns'].get('ms_per_frame')
            res['roofline_detail']['rank_share_8gpu_horns_ms'] = (res['four_k_horns'].get('rank_share_8gpu') or {}).get('ms')
        if not args.no_cpu_baseline and world == 1:          # the CPU legs run at N = 1 only (bench contract): at N > 1 the other ranks would idle behind rank 0's host work
            res['cpu_baseline'], parity = cpu_baseline(ck, poses[0], args.cpu_stride, model)
            if parity is not None:
                res['parity_vs_oracle'] = parity
                rl['parity_psnr_db'] = parity.get('psnr_rgb_db')
        assert len(rl) <= 20 and all(not isinstance(v, (dict, list)) for v in rl.values()), 'roofline: at most 20 scalar entries'
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _side(fn, *a):
    try:
        return fn(*a)
    except Exception as e:
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {'error': f'{type(e).__name__}: {str(e)[:200]}'}


def _cpu_threads():
    cores = os.cpu_count() or 1
    used = min(cores, CPU_THREAD_CAP)
    torch.set_num_threads(used)
    return cores, used


def four_k_frames(model, ck, poses, rk, H, W, K, dev, n_frames, world, rank, mode='bf16x6', check=False, keep=None):
    """BASELINE configs[2] (N=1) / configs[3] (N>1): LLFF 4K render_test = march 1008x756 + SFTNet x4 to 4032x3024,
    reference tile geometry (test_tile=510, tile_pad=10; 252 / 189 when 4 / 8 ranks need tiles), tiles sharded over
    the ranks, ONE all-gather of the final HR pixels per frame.  SFTNet weights: seeded default init."""
    from nerf4k_amd.lib import sr_esrnet, dvgo
    from nerf4k_amd import tile_parallel as tp
    torch.manual_seed(777)
    net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).to(dev).eval()
    if mode is not None:
        net.k4_mode = mode
    mode = net.k4_mode
    tile = {1: 510, 2: 510, 4: 252}.get(world, 189)       # balanced tile counts: 4 / 4 / 12 / 24 tiles
    flop_per_px = 10377728
    tiles = tp.tile_geometry(H, W, tile, 10)
    px = sum((t[5] - t[4]) * (t[7] - t[6]) for t in tiles)
    march_fn, sr_fn = tp.hip_march_fn(model, rk), tp.hip_sr_fn(net)
    frames = []
    with torch.no_grad():
        for p in poses[:max(2, min(n_frames, len(poses)))]:
            frames.append(dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(p).to(dev), True, False, False, False))
        hr = tp.render_frame_tiles(frames[0], H, W, march_fn, sr_fn, tile)           # warm-up (buffers, packing)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_frames + 1)]
        t = time.perf_counter()
        ev[0].record()
        for i in range(n_frames):
            hr = tp.render_frame_tiles(frames[i % len(frames)], H, W, march_fn, sr_fn, tile, out=hr)
            ev[i + 1].record()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n_frames
        per_frame = [ev[i].elapsed_time(ev[i + 1]) for i in range(n_frames)]          # frame-to-frame intervals on the launch stream
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    tflops = flop_per_px * px / dt / 1e12
    base = {'frames_per_s': round(1.0 / dt, 3), 'ms_per_frame': round(dt * 1e3, 2), 'frames_timed': n_frames,
            'ms_per_frame_median': round(float(np.median(per_frame)), 2), 'ms_per_frame_p90': round(float(np.percentile(per_frame, 90)), 2),
            'n_gpus': world, 'test_tile': tile,
            'effective_tflops': round(tflops, 2), 'gather': 'fp32 HR pixels, 146 MB per frame (bit-identical to the single-GPU frame)'}
    if mode == 'f16x3p':
        base['k4_p16_reruns'] = int(net._k4.get('p16_reruns', 0))          # windows redone on the per-tile kernel (overflow words) over all frames above
    if world > 1:
        # the same job with 8-bit pixels on the wire (render_frame_tiles(out_dtype=torch.uint8): every rank applies the reference's
        # to8b rule to its own tiles before the all-gather, 36.6 MB per frame -- what run_sr.py finally writes to disk)
        with torch.no_grad():
            hr8 = tp.render_frame_tiles(frames[0], H, W, march_fn, sr_fn, tile, out_dtype=torch.uint8)
            dist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(n_frames):
                hr8 = tp.render_frame_tiles(frames[i % len(frames)], H, W, march_fn, sr_fn, tile, out=hr8, out_dtype=torch.uint8)
            dist.barrier()
            torch.cuda.synchronize()
            t8 = torch.tensor([(time.perf_counter() - t) / n_frames], dtype=torch.float64, device=dev)
        dist.all_reduce(t8, op=dist.ReduceOp.MAX)
        base['uint8_gather'] = {'ms_per_frame': round(float(t8.item()) * 1e3, 2), 'frames_per_s': round(1.0 / float(t8.item()), 3),
                                'bytes_per_frame': int(hr8.numel())}
    with torch.no_grad():                                 # the frame the arithmetics are compared on: pose 0, whatever number of frames was timed
        hr_cmp = tp.render_frame_tiles(frames[0], H, W, march_fn, sr_fn, tile) if world == 1 else hr
    primary = keep is not None and 'hr' not in keep       # the first arithmetic timed is the default one: the others are compared with its frame
    if keep is not None and primary:
        keep['hr'], keep['mode'] = hr_cmp.clone(), mode
    if not primary:
        per_product = {'bf16x3': 3, 'f16x3': 3, 'f16x3p': 3, 'bf16x6': 6}.get(mode)
        base['arithmetic'] = {
            'f16x3p': 'SR 3x3 convs: the f16x3 products on activations PRE-SPLIT by their producer under calibrated per-tensor scales (K4_SR_MODE=f16x3p)',
            'bf16x3': 'SR 3x3 convs: the two leading bf16 split terms, 3 of the 6 MFMA products (opt-in K4_SR_MODE=bf16x3; >= 75 dB vs the fp32 oracle in tests/test_sr_gpu.py)',
            'f16x3': 'SR 3x3 convs: 2-term fp16 splits with power-of-two scaling, 3 products on v_mfma_f32_32x32x16_f16 (K4_SR_MODE=f16x3; >= 110 dB vs fp32 in tests/test_sr_gpu.py)',
            'bf16x6': 'SR convs: exact 3-term bf16 splits, 6 products on v_mfma_f32_32x32x16_bf16 (K4_SR_MODE=bf16x6)',
            'fp32': 'SR convs on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains; K4_SR_MODE=fp32)'}[mode]
        if per_product:
            base[f'frac_of_{mode}_floor'] = round(tflops / (2500.0 / per_product * world), 4)
        else:
            base['frac_of_fp32_mfma_peak'] = round(tflops / (157.3 * world), 4)
        if keep is not None and 'hr' in keep:      # same pose, same weights: the whole 4032x3024 frame against the default arithmetic's
            d = (hr_cmp.double() - keep['hr'].double())
            mse = float((d ** 2).mean())
            base[f'psnr_vs_{keep["mode"]}_frame_db'] = round(200.0 if mse == 0 else -10.0 * float(np.log10(mse)), 1)
            base[f'max_abs_vs_{keep["mode"]}_frame'] = float(d.abs().max())
        return base
    per_product = 3 if mode in ('f16x3', 'f16x3p') else 6
    peak = 2500.0 / per_product * world
    base.update({
        'output': list(hr.shape), 'scaling': 'strong',
        'workload': ('configs[2]' if world == 1 else 'configs[3]') + ': march 1008x756 + SFTNet x4 tile_process('
                    f'{tile}, pad 10) -> 4032x3024' + (f', tiles of ONE frame sharded over {world} GPUs + RCCL all_gather of HR pixels' if world > 1 else ''),
        'arithmetic': ('marcher fp32; SR convs: exact 3-term bf16 splits, 6 of 9 partial products on v_mfma_f32_32x32x16_bf16, '
                       'fp32 accumulation (fp32-equivalent, dropped terms <= 2^-23 per product)') if mode == 'bf16x6' else
                      ('marcher fp32; SR 3x3 convs: 2-term fp16 splits with power-of-two scaling (22 significant bits per operand), 3 of 4 '
                       'partial products on v_mfma_f32_32x32x16_f16, fp32 accumulation (~2^-21 per product, NOT bit-for-bit fp32; the '
                       'strictly fp32-equivalent decoder is four_k_bf16x6); 1x1 / SFT / conv_last layers: exact 3-term bf16 splits'
                       + ('; dense-block / upsampling activations are written PRE-SPLIT (fp16 hi + lo) by their producer under one '
                          'calibrated power-of-two scale per tensor and go global -> LDS by DMA; the SFT layers behind conv4 / conv5 of a '
                          'dense block run in that 3x3 layer\'s epilogue in the same 3-product fp16 arithmetic (K4_SR_SFT_FUSE=0: own launches, '
                          'bf16 splits); a window that leaves fp16 range is redone on the per-tile kernel (k4_p16_reruns counts them)'
                          if mode == 'f16x3p' else '')),
        'sr_roofline': {'bound': 'mfma', 'achieved': round(tflops, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s (fp32-equivalent)',
                        'frac': round(tflops / peak, 4), 'flop_per_frame': flop_per_px * px,
                        # the other roof: every layer reads its inputs and writes its outputs ONCE as fp32 (no halo, no re-reads): channel
                        # accesses per padded LR pixel x 4 bytes -- conv_first 67, CondNet 417, 5 RRDBs x (3 dense blocks x 1152 + 224),
                        # sftbody 160, conv_body 192, up1 320, up2 1280, conv_hr 2048, conv_last 1072
                        'algorithmic_bytes_per_frame': 23956 * 4 * px,
                        'hbm_floor_ms': round(23956 * 4 * px / (HBM_PEAK_GBS * 1e9 * world) * 1e3, 2),
                        'mfma_floor_ms': round(flop_per_px * px / (peak * 1e12) * 1e3, 2),
                        'note': f'peak = 2.5 PFLOP/s dense bf16|fp16 / {per_product} MFMA per product x n_gpus; time includes the '
                                'marcher, layout copies and the all-gather.  With fp32 activations the decoder sits where the two roofs '
                                'meet for the 3-product arithmetic (hbm_floor_ms vs mfma_floor_ms).  The peak is quoted at 2.4 GHz; '
                                'the 3x3 kernels of this frame were measured at 1,624 MHz (s_memtime / s_memrealtime per wave on an '
                                'instrumented build, profiles/r04_p16_producer_waves_not_faster.md section 3): frac_at_measured_clock '
                                'prices the same time against peak x 1624 / 2400',
                        'frac_at_measured_clock': round(tflops / (peak * 1624.0 / 2400.0), 4)}})
    if world == 1:
        # The N-GPU job's critical path, measured on THIS GPU (no multi-GPU node was available to the builder: every figure below is a
        # PROJECTION, unmeasured on hardware): for N = 2 / 4 / 8 and a few tile sizes, the heaviest rank's share of the frame (its tiles'
        # padded windows marched + decoded exactly as tile_parallel does) + a device-to-device copy of its all-gather slot as a
        # stand-in for the collective.  Parity of a tile size is checked against the oracle with the SAME tile size (tests).
        proj = {}
        with torch.no_grad():
            for n, cands in ((2, (510, 378)), (4, (252, 378)), (8, (189, 168, 126, 252))):
                rows = []
                for ts in cands:
                    tl = tp.tile_geometry(H, W, ts, 10)
                    owned = tp.assign_tiles(tl, n)
                    area = [sum((tl[i][5] - tl[i][4]) * (tl[i][7] - tl[i][6]) for i in o) for o in owned]
                    r = max(range(n), key=lambda q: area[q])
                    sub = _SubsetGeometry(tl, owned[r])
                    slot = max(sum((tl[i][1] - tl[i][0]) * (tl[i][3] - tl[i][2]) * 16 for i in o) for o in owned)
                    src, dst = torch.empty([3, slot], device=dev), torch.empty([3, slot], device=dev)
                    sub.render(frames[0], march_fn, sr_fn)
                    torch.cuda.synchronize()
                    evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
                    evs[0].record()
                    for q in range(8):
                        sub.render(frames[0], march_fn, sr_fn)
                        evs[q + 1].record()
                    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    g0.record(); dst.copy_(src); g1.record()
                    torch.cuda.synchronize()
                    ms = float(np.median([evs[q].elapsed_time(evs[q + 1]) for q in range(8)]))
                    gms = g0.elapsed_time(g1)
                    rows.append({'tile_size': ts, 'tiles_of_heaviest_rank': len(owned[r]), 'halo_factor': round(area[r] * n / (H * W), 3),
                                 'share_ms_median': round(ms, 2), 'gather_standin_ms': round(gms, 3),
                                 'projected_speedup': round(dt * 1e3 / (ms + gms), 2)})
                best = max(rows, key=lambda q: q['projected_speedup'])
                proj[str(n)] = {'best': best, 'candidates': rows}
        base['rank_share_projection'] = dict(proj, note='UNMEASURED ON HARDWARE: one GPU timing the heaviest rank\'s tiles + a D2D copy of its '
                                                        'all-gather slot; speedup = this GPU\'s whole frame (tile 510) / that')
        b8 = proj['8']['best']
        base['rank_share_8gpu'] = {'ms': b8['share_ms_median'], 'tiles': b8['tiles_of_heaviest_rank'], 'tile_size': b8['tile_size'],
                                   'projected_speedup_before_gather': round(dt * 1e3 / b8['share_ms_median'], 2)}
        # the same share with frame i+1's march issued BEFORE frame i's decode (tile_parallel.march_frame_tiles / decode_frame_tiles): at tile 168 a
        # layer of the rank's windows is ~1 round of workgroups and leaves CUs idle, the next frame's march runs under it (verdict r05 item 4b)
        with torch.no_grad():
            tl = tp.tile_geometry(H, W, b8['tile_size'], 10)
            owned = tp.assign_tiles(tl, 8)
            area = [sum((tl[i][5] - tl[i][4]) * (tl[i][7] - tl[i][6]) for i in o) for o in owned]
            sub = _SubsetGeometry(tl, owned[max(range(8), key=lambda q: area[q])])
            nf = len(frames)
            st_next = sub.march(frames[0], march_fn)
            for q in range(3):                                                # warm-up of the pipelined order
                st_cur, st_next = st_next, sub.march(frames[(q + 1) % nf], march_fn)
                sub.decode(st_cur, sr_fn)
            torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
            evs[0].record()
            for q in range(12):
                st_cur, st_next = st_next, sub.march(frames[(q + 1) % nf], march_fn)
                sub.decode(st_cur, sr_fn)
                evs[q + 1].record()
            torch.cuda.synchronize()
            pms = float(np.median([evs[q].elapsed_time(evs[q + 1]) for q in range(12)]))
        base['rank_share_8gpu']['pipelined_ms'] = round(pms, 2)
        base['rank_share_8gpu']['projected_speedup_pipelined'] = round(dt * 1e3 / (pms + b8['gather_standin_ms']), 2)
    if check and rank == 0:
        base.update(four_k_parity_and_cpu(ck, net, poses[(n_frames - 1) % len(frames)], hr, tiles, H, W))
    return base


def four_k_horns(dev, poses, H, W, K, world, rank, n_frames=6):
    """BASELINE configs[3]'s scene: "horns" (the same generator, seed 778; lib/load_data.py:32-34 gives it 8 test views) through the whole 4K
    path -- march + SFTNet x4 at tile 510 on this GPU, and the heaviest rank's share of the 8-GPU job (UNMEASURED ON HARDWARE, as in four_k)."""
    from nerf4k_amd import scene
    from nerf4k_amd.lib import utils
    ck = scene.make_llff_checkpoint(seed=778)
    model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
    views = poses[::max(1, len(poses) // 8)][:8]
    out = four_k_frames(model, ck, views, ck['render_kwargs'], H, W, K, dev, n_frames, world, rank, mode=None, keep={})
    keep = {k: out[k] for k in ('frames_per_s', 'ms_per_frame', 'ms_per_frame_median', 'frames_timed', 'test_tile', 'k4_p16_reruns', 'rank_share_8gpu') if k in out}
    if 'rank_share_projection' in out:
        keep['rank_share_projection_best'] = {n: v['best'] for n, v in out['rank_share_projection'].items() if isinstance(v, dict) and 'best' in v}
    keep['scene'] = 'make_llff_checkpoint(seed=778), 8 views of the spiral'
    del model, ck
    torch.cuda.empty_cache()
    return keep


def sr_kernel_from_profiles():
    """Dominant decoder kernel, name + average duration from the newest committed rocprofv3 kernel stats (profiles/rNN_sr_kernel_stats.csv)."""
    import csv
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_sr_kernel_stats.csv')))
    if not files:
        return None
    try:
        rows = [r for r in csv.reader(open(files[-1])) if r and r[0] != 'Name']
        r = max(rows, key=lambda q: float(q[2]))
        return {'name': r[0].split('(')[0].replace('void ', ''), 'calls': int(r[1]), 'avg_us': round(float(r[3]) / 1e3, 1),
                'share_of_gpu_time': round(float(r[2]) / sum(float(q[2]) for q in rows), 3), 'source': os.path.relpath(files[-1], ROOT)}
    except Exception:       # noqa: BLE001
        return None


class _SubsetGeometry:
    """March + decode a subset of tiles on the caller's GPU exactly as tile_parallel does for one rank (no gather)."""

    def __init__(self, tiles, mine):
        self.tiles = [tiles[i] for i in mine]

    def render(self, rays, march_fn, sr_fn):
        return self.decode(self.march(rays, march_fn), sr_fn)

    def decode(self, st, sr_fn):
        cur = torch.cuda.current_stream(st[2])
        for ev in st[3]:
            cur.wait_event(ev)
        return sr_fn.k4_multi(st[0], st[1])           # one grouped launch per layer over this rank's windows

    def march(self, rays, march_fn):
        """This rank's windows marched on the pool's streams (they wait for what is queued on the current stream NOW: issued before the
        previous frame's decode, the march runs under it)."""
        from nerf4k_amd import tile_parallel as tp
        dev = rays[0].device
        pool = tp._stream_pool(dev, min(4, len(self.tiles)))
        cur = torch.cuda.current_stream(dev)
        for st in pool:
            st.wait_stream(cur)
        imgs, conds = [], []
        for j, (y0, y1, x0, x1, yp0, yp1, xp0, xp1) in enumerate(self.tiles):
            with torch.cuda.stream(pool[j % len(pool)]):
                ro, rd, vd = [r[yp0:yp1, xp0:xp1].reshape(-1, 3).contiguous() for r in rays]
                rgb, depth = march_fn(ro, rd, vd, xp1 - xp0, slot=j % len(pool))
                rgb.record_stream(cur)
                depth.record_stream(cur)
                hh, ww = yp1 - yp0, xp1 - xp0
                imgs.append(rgb.reshape(hh, ww, 3).permute(2, 0, 1).unsqueeze(0))
                conds.append(depth.reshape(1, 1, hh, ww))
        evs = []
        for st in pool:
            ev = torch.cuda.Event()
            ev.record(st)
            evs.append(ev)
        return imgs, conds, dev, evs


def four_k_parity_and_cpu(ck, net, pose, hr, tiles, H, W):
    """The oracle as CHECKER of the 4K path and as its CPU baseline, on a bounded sample: the smallest tile window of the frame
    is marched and decoded by oracle/marcher.py + oracle/sr.py (CPU) and compared with the HIP frame's pixels of that tile."""
    from oracle import marcher, sr as osr
    from nerf4k_amd import scene
    cores, used = _cpu_threads()
    i = min(range(len(tiles)), key=lambda q: (tiles[q][5] - tiles[q][4]) * (tiles[q][7] - tiles[q][6]))
    y0, y1, x0, x1, yp0, yp1, xp0, xp1 = tiles[i]
    hh, ww = yp1 - yp0, xp1 - xp0
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ro, rd, vd = marcher.get_rays_of_a_view(H, W, scene.LLFF_K, pose, ndc=True)
    win = [r[yp0:yp1, xp0:xp1].reshape(-1, 3).contiguous() for r in (ro, rd, vd)]
    t = time.perf_counter()
    o = marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], *win, **dict(ck['render_kwargs'], render_depth=True))
    t_march = time.perf_counter() - t
    t = time.perf_counter()
    want = osr.sftnet_forward(sd, o['rgb_feature'].reshape(hh, ww, 3).permute(2, 0, 1).unsqueeze(0), o['depth'].reshape(1, 1, hh, ww))
    t_sr = time.perf_counter() - t
    s = 4
    oy, ox = (y0 - yp0) * s, (x0 - xp0) * s
    want = want[0, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
    got = hr[0, :, y0 * s:y1 * s, x0 * s:x1 * s].cpu()
    d = (got.double() - want.double())
    mse = float((d ** 2).mean())
    px_all = sum((q[5] - q[4]) * (q[7] - q[6]) for q in tiles)
    frac = hh * ww / px_all
    return {'psnr_vs_oracle_db': round(200.0 if mse == 0 else -10.0 * float(np.log10(mse)), 1), 'max_abs_vs_oracle': float(d.abs().max()),
            'parity_sample': f'tile {i} of {len(tiles)} ({ww}x{hh} LR window -> {(x1 - x0) * s}x{(y1 - y0) * s} HR pixels), oracle march + SFTNet on the CPU',
            'cpu_baseline': {'value': round(frac / (t_march + t_sr), 5), 'unit': '4K frames/s', 'cores': cores, 'threads_used': used,
                             'kind': 'port',
                             'sample': f'that window = {frac:.3f} of the frame\'s padded LR pixels: {t_march:.1f}s march + {t_sr:.1f}s SFTNet on the '
                                       f'CPU, extrapolated to the frame; torch {torch.__version__} CPU kernels'}}


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

    out = {'grid': list(shape)}
    for name, fn, bpe in (
            ('adam_upd', lambda: MA.adam_upd(p, g, m, v, 3, 0.9, 0.99, 1e-3, 1e-8), 28.0),
            ('masked_adam_upd_1pct', lambda: MA.masked_adam_upd(p, gs, m, v, 3, 0.9, 0.99, 1e-3, 1e-8), 4.0 + 24.0 * frac_touched),
            ('tv_add_grad_dense', lambda: G.total_variation_add_grad(p, g, 1e-3, 1e-3, 1e-3, True), 12.0),
            ('tv_add_grad_sparse_1pct', lambda: G.total_variation_add_grad(p, gs, 1e-3, 1e-3, 1e-3, False), 4.0 + 8.0 * frac_touched),
            # the dense term WRITTEN (dense_mode 2: param read, grad written, no zero-fill in front): what the joint loop runs ahead of the backward pass
            ('tv_term_written_dense', lambda: G.total_variation_add_grad(p, g, 1e-3, 1e-3, 1e-3, 'write'), 8.0)):
        ms = timed(fn)
        gbs = n * bpe / (ms * 1e-3) / 1e9
        out[name] = {'ms': round(ms, 3), 'B_per_voxel': round(bpe, 2), 'GBs': round(gbs, 1), 'frac_hbm': round(gbs / HBM_PEAK_GBS, 3)}
    # marcher backward scatter (8f rank 1): d(DenseGrid lookup)/d(grid) for one training batch of 8192 rays x 256 samples,
    # 12 channels, into the same grid; bytes = grad_out + xyz read + 8 corners x 12 ch x 4 B RMW
    npts = 8192 * 256
    gout = torch.randn([npts, 12], device=dev, generator=gen)
    from nerf4k_amd import _native as N_
    bpp = 12 * 4 + 12 + 8 * 12 * 4 * 2

    def scatter(pts, mn, mx):
        # the product path (lib/grid.grid_sample_3d_backward: channel-last scratch + sweep, workspace kept across calls) and the
        # channel-major atomic scatter it replaced (grid.GSB_CHANNEL_LAST = False)
        res = {}
        for name, env in (('', True), ('_channel_major_atomics', False)):
            G.GSB_CHANNEL_LAST = env
            g.zero_()
            res[name] = timed(lambda: G.grid_sample_3d_backward(gout, 12, 417, 353, 256, pts, mn, mx, g))
        G.GSB_CHANNEL_LAST = True
        G._GSB_WS.clear()                                                   # 1.8 GB of workspace: not needed by the rest of the bench
        return res

    def entry(ms):
        return {'ms': round(ms, 3), 'B_per_point': bpp, 'GBs': round(npts * bpp / (ms * 1e-3) / 1e9, 1),
                'frac_hbm': round(npts * bpp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
    pts = torch.rand([npts, 3], device=dev, generator=gen) * 2 - 1
    one = torch.tensor([1., 1., 1.], device=dev)
    for k, ms in scatter(pts, -one, one).items():
        out['grid_sample_bwd_2M_random_points' + k] = dict(entry(ms), note='incoherent points: NOT a workload of the hot path (a training batch scatters rays x consecutive samples, next entry); '
                                                                          '2.1 M random points touch ~13.6 M distinct voxels -- the traffic alone keeps this far from peak, kept as the worst case')
    if frame_rays is not None and model is not None:
        # the coherent case: 8192 rays of the frame x 256 NDC samples each (what a training batch scatters)
        ro, rd = frame_rays[0][:8192], frame_rays[1][:8192]
        t = torch.linspace(0, 1, 256, device=dev)
        pts = (ro[:, None, :] + rd[:, None, :] * t[None, :, None]).reshape(-1, 3).contiguous()
        for k, ms in scatter(pts, model.xyz_min.float().contiguous(), model.xyz_max.float().contiguous()).items():
            out['grid_sample_bwd_8192_rays_x_256' + k] = entry(ms)
    return out


def joint_train_step(ck, frame_rays, H, W, dev, iters=8, world=1, rank=0):
    """BASELINE configs[4] on ONE GPU: iterations of the joint loop (run_sr.py:801-1061 -> 4k-nerf_amd/joint_train.JointTrainer.step) at
    the sizes of configs/llff/fern_lg_joint_l1.py -- a 64x64 ray patch of the 1008x756 view marched through the full 417x353x256
    scene under autograd, SFTNet(5 blocks) x4 to 256x256, L1 + L1 + entropy + distortion + per-point rgb, backward, dense
    total-variation add-grad on both grids, MaskedAdam on the marcher and on the decoder.  Synthetic targets."""
    from nerf4k_amd import joint_train
    from nerf4k_amd.lib import sr_esrnet, utils
    import contextlib
    model = utils.model_from_checkpoint_dict(ck).to(dev).train()
    torch.manual_seed(778)
    net = sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=5, num_grow_ch=32, num_cond=1).to(dev).train()
    cfg = joint_train.JointCfg.fern_lg_joint_l1()
    rk = dict(ck['render_kwargs'], render_depth=True, rand_bkgd=True)
    with contextlib.redirect_stdout(sys.stderr):              # create_optimizer_or_freeze_model prints like upstream; stdout carries the JSON line only
        tr = joint_train.JointTrainer(model, net, cfg, rk, n_train_images=17)
    pr = pc = cfg.N_rand // cfg.N_patch
    gen = torch.Generator(device=dev).manual_seed(5)
    ro, rd, vd = (x.reshape(H, W, 3) for x in frame_rays)

    def batch(i):
        i = i * world + rank                 # data parallel: every rank its own patch (run_sr.py:829-835 is the per-rank unit)
        r0, c0 = (37 * i) % (H - pr), (101 * i) % (W - pc)
        rays = [x[r0:r0 + pr, c0:c0 + pc].reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
        return rays + [torch.rand([pr * pc, 3], device=dev, generator=gen), torch.rand([16 * pr * pc, 3], device=dev, generator=gen), pr, pc]
    first = float(tr.step(*batch(0), global_step=1)['total'])            # warm-up: packing, optimizer state (the second step builds the optimizer's
    for i in range(2):                                                   # fast-path plan), allocator
        tr.step(*batch(1 + i), global_step=2 + i)
    torch.cuda.synchronize()
    # the iteration is paced by the host as much as by the GPU: blocks of iterations, the MEDIAN block is the figure, all of them are reported
    n_samples, blocks, it = 0, [], 3
    per_block = max(1, iters // 2)
    for b in range(5):
        t = time.perf_counter()
        for i in range(per_block):
            tr.step(*batch(it), global_step=it + 1)
            it += 1
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t) / per_block)
    dt = float(np.median(blocks))
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ex = tr.last_exchange or {}
        return {'ms_per_iteration': round(dt * 1e3, 2), 'iterations_per_s': round(1.0 / dt, 2), 'n_gpus': world,
                'rays_per_iteration': pr * pc * world, 'patches_per_iteration': world, 'first_loss': round(first, 5),
                'gradient_exchange': {'sparse_bytes_gathered_per_rank': int(ex.get('bytes_gathered', 0)),
                                      'dense_bucket_bytes': int(ex.get('bytes_dense', 0)),
                                      'touched_voxels_per_rank': [{'counts': c, 'of': v} for c, v in ex.get('touched', [])]},
                'workload': f'configs[4] fern_lg_joint_l1, patch-parallel over {world} GPUs: one 64x64 patch per rank, decoder + small tensors in ONE '
                            'all-reduce bucket, voxel-grid gradients as (index, value) lists in ONE all-gather per grid (RCCL)'}
    # A/B on this box: the decoder as ~20 autograd nodes per RRDB issued call by call (round 5's form; K4_TRAIN_TAPE=0) instead of ONE node on two launch tapes
    from nerf4k_amd.lib import sr_train
    sr_train._TAPE = False
    try:
        for i in range(2):
            tr.step(*batch(it), global_step=it + 1)
            it += 1
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(per_block):
            tr.step(*batch(it), global_step=it + 1)
            it += 1
        torch.cuda.synchronize()
        dt_blocks_graph = (time.perf_counter() - t) / per_block
    finally:
        sr_train._TAPE = True
    # the OTHER 290,000 of the configuration's 300,000 iterations (global_step >= tv_before = 10,000: no total variation, run_sr.py:1005-1011): k0's gradient
    # is the lookups' scatter alone and MaskedAdam skips voxels without one -- the step updates the touched voxels straight from the scatter's scratch image
    late0 = int(cfg.tv_before) + 10
    for i in range(2):
        tr.step(*batch(it), global_step=late0 + it)
        it += 1
    torch.cuda.synchronize()
    blocks_late = []
    for b in range(3):
        t = time.perf_counter()
        for i in range(per_block):
            tr.step(*batch(it), global_step=late0 + it)
            it += 1
        torch.cuda.synchronize()
        blocks_late.append((time.perf_counter() - t) / per_block)
    dt_late = float(np.median(blocks_late))
    # the same iteration's pieces (forward / backward / grid maintenance + optimizers), synchronised
    b = batch(it + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.enable_grad():
        rr, _, ls = tr.forward(*b, global_step=9)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True)
    tr.optimizer_sr.zero_grad(set_to_none=True)
    ls['total'].backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    model.density_total_variation_add_grad(cfg.weight_tv_density / 17, True)
    model.k0_total_variation_add_grad(cfg.weight_tv_k0 / 17, True)
    tr.optimizer.step()
    tr.optimizer_sr.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    n_samples = int(rr['weights'].numel())
    return {'ms_per_iteration': round(dt * 1e3, 2), 'iterations_per_s': round(1.0 / dt, 2),
            'ms_per_iteration_blocks': [round(v * 1e3, 2) for v in blocks], 'statistic': f'median of 5 blocks of {per_block} iterations after 3 warm-up iterations',
            'ms_per_iteration_per_block_graph': round(dt_blocks_graph * 1e3, 2),
            'ms_per_iteration_after_tv_before': round(dt_late * 1e3, 2), 'ms_per_iteration_after_tv_before_blocks': [round(v * 1e3, 2) for v in blocks_late],
            'phases': 'ms_per_iteration = an iteration of the first 10,000 (dense total variation on both grids, every voxel of k0 stepped: the dearer phase, the headline); '
                      'ms_per_iteration_after_tv_before = an iteration of the other 290,000 of fern_lg_joint_l1 (no TV; k0 stepped on its touched voxels from the scatter image)',

            'decoder_pass': 'ONE autograd node on two launch tapes replayed by one native call each (lib/sr_tape.py, k4_tape_*), 3x3 layers of the 64x64 patch on the K-split kernel (K4_CONV_SMALL); every SFT layer backward in two launches (grad_x on the chain, the rest on a third stream: k4_sft_train_bwd_gx / _rest); '
                            'ms_per_iteration_per_block_graph = the same kernels issued call by call from ~20 autograd nodes per RRDB (K4_TRAIN_TAPE=0); round 5: 11.0-11.3 ms; hipGraph form 17.2 ms',
            'rays_per_iteration': pr * pc,
            'shaded_samples': n_samples, 'first_loss': round(first, 5),
            'breakdown_ms': {'forward (march train + SFTNet + losses)': round((t1 - t0) * 1e3, 2), 'backward': round((t2 - t1) * 1e3, 2),
                             'TV add-grad (dense, both grids) + MaskedAdam x 2': round((t3 - t2) * 1e3, 2)},
            'workload': 'configs[4] fern_lg_joint_l1 on 1 GPU: 64x64 patch, 417x353x256 grids (k0 9 ch), rgbnet 15->64->64->3 on k4_rgbnet_*, '
                        'SFTNet 5 blocks on the MFMA conv kernels (fwd / dgrad / wgrad), synthetic targets'}


def own_staged_pipeline(model, rays, rk, chunk=8192, frames=2):
    """The reference's pipeline STRUCTURE on this MI355X -- 8192-ray chunks (run_sr.py:121-124), one launch per op, boolean-mask
    compactions with their host syncs, the op sequence of lib/dmpigo.py:300-427 -- on THIS package's staged gfx950 kernels
    (`k4_staged=True`; rgbnet on k4_rgbnet_fwd).  Context only: it is our own slow path, not the reference's kernels."""
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
            'what': 'reference op sequence, per-op launches of our own staged kernels, 8192-ray chunks, same GPU'}


def scene_sweep(dev, H, W, K, poses):
    """The marcher on OTHER density fields than the one every kernel was tuned on (K1's live mask and skip groups, K2's longest-first
    queue depend on sparsity): SURVEY 8d's second scene ("horns": the same generator, seed 778, 8 test views), a 3x denser field
    (72 blobs) and a 3x sparser one (8 blobs).  Stream-synchronised per-call rate (median of the views), the algorithm's sample counts
    and the same algorithmic-byte roofline fraction as the headline."""
    from nerf4k_amd import scene
    from nerf4k_amd.lib import utils, dvgo
    out = {}
    for name, kw, n_views in (('horns_seed778', dict(seed=778), 8), ('dense_72_blobs', dict(seed=779, n_blobs=72), 4),
                              ('sparse_8_blobs', dict(seed=780, n_blobs=8), 4), ('opaque_wall_seed781', dict(seed=781, opaque=True), 4)):
        ck = scene.make_llff_checkpoint(**kw)
        model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
        rk = ck['render_kwargs']
        views = scene.llff_spiral_poses()[::max(1, 20 // n_views)][:n_views]
        cnt = torch.zeros(8, dtype=torch.int64, device=dev)
        ms = []
        with torch.no_grad():
            for i, p in enumerate(views):
                ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in dvgo.get_rays_of_a_view(H, W, K, torch.from_numpy(p).to(dev), True, False, False, False)]
                model(ro, rd, vd, k4_img_w=W, k4_counters=cnt, **rk)
                model(ro, rd, vd, k4_img_w=W, **rk)
                torch.cuda.synchronize()
                for _ in range(3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); model(ro, rd, vd, k4_img_w=W, **rk); b.record()
                    torch.cuda.synchronize()
                    ms.append(a.elapsed_time(b))
        inb, msk, alp, shd, behind = [c / len(views) for c in cnt.cpu().tolist()[:5]]
        m = float(np.median(ms))
        b_alg = H * W * 56 + inb + msk * 32 + shd * 8 * model.k0_dim * 4
        out[name] = {'views': len(views), 'ms_per_call_median': round(m, 4), 'mrays_isolated': round(H * W / (m * 1e-3) / 1e6, 1),
                     'samples_per_frame': {'in_bbox': int(inb), 'mask': int(msk), 'alpha': int(alp), 'shaded': int(shd)},
                     'shaded_fraction': round(shd / (H * W * 256), 4),
                     'alpha_behind_stop_fraction': round(behind / max(alp, 1), 4),      # density-stage samples the transmittance scan then drops (T < 1e-3 reached in front of them)
                     'roofline_frac': round(b_alg / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del model, ck
        torch.cuda.empty_cache()
    return out


def reference_pipeline_rocm(ck, frame_rays, dev, frames=2, chunk=8192):
    """Denominator of north_star's ">= 30x the reference single-GPU Mrays/s" (BASELINE.md B1): the REFERENCE's pipeline on
    this MI355X -- its own lib/cuda/render_utils_kernel.cu compiled for gfx950 (oracle/_ref, MEASUREMENT ONLY: never loaded by the
    product) serving every native step, F.grid_sample / nn.Linear-shaped torch ops for the rest, boolean-mask compactions with
    their host syncs, in the 8192-ray chunk loop of run_sr.py:121-124 -- as oracle/marcher.py restates DirectMPIGO.forward."""
    from oracle import build_ref, marcher
    if not os.path.exists(build_ref.so_path('render_utils_cuda_ref')):
        return {'error': 'oracle/_ref/render_utils_cuda_ref.so not built (needs /root/reference in the build container)'}
    ref = build_ref.load('render_utils_cuda_ref')
    sd = {k: v.to(dev) for k, v in ck['model_state_dict'].items()}
    ro, rd, vd = frame_rays
    rk = dict(ck['render_kwargs'], render_depth=True)
    saved = marcher.nat
    marcher.nat = ref
    try:
        with torch.no_grad():
            marcher.forward('DirectMPIGO', ck['model_kwargs'], sd, ro[:2 * chunk], rd[:2 * chunk], vd[:2 * chunk], chunk=chunk, **rk)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(frames):
                out = marcher.forward('DirectMPIGO', ck['model_kwargs'], sd, ro, rd, vd, chunk=chunk, **rk)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / frames
    finally:
        marcher.nat = saved
    return {'value': round(ro.shape[0] / dt / 1e6, 3), 'unit': 'Mrays/s', 'ms_per_frame': round(dt * 1e3, 2), 'rays': int(ro.shape[0]),
            'mean_rgb': round(float(out['rgb_marched'].mean()), 5),
            'what': 'reference pipeline on 1x MI355X: the reference\'s own render_utils_cuda kernels (hipified by its own build call, '
                    'oracle/build_ref.py) + torch-ROCm eager ops, 8192-ray chunks; measurement only'}


def dvgo_config0(dev, with_oracle=True):
    """BASELINE configs[0] at its own size on the GPU: DirectVoxGO 160^3, rgbnet_dim 12, viewbase_pe 4, rgbnet 39->128->128->3,
    stepsize 0.5, 64x64 rays (configs/default.py:107-119) -- HIP event time of the fused call, parity against the CPU oracle, and
    an 800x800 frame of the same scene for a throughput figure of k4_geom3_kernel<DVGO> / k4_shade_kernel<DVGO,128,1>."""
    from nerf4k_amd import scene
    from nerf4k_amd.lib import utils, dvgo
    from oracle import marcher
    ck = scene.make_lego_checkpoint()
    model = utils.model_from_checkpoint_dict(ck).to(dev).eval()
    rk = ck['render_kwargs']
    out = {'workload': 'configs[0]: nerf_synthetic-lego-like DirectVoxGO ' + 'x'.join(str(int(v)) for v in model.world_size.tolist())
                       + ', rgbnet 39->128->128->3 (2-term bf16 splits, width-128 kernel at one workgroup per CU), stepsize 0.5'}
    pose = scene.lego_pose()
    with torch.no_grad():
        for H in (64, 800):
            K = scene.lego_K(H, H)
            ro, rd, vd = [x.reshape(-1, 3).contiguous() for x in
                          dvgo.get_rays_of_a_view(H, H, K, torch.from_numpy(pose[:3, :4].astype(np.float32)).to(dev), False, False, False, False)]
            cnt = torch.zeros(8, dtype=torch.int64, device=dev)
            model(ro, rd, vd, k4_img_w=H, k4_counters=cnt, **rk)
            got = model(ro, rd, vd, k4_img_w=H, **rk)
            torch.cuda.synchronize()
            ev = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); model(ro, rd, vd, k4_img_w=H, **rk); b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            inb, msk, alp, shd = cnt.cpu().tolist()[:4]
            b_alg = H * H * 56 + inb + msk * 32 + shd * 8 * model.k0_dim * 4
            out[f'{H}x{H}'] = {'ms': round(ms, 4), 'mrays_per_s': round(H * H / (ms * 1e-3) / 1e6, 2),
                               'samples': {'in_bbox': inb, 'mask': msk, 'alpha': alp, 'shaded': shd},
                               'algorithmic_GBs': round(b_alg / (ms * 1e-3) / 1e9, 1)}
            if H == 64 and with_oracle:
                want = marcher.forward('DirectVoxGO', ck['model_kwargs'], ck['model_state_dict'], ro.cpu(), rd.cpu(), vd.cpu(), **rk)
                d = got['rgb_marched'].cpu().double() - want['rgb_marched'].double()
                mse = float((d ** 2).mean())
                out['64x64']['psnr_vs_oracle_db'] = round(200.0 if mse == 0 else -10.0 * float(np.log10(mse)), 1)
                out['64x64']['max_abs_vs_oracle'] = float(d.abs().max())
    return out


def cpu_baseline(ck, pose, stride, model=None):
    """The CPU oracle ("port": oracle/marcher.py on torch CPU kernels) on a bounded sample of the same frame."""
    from oracle import marcher
    from nerf4k_amd import scene
    H, W = scene.LLFF_HW
    cores, used = _cpu_threads()
    ro, rd, vd = marcher.get_rays_of_a_view(H, W, scene.LLFF_K, pose, ndc=True)
    sel = (slice(None, None, stride), slice(None, None, stride))
    ro, rd, vd = [x[sel].reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], ro[:8192], rd[:8192], vd[:8192],
                    **ck['render_kwargs'])                                  # warm-up
    t = time.perf_counter()
    want = marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], ro, rd, vd, **ck['render_kwargs'])
    dt = time.perf_counter() - t
    base = {'value': round(len(ro) / dt / 1e6, 5), 'unit': 'Mrays/s', 'cores': cores, 'threads_used': used, 'kind': 'port',
            'sample': f'{len(ro)} rays = every {stride}th row and column of one 1008x756 frame, 8192-ray chunks '
                      f'as run_sr.py:121-124, {dt:.1f}s of CPU work, torch {torch.__version__} CPU kernels'}
    # once at every hardware thread (SURVEY 8d asked for os.cpu_count()): a small sample of the same frame, for the record -- torch's CPU
    # kernels stop scaling long before 256 threads (measured: 0.0004 Mrays/s at 256 against 0.06 at 32), which is why the figure above uses CPU_THREAD_CAP
    if cores > used:
        torch.set_num_threads(cores)
        n8 = min(len(ro), 4096)                 # half a chunk of the reference's loop: at 256 threads torch's CPU kernels run ~100x SLOWER than at 32
        t = time.perf_counter()
        marcher.forward('DirectMPIGO', ck['model_kwargs'], ck['model_state_dict'], ro[:n8], rd[:n8], vd[:n8], **ck['render_kwargs'])
        dt8 = time.perf_counter() - t
        base['all_threads'] = {'value': round(n8 / dt8 / 1e6, 5), 'unit': 'Mrays/s', 'threads_used': cores,
                               'sample': f'the first {n8} rays of the same frame, {dt8:.1f}s'}
        torch.set_num_threads(used)
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
                  'what': 'HIP fused marcher vs the CPU oracle on the frame timed for cpu_baseline'}
    return base, parity


if __name__ == '__main__':
    main()
