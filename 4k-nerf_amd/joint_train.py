"""Joint training step of the marcher and the VC-Decoder (BASELINE configs[4]; SURVEY.md 3.4, 8e "Training", 8f ranks 1-3).

Restates ONE iteration of the reference's joint loop (/root/reference/run_sr.py:801-1061, configuration
configs/llff/fern_lg_joint_l1.py) on this package's HIP training graph:

    patch of rays -> DirectMPIGO / DirectVoxGO forward (staged HIP ops under autograd, colour MLP on k4_rgbnet_*)
                  -> SFTNet(rgb_feature, depth)   (lib/sr_train.py: every convolution forward / dgrad / wgrad on MFMA kernels)
                  -> L1(LR) + L1(HR) + background entropy + distortion (k4_distortion_loss) + per-point rgb
                  -> backward -> [data parallel: gradient exchange] -> total-variation add-grad -> MaskedAdam x 2 -> lr decay

The perceptual / GAN terms (weight_pcp, weight_gan; VGG and discriminator networks) are outside SURVEY.md 8 and raise.

Data parallelism (one 64x64 patch per rank, run_sr.py:829-835): the decoder's 15.8 MB of gradients, the rgbnet and every other
small tensor travel in ONE flat all-reduce (lib/sr_train.allreduce_gradients); the voxel grids do not -- `k0.grid` is 1.36 GB
dense while a patch touches well under 1 % of it -- they use ``sparse_grad_allreduce``: every rank compacts the voxels its
backward touched to (int32 index, values) lists, ONE all-gather of the padded lists moves them, and every rank adds all
lists into its dense gradient in RANK order, so the replicas hold bit-identical gradients (and MaskedAdam's "skip voxels
with zero gradient" sees the union of the touched voxels).  Pure torch.distributed plumbing: RCCL on the GPUs, gloo in the
CPU tests.
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _native as N
from .lib import grid as G, sr_train, train_ops, utils
from .lib.masked_adam import MaskedAdam

_ADAM_SIDE = True   # False: the k0 grid's optimizer step on the current stream (A/B, tests)
_TV_SEED = True     # False: dense total variation added after the backward pass, as run_sr.py orders it (A/B, tests)
_FUSED_LOSSES = True     # False: the elementwise loss terms as tensor-library ops (A/B, tests; CPU tensors always)
_SPLIT_GRID_STEP = True      # False: k0's optimizer step of an iteration with a dense TV term in one pass after the backward pass (A/B, tests)
_SPARSE_GRID_GRAD = True     # False: k0's gradient as a dense tensor in the iterations without TV too (A/B, tests)

SPARSE_MIN_NUMEL = 1 << 20          # tensors at least this large are exchanged as (index, value) lists


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def touched_voxels(g, cap=None):
    """Sorted int32 indices of the columns of g [C, V] with a non-zero entry.  On the GPU: one pass of k4_touched_voxels over the gradient
    (no [C, V] boolean temporary -- 340 MB next to the 1.36 GB k0 gradient -- and one host synchronisation, for the count)."""
    C, V = g.shape
    if not g.is_cuda:
        return (g != 0).any(0).nonzero().flatten().to(torch.int32)
    assert g.is_contiguous() and g.dtype == torch.float32
    cap = max(1024, V // 16) if cap is None else cap
    counter = torch.empty([1], dtype=torch.int64, device=g.device)
    while True:
        idx = torch.empty([cap], dtype=torch.int32, device=g.device)
        N.check(N.lib().k4_touched_voxels(N.f32(g), C, V, N.ptr(idx), cap, N.ptr(counter), N.stream()), 'k4_touched_voxels')
        n = int(counter)
        if n <= cap:
            return idx[:n].sort().values
        cap = n


def sparse_grad_allreduce(params, group=None, average=True):
    """Sum (average) the gradients of voxel-grid parameters [1, C, X, Y, Z] over the ranks by exchanging only touched voxels.
    A voxel is touched when any of its C channels has a non-zero gradient.  Returns a dict of exchange statistics."""
    world = _world(group)
    stats = {'world': world, 'bytes_gathered': 0, 'touched': []}
    params = [p for p in params if p.requires_grad]
    if world == 1 and not (N.FORCE_COLLECTIVES and dist.is_initialized()):
        return stats
    for p in params:
        C = p.shape[1] if p.dim() == 5 else 1
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        g = p.grad.view(C, -1)
        V = g.shape[1]
        assert V < 2 ** 31
        idx = touched_voxels(g)
        n = torch.tensor([idx.numel()], dtype=torch.int64, device=g.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n, group=group)
        counts = [int(c) for c in counts]
        cap = max(max(counts), 1)
        # one message per rank: [cap] int32 indices followed by the bits of the [C][cap] fp32 values (integer buffers: plain byte moves)
        send = torch.zeros([(C + 1) * cap], dtype=torch.int32, device=g.device)
        send[:idx.numel()] = idx
        send[cap:].view(torch.float32).view(C, cap)[:, :idx.numel()] = g[:, idx.long()]
        recv = torch.empty([world * (C + 1) * cap], dtype=torch.int32, device=g.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        g[:, idx.long()] = 0                                     # own contribution comes back with the others, in rank order
        scale = 1.0 / world if average else 1.0
        for r in range(world):
            msg = recv[r * (C + 1) * cap:(r + 1) * (C + 1) * cap]
            ri = msg[:counts[r]].long()
            g.index_add_(1, ri, msg[cap:].view(torch.float32).view(C, cap)[:, :counts[r]] * scale)
        stats['bytes_gathered'] += recv.numel() * 4
        stats['touched'].append((counts, V))
    return stats


def exchange_gradients(model, net_sr, group=None):
    """Data-parallel gradient exchange of the joint step: big grids sparse, everything else in one dense bucket."""
    if _world(group) == 1 and not (N.FORCE_COLLECTIVES and dist.is_initialized()):
        return {'world': 1}
    everything = [p for p in list(model.parameters()) + list(net_sr.parameters()) if p.requires_grad]
    big = [p for p in everything if p.numel() >= SPARSE_MIN_NUMEL and p.dim() == 5]
    big_ids = {id(p) for p in big}
    small = [p for p in everything if id(p) not in big_ids]
    stats = sparse_grad_allreduce(big, group=group)
    stats['bytes_dense'] = sr_train.allreduce_gradients(small, group=group)
    return stats


class JointCfg(dict):
    """Attribute-style view of the `fine_train` section (mmcv ConfigDict upstream)."""
    __getattr__ = dict.__getitem__

    @classmethod
    def fern_lg_joint_l1(cls, **over):
        """configs/llff/fern_lg_joint_l1.py on top of llff_default_lg.py and default.py (the values the joint loop reads)."""
        cfg = dict(N_iters=300000, N_rand=4096, N_patch=64, ray_sampler='patch_mimg',
                   lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_srnet=2e-4, lrate_decay=300,
                   skip_zero_grad_fields=['density', 'k0'],
                   weight_main=1.0, weight_entropy_last=0.001, weight_nearclip=0, weight_distortion=0.01, weight_rgbper=0.01,
                   weight_pcp=0, weight_gan=0, weight_style=0,
                   tv_every=1, tv_after=0, tv_before=10000, tv_dense_before=10000, weight_tv_density=1e-5, weight_tv_k0=1e-6)
        cfg.update(over)
        return cls(cfg)


class JointTrainer:
    """Optimizers + one-iteration method of the joint loop.  ``render_kwargs`` as run_sr.py:690-702 builds them
    (``render_depth=True``; ``rand_bkgd`` for LLFF); ``n_train_images`` = len(rays_o_tr), the TV weights' divisor (:1008-1011)."""

    def __init__(self, model, net_sr, cfg_train, render_kwargs, n_train_images, sr_ratio=4, num_cond=1, dim_rend=3, group=None, use_graph=None):
        if cfg_train.weight_pcp > 0 or cfg_train.weight_gan > 0:
            raise NotImplementedError('perceptual / GAN losses (run_sr.py:934-957) are outside the hot-path scope (SURVEY.md 8)')
        if num_cond != 1 or dim_rend != 3:
            raise NotImplementedError('joint step: num_cond=1 (depth condition), dim_rend=3 as configs/llff/fern_lg_joint_l1.py')
        self.model, self.net_sr, self.cfg, self.group = model, net_sr, cfg_train, group
        self.render_kwargs = dict(render_kwargs)
        self.n_train_images, self.sr_ratio = n_train_images, sr_ratio
        self.optimizer = utils.create_optimizer_or_freeze_model(model, cfg_train, global_step=0)                  # run_sr.py:640
        self._side_stream_updates()
        self.optimizer_sr = MaskedAdam([{'params': net_sr.parameters(), 'lr': cfg_train.lrate_srnet, 'kname': 'srnet',
                                         'skip_zero_grad': False}])                                               # run_sr.py:665-667
        self.last_exchange = None
        self._after_march = None
        # K4_TRAIN_GRAPH=1 / use_graph: the decoder's forward + backward of the full-size patch replayed as hipGraphs (lib/sr_train.GraphedDecoder)
        self.use_graph = (os.environ.get('K4_TRAIN_GRAPH', '0') == '1') if use_graph is None else bool(use_graph)
        self._graphed = None

    def rebuild_optimizer(self, global_step=0):
        """Re-create the marcher's optimizer after ``model.scale_volume_grid`` replaced the grid parameters (run_sr.py:812-818 does the
        same after every progressive-growing step): the old MaskedAdam would keep stepping the dead tensors."""
        self.optimizer = utils.create_optimizer_or_freeze_model(self.model, self.cfg, global_step=global_step)
        self._side_stream_updates()
        return self.optimizer

    def _dense_tv_ahead(self, global_step):
        """Write the dense TV term ahead of the backward pass?  Only while TV is dense, and only in a single-process job: under data parallelism (a group of
        its own OR the default process group -- ``self.group`` is None for both that and no job at all) the exchange between backward and TV sends the
        touched voxels, which a dense seed would make all of them (3.6 GB per rank instead of a few MB)."""
        return _TV_SEED and _world(self.group) == 1 and global_step < self.cfg.tv_dense_before

    def _sparse_grid_owners(self):
        """The grids whose gradient may stay in the scatter's scratch image this iteration: multi-channel DenseGrids stepped by MaskedAdam on the side stream
        (the optimizer looks for pending sums there) in a param group that skips zero gradients, without a per-voxel learning rate."""
        opt = self.optimizer
        if not isinstance(opt, MaskedAdam):
            return []
        out = []
        for owner in opt._side:
            g = owner.grid
            if g.is_cuda and g.dim() == 5 and g.shape[1] > 1 and g.requires_grad and not (opt.per_lr is not None and opt.per_lr.shape == g.shape) \
                    and any(pg.get('skip_zero_grad') and any(p is g for p in pg['params']) for pg in opt.param_groups):
                out.append(owner)
        return out

    def _side_stream_updates(self):
        """The feature grid's optimizer step (1.9 ms of HBM time on the LLFF scene) on a second stream: the next iteration's sample selection
        reads the density grid only and no longer queues behind it; DenseGrid makes every reader of k0 wait (lib/grid.DenseGrid.params_ready)."""
        k0 = getattr(self.model, 'k0', None)
        if _ADAM_SIDE and isinstance(self.optimizer, MaskedAdam) and hasattr(k0, 'note_pending_update') and k0.grid.is_cuda:
            self.optimizer.update_on_side_stream(k0.grid, k0)

    def losses(self, rr, rgb_sr, target, target_4x, pr, pc, n_rays):
        """run_sr.py:877-995: the scalar terms of one iteration (dict of tensors; 'total' is what is back-propagated)."""
        cfg, s = self.cfg, self.sr_ratio
        if _FUSED_LOSSES and rgb_sr.is_cuda and rgb_sr.dtype == torch.float32 and rr['rgb_feature'].dtype == torch.float32 and cfg.weight_nearclip <= 0 \
                and tuple(rgb_sr.shape) == (1, 3, s * pr, s * pc):
            # the elementwise terms as ONE autograd node (two launches forward, one backward: lib/train_ops.JointSmallLosses) instead of ~40 tensor-library launches
            ent, per = cfg.weight_entropy_last > 0, cfg.weight_rgbper > 0
            small, terms = train_ops.joint_small_losses(rr['rgb_feature'], rgb_sr, rr['alphainv_last'] if ent else None, rr['raw_rgb'] if per else None,
                                                        target, target_4x.detach().reshape(-1, 3), rr['weights'] if per else None, rr['ray_id'] if per else None,
                                                        cfg.weight_main, cfg.weight_entropy_last if ent else 0.0, cfg.weight_rgbper if per else 0.0)
            out = {'photo': terms[0], 'l1': terms[1], 'psnr_sr': terms[2]}
            if ent:
                out['entropy_last'] = terms[3]
            if cfg.weight_distortion > 0:
                out['distortion'] = cfg.weight_distortion * train_ops.flatten_eff_distloss(rr['weights'], rr['s'], 1 / rr['n_max'], rr['ray_id'],
                                                                                            n_rays=rr['alphainv_last'].shape[0])
            if per:
                out['rgbper'] = terms[4]
            out['total'] = small + out['distortion'] if cfg.weight_distortion > 0 else small
            return out
        out = {'photo': cfg.weight_main * F.l1_loss(rr['rgb_feature'], target)}
        rgb_hr = target_4x.detach().reshape(s * pr, s * pc, 3).movedim(-1, 0).unsqueeze(0)
        out['l1'] = F.l1_loss(rgb_sr, rgb_hr)
        out['psnr_sr'] = -10.0 * torch.log10((rgb_sr.detach().clamp(0, 1) - rgb_hr).pow(2).mean())
        if cfg.weight_entropy_last > 0:
            p = rr['alphainv_last'].clamp(1e-6, 1 - 1e-6)
            out['entropy_last'] = -(p * torch.log(p) + (1 - p) * torch.log(1 - p)).mean() * cfg.weight_entropy_last
        if cfg.weight_nearclip > 0:
            raise NotImplementedError("weight_nearclip needs the 't' / 'raw_density' keys no BASELINE configuration produces")
        if cfg.weight_distortion > 0:
            out['distortion'] = cfg.weight_distortion * train_ops.flatten_eff_distloss(rr['weights'], rr['s'], 1 / rr['n_max'], rr['ray_id'],
                                                                                        n_rays=rr['alphainv_last'].shape[0])
        if cfg.weight_rgbper > 0:
            per = (rr['raw_rgb'] - target[rr['ray_id']]).pow(2).sum(-1)
            out['rgbper'] = cfg.weight_rgbper * (per * rr['weights'].detach()).sum() / n_rays
        out['total'] = sum(v for k, v in out.items() if k != 'psnr_sr')
        return out

    def forward(self, rays_o, rays_d, viewdirs, target, target_4x, pr, pc, global_step):
        rr = self.model(rays_o, rays_d, viewdirs, global_step=global_step, is_train=True, **self.render_kwargs)
        if self._after_march is not None:                        # (step: the first part of a split grid step, as early as the lookups' points are known)
            self._after_march()
        rgb_cache = rr['rgb_feature'].reshape(1, pr, pc, -1).movedim(-1, 1)
        cond = rr['depth'].reshape(1, pr, pc, 1).movedim(-1, 1)                          # num_cond == 1 (run_sr.py:894-897)
        rgb_sr = self._decoder(rgb_cache, cond)                                          # run_sr.py:918
        return rr, rgb_sr, self.losses(rr, rgb_sr, target, target_4x, pr, pc, len(rays_o))

    def _decoder(self, x, cond):
        if self.use_graph and torch.is_grad_enabled() and x.requires_grad:
            if self._graphed is None:
                N_patch = self.cfg.N_rand // self.cfg.N_patch
                if tuple(x.shape[2:]) == (N_patch, N_patch):                 # capture once, for the full-size patch (edge patches stay eager)
                    try:
                        self._graphed = sr_train.GraphedDecoder(self.net_sr, x.shape, cond.shape)
                    except Exception as e:                                    # capture failed (e.g. an op that synchronises): stay eager
                        print(f'JointTrainer: hipGraph capture of the decoder failed ({type(e).__name__}: {e}); using the eager path', flush=True)
                        self.use_graph = False
            if self._graphed is not None and self._graphed.matches(x, cond):
                return self._graphed(x, cond)
        return self.net_sr(x, cond)

    def step(self, rays_o, rays_d, viewdirs, target, target_4x, pr, pc, global_step):
        """One iteration (run_sr.py:869-1014,1052-1061).  Returns the dict of loss tensors (detached)."""
        cfg = self.cfg
        tv_now = cfg.tv_after < global_step < cfg.tv_before and global_step % cfg.tv_every == 0                 # run_sr.py:1005-1011
        # DENSE total variation does not look at the gradient: its term is written ahead of the backward pass (side stream) into the buffer
        # the grid lookups' backward accumulates into -- the same sum with a third of the memory traffic, and none of it at the end of the
        # iteration where the next iteration's sample selection waits (lib/grid.py total_variation_seed_grad).  Not under data parallelism:
        # the gradient exchange between backward and TV sends the TOUCHED voxels, which a dense seed would make all of them.
        seed_tv = tv_now and self._dense_tv_ahead(global_step)
        seeded = []
        # Without TV (after tv_before: 290,000 of fern_lg_joint_l1's 300,000 iterations) the lookups' backward is the only contribution to k0's gradient and
        # MaskedAdam skips voxels without one: the backward stops after its scatter and the optimizer updates the touched voxels from the scratch image
        # (lib/grid.DenseGrid._k4_sparse_grad, MaskedAdam._sparse_step) -- no dense 1.36 GB gradient per iteration.  Single process only: the data-parallel
        # exchange reads the dense gradient.
        sparse = self._sparse_grid_owners() if (_SPARSE_GRID_GRAD and not tv_now and _world(self.group) == 1) else []
        for grid in sparse:
            grid._k4_sparse_grad = True
        if seed_tv:
            for weight, grid, fn in ((cfg.weight_tv_density, getattr(self.model, 'density', None), self.model.density_total_variation_add_grad),
                                     (cfg.weight_tv_k0, getattr(self.model, 'k0', None), self.model.k0_total_variation_add_grad)):
                if weight > 0 and hasattr(grid, 'finish_grad_seed') and grid.grid.requires_grad:
                    fn(weight / self.n_train_images, 'seed')
                    seeded.append(grid)
        # ... and a grid whose gradient is that term + what its lookups' backward scatters is stepped in two exact parts: every voxel the scatter cannot touch right after
        # the forward pass (beside the decoder's passes), the touched ones after the backward pass (MaskedAdam.early_step) -- the dense pass over k0 (1.8 ms) no longer
        # sits between this iteration's backward pass and the next iteration's lookup
        split = [g for g in self._sparse_grid_owners() if g in seeded] if (_SPLIT_GRID_STEP and seed_tv) else []
        for grid in split:
            flags = grid.__dict__.get('_k4_split_flags')
            if flags is None or flags.numel() != grid.grid[0, 0].numel() or flags.device != grid.grid.device:
                flags = grid.__dict__['_k4_split_flags'] = torch.zeros([grid.grid[0, 0].numel()], dtype=torch.uint8, device=grid.grid.device)
            grid._k4_split = {'flags': flags}
        done = False

        def early():                                             # between the marcher's forward pass and the decoder's: the lookups' points are known
            for grid in split:
                seed, ev = grid._k4_seed
                if self.optimizer.early_step(grid, seed, ev):
                    grid._k4_seed = None                         # consumed: the backward pass leaves its sums in the scratch image
                else:
                    grid._k4_split = None                        # one-pass step after all
                    grid.__dict__.pop('_k4_split_flags', None)
        try:
            with torch.enable_grad():
                if split:
                    # (zero_grad makes the current stream wait for a grid's pending update: in front of the first part, not behind it; its place relative to the
                    # forward pass changes nothing -- run_sr.py:961-962 clears the gradients between the loss terms and the backward pass (:1003))
                    self.optimizer.zero_grad(set_to_none=True)
                    self.optimizer_sr.zero_grad(set_to_none=True)
                    self._after_march = early
                try:
                    rr, rgb_sr, ls = self.forward(rays_o, rays_d, viewdirs, target, target_4x, pr, pc, global_step)
                finally:
                    self._after_march = None
                if hasattr(self.model, '_k4_params_ready'):
                    self.model._k4_params_ready()                # (a forward that never read k0: its pending update still precedes what follows)
                if not split:
                    self.optimizer.zero_grad(set_to_none=True)
                    self.optimizer_sr.zero_grad(set_to_none=True)
                ls['total'].backward()
            for grid in seeded:
                grid.finish_grad_seed()
            done = True
        finally:
            for grid in sparse:
                grid._k4_sparse_grad = False
                if not done:
                    G.discard_pending_grad(grid)
            if not done:                # forward / loss / backward raised (e.g. an out-of-memory batch the caller skips): a parked seed must not
                for grid in seeded:     # reach a LATER iteration's gradient -- it holds a TV term of parameters that iteration no longer has
                    grid._k4_seed = None
                for grid in split:      # (a split step's first part stays applied -- those voxels' step of this iteration; flags and scratch image start afresh)
                    G.discard_pending_grad(grid)
                    grid._k4_split = None
                    grid.__dict__.pop('_k4_split_flags', None)
        stepped = False
        try:
            self.last_exchange = exchange_gradients(self.model, self.net_sr, self.group)
            if tv_now:
                if cfg.weight_tv_density > 0 and getattr(self.model, 'density', None) not in seeded:
                    self.model.density_total_variation_add_grad(cfg.weight_tv_density / self.n_train_images, global_step < cfg.tv_dense_before)
                if cfg.weight_tv_k0 > 0 and getattr(self.model, 'k0', None) not in seeded:
                    self.model.k0_total_variation_add_grad(cfg.weight_tv_k0 / self.n_train_images, global_step < cfg.tv_dense_before)
            self.optimizer.step()
            stepped = True
        finally:
            if not stepped:             # sums a scatter-only backward left in the scratch image must not be added to by the next iteration's
                for grid in sparse + split:
                    G.discard_pending_grad(grid)
            for grid in split:
                if grid._k4_split is not None:                   # (the second part did not run)
                    grid._k4_split = None
                    grid.__dict__.pop('_k4_split_flags', None)
        self.optimizer_sr.step()
        factor = 0.1 ** (1 / (cfg.lrate_decay * 1000))                                                           # run_sr.py:1052-1061
        for opt in (self.optimizer, self.optimizer_sr):
            for pg in opt.param_groups:
                pg['lr'] = pg['lr'] * factor
        return {k: v.detach() for k, v in ls.items()}
