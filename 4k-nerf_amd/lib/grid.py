"""Voxel grids with the reference's interface (/root/reference/lib/grid.py).

``DenseGrid`` (lib/grid.py:108-151) and ``MaskGrid`` (lib/grid.py:274-307) keep constructor
arguments, parameter / buffer names (``grid``, ``xyz_min``, ``xyz_max``, ``mask``, ``xyz2ijk_scale``,
``xyz2ijk_shift``) and forward semantics; the lookups run on the gfx950 kernels of lib4k_hip.so.
``TensoRFGrid`` / ``VQGrid`` are not selected by any BASELINE configuration
(configs/default.py:85-86) and are out of the hot-path scope.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as N
from . import render_utils_cuda


def _grid_sample_fwd(grid, pts, xyz_min, xyz_max):
    C_ = grid.shape[1]
    out = torch.empty([pts.shape[0], C_], dtype=torch.float32, device=pts.device)
    N.check(N.lib().k4_grid_sample_3d(N.f32(grid), C_, grid.shape[2], grid.shape[3], grid.shape[4],
                                      N.f32(pts), N.f32(xyz_min), N.f32(xyz_max),
                                      pts.shape[0], N.f32(out), N.stream()), 'grid_sample_3d')
    return out


class GridSample3D(torch.autograd.Function):
    """DenseGrid lookup with a HIP backward: d/d(grid) by fp32 atomic scatter-add (what grid_sampler_3d_backward does for
    lib/grid.py:124 in the reference's training step).  No gradient w.r.t. the sample points (they come from rays)."""

    @staticmethod
    def forward(ctx, grid, pts, xyz_min, xyz_max, owner=None):
        ctx.save_for_backward(pts, xyz_min, xyz_max)
        ctx.grid_shape = tuple(grid.shape)
        ctx.owner = owner                                  # the DenseGrid: its backward may find a pre-seeded gradient buffer there
        return _grid_sample_fwd(grid.detach().contiguous(), pts, xyz_min, xyz_max)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        pts, xyz_min, xyz_max = ctx.saved_tensors
        _, C_, X, Y, Z = ctx.grid_shape
        owner = ctx.owner
        split = owner is not None and owner._k4_split is not None and owner._k4_split.get('early', False)
        if split:
            # MaskedAdam has stepped every voxel this scatter cannot touch already (early_step): the sums stay in the scratch image for the second part
            if not grid_sample_3d_backward_scatter(grad_out.float().contiguous(), C_, X, Y, Z, pts, xyz_min, xyz_max):
                raise N.K4Error('GridSample3D.backward: the scratch image of a grid whose step was split is gone')
            owner._k4_sparse_pending = True
            return None, None, None, None, None
        if owner is not None and owner._k4_sparse_grad and owner._k4_seed is None:
            # the trainer consumes this gradient where the scatter leaves it (MaskedAdam -> k4_masked_adam_upd_sparse_cl): no dense tensor, `.grad` stays None
            if grid_sample_3d_backward_scatter(grad_out.float().contiguous(), C_, X, Y, Z, pts, xyz_min, xyz_max):
                owner._k4_sparse_pending = True
                return None, None, None, None, None
        gg = ctx.owner._take_grad_seed(ctx.grid_shape, grad_out.device) if ctx.owner is not None else None
        if gg is None:
            gg = torch.zeros(ctx.grid_shape, dtype=torch.float32, device=grad_out.device)
        go = grad_out.float().contiguous()
        grid_sample_3d_backward(go, C_, X, Y, Z, pts, xyz_min, xyz_max, gg)
        return gg, None, None, None, None


_GSB_WS = {}          # device -> [(C, X, Y, Z), all-zero workspace of k4_grid_sample_3d_backward_cl, event of its last use]; one grid shape per device


def release_grid_sample_workspace(device=None):
    """Free the cached scratch image(s) of the channel-last grid_sample_3d backward (as large as the gradient: 1.4 GB for the LLFF k0)."""
    for d in list(_GSB_WS) if device is None else [torch.device(device)]:
        _GSB_WS.pop(d, None)


GSB_CHANNEL_LAST = True


def _gsb_workspace(device, C_, X, Y, Z):
    """The device's cleared scratch image for this grid shape ([shape, int32 tensor, event of its last use]) or None (one channel, switched off, out of memory)."""
    nbytes = int(N.lib().k4_grid_sample_3d_backward_workspace_bytes(C_, X, Y, Z)) if GSB_CHANNEL_LAST else -1
    if nbytes <= 0:
        return None
    hit = _GSB_WS.get(device)
    if hit is None or hit[0] != (C_, X, Y, Z):
        _GSB_WS.pop(device, None)
        try:
            hit = _GSB_WS[device] = [(C_, X, Y, Z), torch.zeros([nbytes // 4], dtype=torch.int32, device=device), None]
        except torch.OutOfMemoryError:
            return None
    return hit


def grid_sample_3d_backward_scatter(go, C_, X, Y, Z, pts, xyz_min, xyz_max):
    """The scatter half of ``grid_sample_3d_backward``: the touched voxels' sums stay in the device's scratch image (added to what an earlier scatter left
    there) until ``sweep_pending_grad`` moves them into a dense gradient or MaskedAdam consumes them in place.  -> False when there is no scratch image for
    this grid (the caller takes the dense path)."""
    hit = _gsb_workspace(go.device, C_, X, Y, Z)
    if hit is None:
        return False
    n = pts.shape[0]
    if n == 0:
        return True
    ws, cur = hit[1], torch.cuda.current_stream(go.device)
    if hit[2] is not None:
        cur.wait_event(hit[2])
    ws.record_stream(cur)
    try:
        N.check(N.lib().k4_grid_sample_3d_backward_cl_scatter(N.f32(go), C_, X, Y, Z, N.f32(pts), N.f32(xyz_min), N.f32(xyz_max), n, N.ptr(ws), N.stream()),
                'grid_sample_3d_backward_cl_scatter')
    except Exception:
        _GSB_WS.pop(go.device, None)
        raise
    hit[2] = torch.cuda.Event()
    hit[2].record(cur)
    return True


def sweep_pending_grad(owner):
    """Move the sums a scatter-only backward left in the scratch image into ``owner.grid.grad`` (created when missing): the dense gradient after all."""
    owner._k4_sparse_pending = False
    g = owner.grid
    _, C_, X, Y, Z = g.shape
    hit = _GSB_WS.get(g.device)
    if hit is None or hit[0] != (C_, X, Y, Z):
        raise N.K4Error('sweep_pending_grad: the scratch image of the pending gradient is gone')
    if g.grad is None:
        g.grad = torch.zeros_like(g, memory_format=torch.contiguous_format)
    cur = torch.cuda.current_stream(g.device)
    if hit[2] is not None:
        cur.wait_event(hit[2])
    hit[1].record_stream(cur)
    N.check(N.lib().k4_grid_sample_3d_backward_cl_sweep(C_, X, Y, Z, N.ptr(hit[1]), N.f32(g.grad), N.stream()), 'grid_sample_3d_backward_cl_sweep')
    hit[2] = torch.cuda.Event()
    hit[2].record(cur)


def discard_pending_grad(owner):
    """Forget a scatter-only backward's sums (an iteration the caller gives up on): the scratch image is dropped, the next use allocates a cleared one."""
    if owner._k4_sparse_pending:
        owner._k4_sparse_pending = False
        _GSB_WS.pop(owner.grid.device, None)


def grid_sample_3d_backward(go, C_, X, Y, Z, pts, xyz_min, xyz_max, gg):
    """gg [1|-, C, X, Y, Z] += d(trilinear lookup)/d(grid) for grad_out `go` [n, C] at `pts` [n, 3].  More than one channel: through the
    channel-last scratch image (k4_grid_sample_3d_backward_cl; the workspace, as large as the gradient, is allocated and cleared once
    per device and grid shape and kept -- GSB_CHANNEL_LAST = False (module attribute: bench A/B) or an allocation failure selects the channel-major atomic scatter).
    The workspace must be all-zero on entry and is left all-zero by the sweep; it is ONE buffer per device, so a use on another HIP
    stream waits for the previous use (event), and a failed launch drops it (the next call allocates a cleared one)."""
    L = N.lib()
    n = pts.shape[0]
    hit = _gsb_workspace(go.device, C_, X, Y, Z) if n > 0 else None
    if hit is not None:
        ws, cur = hit[1], torch.cuda.current_stream(go.device)
        if hit[2] is not None:
            cur.wait_event(hit[2])                     # scatter + sweep of the previous call (possibly on another stream) have finished
        ws.record_stream(cur)
        try:
            N.check(L.k4_grid_sample_3d_backward_cl(N.f32(go), C_, X, Y, Z, N.f32(pts), N.f32(xyz_min), N.f32(xyz_max), n, N.f32(gg), N.ptr(ws),
                                                    N.stream()), 'grid_sample_3d_backward_cl')
        except Exception:
            _GSB_WS.pop(go.device, None)               # scatter done but sweep not: the image may hold non-zero sums
            raise
        hit[2] = torch.cuda.Event()
        hit[2].record(cur)
    else:
        N.check(L.k4_grid_sample_3d_backward(N.f32(go), C_, X, Y, Z, N.f32(pts), N.f32(xyz_min), N.f32(xyz_max), n, N.f32(gg), N.stream()),
                'grid_sample_3d_backward')


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    """total_variation_cuda.total_variation_add_grad (lib/cuda/total_variation.cpp:16-20): grad += TV gradient of param,
    in place.  param, grad: [1, C, X, Y, Z] contiguous fp32 device tensors (CHECK_INPUT upstream)."""
    if param.dim() != 5 or grad.shape != param.shape:
        raise ValueError('total_variation_add_grad: param/grad must be [1,C,X,Y,Z] of equal shape')
    for t in (param, grad):
        if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError('total_variation_add_grad: tensors must be contiguous fp32 device tensors')
    N.check(N.lib().k4_total_variation_add_grad(N.ptr(param), N.ptr(grad), float(wx), float(wy), float(wz),
                                                param.size(2), param.size(3), param.size(4), param.numel(),
                                                2 if dense_mode == 'write' else 1 if dense_mode else 0, N.stream()),
            'k4_total_variation_add_grad')


def _tv_stream(device):
    """The stream the dense TV term is written on ahead of the backward pass: verified to run beside the current stream and the package's other side
    streams (_native.overlapping_stream)."""
    return N.overlapping_stream(device, 'dense total variation')


def _vec3(v):
    """float32 CPU copy of a bbox corner given as list / numpy / tensor on any device (the reference's torch.Tensor(v) needs its
    global CUDA default tensor type for device inputs)."""
    if torch.is_tensor(v):
        return v.detach().float().cpu().clone()
    return torch.Tensor(v)


def create_grid(type, **kwargs):
    if type == 'DenseGrid':
        return DenseGrid(**kwargs)
    raise NotImplementedError(f'{type}: only DenseGrid is on the 4K-NeRF hot path (SURVEY.md 2.1 #6)')


class DenseGrid(nn.Module):
    def __init__(self, channels, world_size, xyz_min, xyz_max, **kwargs):
        super().__init__()
        self.channels, self.world_size = channels, world_size
        for name, val in (('xyz_min', xyz_min), ('xyz_max', xyz_max)):
            self.register_buffer(name, _vec3(val))
        self.grid = nn.Parameter(torch.zeros([1, channels, *[int(v) for v in world_size]]))          # [1, C, X, Y, Z], Z fastest

    def forward(self, xyz):
        """Trilinear lookup == F.grid_sample(bilinear, align_corners=True, zero pad) of lib/grid.py:117-128: HIP kernel
        k4_grid_sample_3d; under autograd its backward is k4_grid_sample_3d_backward (gradient w.r.t. the grid)."""
        shape = xyz.shape[:-1]
        pts = xyz.reshape(-1, 3).contiguous()
        self.params_ready()
        if torch.is_grad_enabled() and self.grid.requires_grad:
            sp = self._k4_split
            if sp is not None and pts.shape[0] > 0:            # (a split optimizer step, JointTrainer.step: the voxels this lookup's backward will touch)
                if sp.get('early', False):
                    raise N.K4Error('DenseGrid.forward: a lookup under autograd after the first part of the split optimizer step')
                _, _, X, Y, Z = self.grid.shape
                N.check(N.lib().k4_grid_flag_corners(X, Y, Z, N.f32(pts), N.f32(self.xyz_min), N.f32(self.xyz_max), pts.shape[0], N.ptr(sp['flags']), N.stream()),
                        'k4_grid_flag_corners')
            out = GridSample3D.apply(self.grid, pts.detach(), self.xyz_min, self.xyz_max, self)
        else:
            out = _grid_sample_fwd(self.grid.detach(), pts, self.xyz_min, self.xyz_max)
        out = out.reshape(*shape, self.channels)
        if self.channels == 1:
            out = out.squeeze(-1)
        return out

    def scale_volume_grid(self, new_world_size):
        """Trilinear resample to a new resolution (progressive growing, lib/grid.py:130-135): F.interpolate(trilinear,
        align_corners=True) on the HIP kernel k4_resample_trilinear.  The new tensor replaces the parameter, as upstream."""
        size = tuple(int(v) for v in new_world_size)
        self.params_ready()
        old = self.grid.data
        if self.channels == 0:
            data = torch.zeros([1, 0, *size], device=old.device)
        else:
            if not old.is_cuda:
                # deliberately no F.interpolate fallback: nothing in this package computes on the CPU (move the model to the GPU first)
                raise N.K4Error('scale_volume_grid: the grid must be on the GPU -- call model.to(device) before growing it (no CPU path)')
            src = old.float().contiguous()
            data = torch.empty([1, self.channels, *size], dtype=torch.float32, device=old.device)
            N.check(N.lib().k4_resample_trilinear(N.f32(src), self.channels, src.shape[2], src.shape[3], src.shape[4],
                                                  N.f32(data), size[0], size[1], size[2], N.stream()), 'k4_resample_trilinear')
        # a NEW parameter, as upstream: optimizers holding the old tensor must be re-created (run_sr.py:818; JointTrainer.rebuild_optimizer)
        self.grid = nn.Parameter(data)
        self.world_size = new_world_size

    def total_variation_add_grad(self, wx, wy, wz, dense_mode):
        '''Add gradients by total variation loss in-place (lib/grid.py:137-140).  dense_mode == 'seed' (no reference counterpart): the
        dense term ahead of the backward pass, see total_variation_seed_grad.'''
        if isinstance(dense_mode, str) and dense_mode == 'seed':
            return self.total_variation_seed_grad(wx, wy, wz)
        self.params_ready()
        total_variation_add_grad(self.grid, self.grid.grad, wx, wy, wz, dense_mode)

    # ---- the dense TV term BEFORE the backward pass (no reference counterpart; joint_train.JointTrainer.step) ----
    # The reference adds the term after backward (run_sr.py:1005-1011): zero-fill the gradient (4 B / voxel), scatter, then read grad + param
    # and write grad (12 B) -- on the 339 M-float LLFF k0 that is 1.1 ms at the END of the iteration, where the next iteration's sample
    # selection (a device-to-host read) waits for it.  Dense mode does not look at the gradient, so the term can be WRITTEN into a fresh
    # buffer first (8 B / voxel, on a side stream under the iteration's host-paced phases) and the lookup's backward accumulates into
    # that buffer instead of into zeros: grad = term + scatter, the same sum.
    _k4_seed = None
    # ---- iterations in which the lookups' backward is the ONLY contribution to the gradient and MaskedAdam skips zero-gradient voxels (the joint loop after
    # tv_before: 290,000 of fern_lg_joint_l1's 300,000 iterations): the trainer sets _k4_sparse_grad, the backward then stops after its scatter (sums in the
    # channel-last scratch image, `.grad` stays None, _k4_sparse_pending raised) and MaskedAdam.step updates exactly the touched voxels from there
    # (k4_masked_adam_upd_sparse_cl) -- no 1.36 GB gradient cleared, swept into and read again per iteration.
    _k4_sparse_grad = False
    _k4_sparse_pending = False
    # The step of this grid in two exact parts (MaskedAdam.early_step / _sparse_step; JointTrainer.step sets this for the iterations with a dense TV term written
    # ahead): {'flags': uint8 [X*Y*Z], all-zero before the iteration's first lookup} -- lookups under autograd flag the voxels their backward will touch;
    # early_step adds 'early', 'seed', 'step', 'hyper'
    _k4_split = None
    # ---- an optimizer step of the grid running on a second stream (lib/masked_adam.MaskedAdam.update_on_side_stream) ----
    _k4_pending = None

    def note_pending_update(self, event):
        self._k4_pending = event
        self._k4_pending_seen = set()                     # streams that already wait for it

    def params_ready(self, stream=None, clear=True):
        """Everything queued on `stream` (default: the current one) after this call sees the finished parameter update.  Called by every
        reader of the parameter in this package (lookups, total variation, resampling, the fused marchers' descriptors, state_dict)."""
        ev = self._k4_pending
        if ev is not None:
            # the event stays until the NEXT update replaces it: a reader on another stream later on must wait too (a cleared event made only
            # the first waiting stream safe); a stream waits once
            st = stream if stream is not None else torch.cuda.current_stream(self.grid.device)
            seen = self.__dict__.setdefault('_k4_pending_seen', set())
            if st.cuda_stream not in seen:
                st.wait_event(ev)
                seen.add(st.cuda_stream)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.params_ready()
        return super()._save_to_state_dict(destination, prefix, keep_vars)

    def _apply(self, fn, *args, **kwargs):                 # .to() / .cpu() / .cuda() / .float(): copies of the parameter on the current stream
        self.params_ready()
        return super()._apply(fn, *args, **kwargs)

    def __deepcopy__(self, memo):                          # copy.deepcopy(model) clones the parameter on the current stream
        self.params_ready()
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            if k in ('_k4_seed', '_k4_pending', '_k4_pending_seen', '_k4_sparse_grad', '_k4_sparse_pending', '_k4_split'):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def total_variation_seed_grad(self, wx, wy, wz):
        """Start computing the dense TV term of the CURRENT parameter values into a new buffer (side stream).  The next backward pass
        through this grid accumulates into it; ``finish_grad_seed`` after the backward pass covers a pass that never reached the grid."""
        cur = torch.cuda.current_stream(self.grid.device)
        seed = torch.empty_like(self.grid.data, memory_format=torch.contiguous_format)
        side = _tv_stream(self.grid.device)
        side.wait_stream(cur)                              # the optimizer step that produced these parameter values (and the allocation point)
        self.params_ready(side, clear=False)               # ... also when that step runs on a stream of its own (the current stream still has to wait)
        with torch.cuda.stream(side):
            total_variation_add_grad(self.grid.data, seed, wx, wy, wz, 'write')
            ev = torch.cuda.Event()
            ev.record(side)
        self._k4_seed = (seed, ev)

    def _take_grad_seed(self, shape, device):
        hit, self._k4_seed = self._k4_seed, None
        if hit is None:
            return None
        seed, ev = hit
        torch.cuda.current_stream(device).wait_event(ev)
        if tuple(seed.shape) != tuple(shape) or seed.device != device:        # the grid was replaced in between (scale_volume_grid)
            raise N.K4Error('total_variation_seed_grad: the grid changed shape between the seed and the backward pass')
        return seed

    def finish_grad_seed(self):
        """After the backward pass: a seed no lookup consumed becomes (or is added to) the gradient."""
        if self._k4_seed is not None:
            seed = self._take_grad_seed(self.grid.shape, self.grid.device)
            self.grid.grad = seed if self.grid.grad is None else self.grid.grad.add_(seed)

    def get_dense_grid(self):
        self.params_ready()
        return self.grid

    @torch.no_grad()
    def __isub__(self, val):
        self.grid.data -= val
        torch.autograd.graph.increment_version(self.grid)      # `.data` edits bypass the version counter the repack caches key on
        return self

    def extra_repr(self):
        ws = self.world_size.tolist() if torch.is_tensor(self.world_size) else list(self.world_size)
        return f'channels={self.channels}, world_size={ws}'


def _mask_from_coarse_checkpoint(path, thres):
    """Occupancy of a coarse-stage DVGO checkpoint (lib/grid.py:277-284): alpha of the 3x3x3 max-pooled density >= thres."""
    st = torch.load(path, map_location='cpu', weights_only=False)
    sd, kw = st['model_state_dict'], st['model_kwargs']
    pooled = F.max_pool3d(sd['density.grid'], kernel_size=3, padding=1, stride=1)
    alpha = 1 - torch.exp(-F.softplus(pooled + sd['act_shift']) * kw['voxel_size_ratio'])
    return (alpha >= thres)[0, 0], kw['xyz_min'], kw['xyz_max']


def occupancy_from_alpha(alpha, thres):
    """(F.max_pool3d(alpha[None,None], kernel_size=3, padding=1, stride=1)[0,0] > thres) for an [X,Y,Z] fp32 device tensor, on
    the HIP kernel k4_alpha_maxpool3_gt -> bool tensor [X,Y,Z]."""
    a = alpha.detach().float().contiguous()
    if a.dim() != 3 or not a.is_cuda:
        raise N.K4Error('occupancy_from_alpha: [X,Y,Z] device tensor expected')
    out = torch.empty(a.shape, dtype=torch.bool, device=a.device)
    N.check(N.lib().k4_alpha_maxpool3_gt(N.f32(a), a.shape[0], a.shape[1], a.shape[2], float(thres), N.ptr(out), N.stream()),
            'k4_alpha_maxpool3_gt')
    return out


def grid_nodes(xyz_min, xyz_max, shape):
    """World positions of the nodes of an [X,Y,Z] grid spanning the bbox (the meshgrid of linspaces of lib/dmpigo.py:199-203)."""
    dev = xyz_min.device
    axes = [torch.linspace(float(xyz_min[i]), float(xyz_max[i]), int(shape[i]), device=dev) for i in range(3)]
    return torch.stack(torch.meshgrid(*axes, indexing='ij'), -1)


class MaskGrid(nn.Module):
    """Boolean occupancy grid + the affine map world -> voxel index (buffers `mask`, `xyz2ijk_scale`, `xyz2ijk_shift`)."""

    def __init__(self, path=None, mask_cache_thres=None, mask=None, xyz_min=None, xyz_max=None):
        super().__init__()
        if path is not None:
            self.mask_cache_thres = mask_cache_thres
            mask, xyz_min, xyz_max = _mask_from_coarse_checkpoint(path, mask_cache_thres)
        lo, hi = _vec3(xyz_min), _vec3(xyz_max)
        self.register_buffer('mask', mask.bool())
        scale = (torch.Tensor(list(mask.shape)) - 1) / (hi - lo)
        self.register_buffer('xyz2ijk_scale', scale)
        self.register_buffer('xyz2ijk_shift', -lo * scale)

    @torch.no_grad()
    def forward(self, xyz):
        """Skip known free space: nearest-voxel lookup with C round() (lib/grid.py:295-304) on k4_maskcache_lookup."""
        pts = xyz.reshape(-1, 3).contiguous()
        hit = render_utils_cuda.maskcache_lookup(self.mask, pts, self.xyz2ijk_scale, self.xyz2ijk_shift)
        return hit.reshape(xyz.shape[:-1])

    def extra_repr(self):
        return f'mask.shape={list(self.mask.shape)}'
