"""MaskedAdam -- the reference's grid optimizer (lib/masked_adam.py:18-71) over the gfx950 streaming kernels of
csrc/k4_opt.hip.  Same constructor, param-group keys (`lr`, `betas`, `eps`, `skip_zero_grad`), state keys (`step`,
`exp_avg`, `exp_avg_sq`), `set_pervoxel_lr` and kernel selection order as upstream; there is no CPU path."""
import contextlib
from operator import is_ as _is

import torch

from .. import _native as N


def _launch(name, param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    for t in (param, grad, exp_avg, exp_avg_sq) + (() if perlr is None else (perlr,)):
        if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError(f'{name}: tensors must be contiguous fp32 device tensors')   # adam_upd.cpp CHECK_INPUT
        if t.numel() != param.numel():
            raise ValueError(f'{name}: size mismatch')
    fn = getattr(N.lib(), name)
    st = N.stream()
    if perlr is None:
        rc = fn(N.ptr(param), N.ptr(grad), N.ptr(exp_avg), N.ptr(exp_avg_sq), param.numel(), int(step),
                float(beta1), float(beta2), float(lr), float(eps), st)
    else:
        rc = fn(N.ptr(param), N.ptr(grad), N.ptr(exp_avg), N.ptr(exp_avg_sq), N.ptr(perlr), param.numel(), int(step),
                float(beta1), float(beta2), float(lr), float(eps), st)
    N.check(rc, name)
    # the kernels write through raw pointers: tell autograd / every `_version`-keyed cache (the fused marcher's k0 repack and
    # packed rgbnet, SFTNet's packed convs) that these tensors changed -- a render after an optimizer step must not see stale copies
    for t in (param, exp_avg, exp_avg_sq):
        torch.autograd.graph.increment_version(t)


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    """adam_upd_cuda.adam_upd (lib/cuda/adam_upd.cpp:34-44)."""
    _launch('k4_adam_upd', param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps)


def masked_adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    """adam_upd_cuda.masked_adam_upd (lib/cuda/adam_upd.cpp:46-56)."""
    _launch('k4_masked_adam_upd', param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps)


def adam_upd_with_perlr(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    """adam_upd_cuda.adam_upd_with_perlr (lib/cuda/adam_upd.cpp:58-69)."""
    _launch('k4_adam_upd_with_perlr', param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps)


def _side_stream(device):
    """The stream a large grid's update runs on: verified to run beside the current stream and the package's other side streams (_native.overlapping_stream)."""
    return N.overlapping_stream(device, 'grid optimizer step', low_priority=_SIDE_LOW_PRIORITY)


# The first part of a split grid step on at most this many workgroups (0: no cap).  It runs BESIDE the decoder's latency-bound kernels, which an HBM-saturating pass
# slows by more than it saves (uncapped: 7.26 against 7.21 ms per joint iteration without the split); at 384-512 workgroups -- under two per CU, 2.1 ms for the LLFF
# k0's 9.5 GB, beside the decoder's forward pass -- the iteration is 6.8-7.0 ms (profiles/r06_split_grid_step.md)
_EARLY_WORKGROUPS = 448
_SIDE_LOW_PRIORITY = False      # True: the grids' step stream at the device's lowest priority (A/B)


_MULTI_BELOW = 1 << 20          # tensors smaller than this are updated through k4_adam_upd_multi (same arithmetic per element)


_MULTI_PLANS = {}               # (ids of the parameter tensors) -> (job table with the static pointers filled in, tensors kept alive, [params, m, v])


def adam_upd_multi(items, masked, step, beta1, beta2, lr, eps):
    """adam_upd / masked_adam_upd of many (param, grad, exp_avg, exp_avg_sq) tuples in ceil(len / 64) launches (k4_adam_upd_multi).
    The job table of a parameter set is built (and its tensors validated) once; later steps only refresh the gradient pointers --
    the decoder's 458 tensors cost ~3 ms of host time per step otherwise."""
    if not items:
        return
    key = tuple(id(ts[0]) for ts in items)
    plan = _MULTI_PLANS.get(key)
    if plan is not None and any((a[0].data_ptr(), a[2].data_ptr(), a[3].data_ptr()) != ptrs for a, ptrs in zip(items, plan[3])):
        plan = None                                             # a parameter or its state was re-allocated (scale_volume_grid, load_state_dict)
    if plan is None:
        jobs = (N.AdamJob * len(items))()
        for j, ts in enumerate(items):
            for t in ts:
                if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32 or t.numel() != ts[0].numel():
                    raise ValueError('adam_upd_multi: tensors must be contiguous fp32 device tensors of one size')     # adam_upd.cpp CHECK_INPUT
            jobs[j].param, jobs[j].grad, jobs[j].exp_avg, jobs[j].exp_avg_sq = (t.data_ptr() for t in ts)
            jobs[j].n = ts[0].numel()
        touched = [t for ts in items for t in (ts[0], ts[2], ts[3])]
        plan = (jobs, [tuple(ts[i] for i in (0, 2, 3)) for ts in items], touched, [(ts[0].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr()) for ts in items])
        if len(_MULTI_PLANS) > 16:
            _MULTI_PLANS.clear()
        _MULTI_PLANS[key] = plan
    else:
        jobs = plan[0]
        for j, ts in enumerate(items):
            g = ts[1]
            if not g.is_cuda or not g.is_contiguous() or g.dtype != torch.float32 or g.numel() != jobs[j].n:
                raise ValueError('adam_upd_multi: tensors must be contiguous fp32 device tensors of one size')
            jobs[j].grad = g.data_ptr()
    N.check(N.lib().k4_adam_upd_multi(jobs, len(items), int(bool(masked)), int(step), float(beta1), float(beta2), float(lr), float(eps),
                                      N.stream()), 'k4_adam_upd_multi')
    torch.autograd.graph.increment_version(plan[2])


class MaskedAdam(torch.optim.Optimizer):
    """Adam with (1) per-voxel learning rate and (2) masked update that skips zero-gradient voxels."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        # same argument checks (and messages) as lib/masked_adam.py:21-28
        checks = ((lr >= 0.0, f'Invalid learning rate: {lr}'), (eps >= 0.0, f'Invalid epsilon value: {eps}'),
                  (0.0 <= betas[0] < 1.0, f'Invalid beta parameter at index 0: {betas[0]}'),
                  (0.0 <= betas[1] < 1.0, f'Invalid beta parameter at index 1: {betas[1]}'))
        for ok, msg in checks:
            if not ok:
                raise ValueError(msg)
        self.per_lr = None
        self._fast = {}             # id(param group) -> plan of the group's small tensors (see _fast_step)
        self._side = []             # owners (objects with ``.grid``, ``note_pending_update(event)``, ``params_ready()``): see update_on_side_stream
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def update_on_side_stream(self, param, owner):
        """The update of `param` (a large voxel grid) runs on a second HIP stream, behind the gradient on the current stream; the event of
        its completion is handed to ``owner.note_pending_update`` -- the owner (lib/grid.DenseGrid) makes every later reader of the
        parameter wait for it.  The 339 M-float LLFF k0 takes 1.9 ms of HBM time per step (28 bytes per element): on the current stream
        that is 1.9 ms between the end of one training iteration and the first kernel of the next, whose sample selection (a device-to-host
        read of counts, density grid only) the host waits for.  No reference counterpart (one stream there); same values."""
        assert owner.grid is param
        if not any(o is owner for o in self._side):
            self._side.append(owner)             # keyed by the OWNER: its current ``.grid`` is looked up at step time (a .to() / resampled grid is a new Parameter)

    def zero_grad(self, set_to_none=True):
        # a gradient an update on the side stream may still be reading goes back to the current stream's allocator here: wait for that update first
        for owner in self._side:
            owner.params_ready()
        return super().zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        # exp_avg / exp_avg_sq of a grid whose step runs on the second stream: a checkpoint copy taken on the current stream waits for it
        for owner in self._side:
            owner.params_ready()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        for owner in self._side:
            owner.params_ready()
        return super().load_state_dict(state_dict)

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]['params'][0].shape == count.shape
        self.per_lr = (count.float() / count.max()).contiguous()
        self._fast.clear()

    def _fast_step(self, group, masked, beta1, beta2, lr, eps):
        """The group's SMALL tensors (the decoder: 458 of them) in one pass over a plan built by the previous regular step: per tensor one
        ``.grad`` read, one pointer refresh, one step-count store -- the regular path below spent ~2 ms of host time per iteration on them
        (state lookups, contiguity / dtype checks, list building, job-table key), which paces the joint training iteration
        (profiles/r04_train_wgrad_side_stream_slower.md).  Returns False (nothing done) whenever the plan's assumptions do not hold: a tensor
        or its state re-allocated, a missing or non-contiguous gradient, step counts that differ."""
        f = self._fast.get(id(group))
        if f is None or f['masked'] != bool(masked) or len(group['params']) != f['n_group']:
            return False
        plist, states, jobs = f['params'], f['states'], f['jobs']
        if [p.data_ptr() for p in plist] != f['pptr']:
            return False
        sget = self.state.get
        for p, st in zip(plist, states):                    # EVERY tensor's state dict is the one the plan was built on (a replaced state: regular path)
            if sget(p) is not st:
                return False
        grads = [p.grad for p in plist]
        steps = [st['step'] for st in states]
        step = steps[0]
        if steps.count(step) != len(steps):
            return False
        if not all(map(_is, [st['exp_avg'] for st in states], f['m'])) or not all(map(_is, [st['exp_avg_sq'] for st in states], f['v'])):
            return False                                    # a moment tensor was replaced inside its state dict
        # The SAME gradient tensors as at the previous step?  (The decoder's training tape hands every parameter the same view of its flat gradient buffer on
        # every pass; `last_grads` keeps them alive, so "same object" cannot be a recycled id.)  Then their dtype / contiguity / size were checked at that step
        # and their addresses are in the job table already: 458 x (four attribute checks, three data_ptr calls, a pointer store) less per step -- which counts
        # once the iteration is paced by the host (profiles/r06_joint_phase_events.md, section 16).
        last = f['last_grads']
        if last is None or not all(map(_is, grads, last)):
            f32 = torch.float32
            for g, nel in zip(grads, f['numel']):
                if g is None or g.dtype is not f32 or not g.is_cuda or not g.is_contiguous() or g.numel() != nel:
                    return False
            if [t.data_ptr() for t in f['m']] != f['mptr'] or [t.data_ptr() for t in f['v']] != f['vptr']:
                return False
            for j, g in enumerate(grads):
                jobs[j].grad = g.data_ptr()
            f['last_grads'] = grads
        step += 1
        for st in states:
            st['step'] = step
        N.check(N.lib().k4_adam_upd_multi(jobs, len(plist), int(bool(masked)), int(step), float(beta1), float(beta2), float(lr), float(eps),
                                          N.stream()), 'k4_adam_upd_multi')
        torch.autograd.graph.increment_version(f['touched'])
        return True

    def _fast_plan(self, group, masked, items):
        """Remember the small tensors a regular step just updated (all at one step count) for _fast_step."""
        plist = [ts[0] for ts in items]
        states = [self.state[p] for p in plist]
        jobs = (N.AdamJob * len(items))()
        for j, ts in enumerate(items):
            jobs[j].param, jobs[j].grad, jobs[j].exp_avg, jobs[j].exp_avg_sq = (t.data_ptr() for t in ts)
            jobs[j].n = ts[0].numel()
        self._fast[id(group)] = {'params': plist, 'states': states, 'jobs': jobs, 'masked': bool(masked), 'n_group': len(group['params']),
                                 'pptr': [p.data_ptr() for p in plist], 'mptr': [st['exp_avg'].data_ptr() for st in states],
                                 'vptr': [st['exp_avg_sq'].data_ptr() for st in states],
                                 'm': [st['exp_avg'] for st in states], 'v': [st['exp_avg_sq'] for st in states], 'last_grads': None,
                                 'numel': [p.numel() for p in plist], 'ids': {id(p) for p in plist},
                                 'touched': [t for ts in items for t in (ts[0], ts[2], ts[3])]}

    @torch.no_grad()
    def early_step(self, owner, seed, seed_event):
        """First part of this iteration's step of ``owner.grid`` (a multi-channel grid whose step runs on the side stream), BEFORE the backward pass: every voxel the
        lookups' backward cannot touch -- ``owner._k4_split['flags']`` marks those it can -- is stepped with gradient `seed` (the dense TV term written ahead, complete at
        `seed_event`): for such a voxel that is the iteration's whole gradient.  The second part (the flagged voxels, gradient = seed + the scatter's sums) runs in
        ``step`` through ``_sparse_step``.  Adam is elementwise: every element is stepped exactly once with its complete gradient, the same values as the one-pass step.
        What it buys: the dense pass over the grid (1.8 ms for the LLFF k0) runs beside the rest of the iteration instead of between its backward pass and the next
        iteration's lookup.  -> False (nothing done, take the one-pass path) when the split form does not apply."""
        from . import grid as G
        param, sp = owner.grid, owner._k4_split
        group = next((g for g in self.param_groups if any(p is param for p in g['params'])), None)
        if sp is None or sp.get('early', False) or group is None or not group.get('skip_zero_grad') or not any(o is owner for o in self._side):
            return False
        if not param.is_cuda or param.dim() != 5 or param.shape[1] <= 1 or param.grad is not None or not param.is_contiguous() or param.dtype != torch.float32:
            return False
        if self.per_lr is not None and self.per_lr.shape == param.shape:
            return False
        if seed.shape != param.shape or not seed.is_contiguous() or seed.dtype != torch.float32:
            return False
        _, C_, X, Y, Z = param.shape
        if G._gsb_workspace(param.device, C_, X, Y, Z) is None:           # the second part reads the scatter's scratch image
            return False
        (beta1, beta2), lr, eps = group['betas'], group['lr'], group['eps']
        state = self.state[param]
        if not state:
            state.update(step=0, exp_avg=torch.zeros_like(param, memory_format=torch.preserve_format),
                         exp_avg_sq=torch.zeros_like(param, memory_format=torch.preserve_format))
        step = int(state['step']) + 1
        for t in (state['exp_avg'], state['exp_avg_sq']):
            if not t.is_contiguous() or t.dtype != torch.float32:
                return False
        cur, side = torch.cuda.current_stream(param.device), _side_stream(param.device)
        side.wait_stream(cur)                                  # the flags are complete, every read of the old values (lookups, the TV term's stencil) is issued ...
        side.wait_event(seed_event)                            # ... and the TV term itself is done
        seed.record_stream(side)
        sp['flags'].record_stream(side)
        with torch.cuda.stream(side):
            rc = N.lib().k4_masked_adam_upd_unflagged(N.ptr(param), N.ptr(seed), N.ptr(state['exp_avg']), N.ptr(state['exp_avg_sq']), C_, X * Y * Z, N.ptr(sp['flags']),
                                                      step, float(beta1), float(beta2), float(lr), float(eps), int(_EARLY_WORKGROUPS), N.stream())
            if rc == N.K4_ERR_UNSUPPORTED:
                return False
            N.check(rc, 'k4_masked_adam_upd_unflagged')
            ev = torch.cuda.Event()
            ev.record(side)
        owner.note_pending_update(ev)
        owner._k4_sparse_pending = True                        # the second part runs even if no backward pass reaches the grid (its flagged voxels: seed alone)
        sp.update(early=True, seed=seed, step=step, hyper=(float(beta1), float(beta2), float(lr), float(eps)))
        for t in (param, state['exp_avg'], state['exp_avg_sq']):
            torch.autograd.graph.increment_version(t)
        return True

    def _split_late_step(self, param, owner):
        """Second part of a split step (early_step): the flagged voxels, gradient = seed + the sums the scatter left in the scratch image."""
        from . import grid as G
        sp = owner._k4_split
        owner._k4_sparse_pending = False
        owner._k4_split = None
        _, C_, X, Y, Z = param.shape
        hit = G._GSB_WS.get(param.device)
        if hit is None or hit[0] != (C_, X, Y, Z):
            raise N.K4Error('MaskedAdam: the scratch image of the pending grid gradient is gone')
        state = self.state[param]
        if int(state['step']) + 1 != sp['step'] or param.grad is not None:
            raise N.K4Error('MaskedAdam: the second part of a split step does not follow its first part')
        state['step'] = sp['step']
        beta1, beta2, lr, eps = sp['hyper']
        cur, side = torch.cuda.current_stream(param.device), _side_stream(param.device)
        side.wait_stream(cur)                                  # the scatter is done
        hit[1].record_stream(side)
        with torch.cuda.stream(side):
            N.check(N.lib().k4_masked_adam_upd_sparse_cl_seeded(N.ptr(param), N.ptr(state['exp_avg']), N.ptr(state['exp_avg_sq']), N.ptr(hit[1]), N.ptr(sp['seed']),
                                                                N.ptr(sp['flags']), C_, X, Y, Z, int(sp['step']), beta1, beta2, lr, eps, N.stream()),
                    'k4_masked_adam_upd_sparse_cl_seeded')
            ev = torch.cuda.Event()
            ev.record(side)
        hit[2] = ev
        owner.note_pending_update(ev)
        for t in (param, state['exp_avg'], state['exp_avg_sq']):
            torch.autograd.graph.increment_version(t)

    def _sparse_step(self, param, owner, masked, beta1, beta2, lr, eps):
        """`param`'s gradient of this iteration is the sums its lookups' backward left in the channel-last scratch image (DenseGrid._k4_sparse_grad, set by the
        trainer for iterations without any other contribution to it): the masked update of exactly those voxels from there (k4_masked_adam_upd_sparse_cl), on
        the grid's side stream like the dense step.  Anything the in-place form does not cover (a `.grad` that exists after all, no zero-gradient skipping,
        a per-voxel learning rate on this tensor) sweeps the sums into the dense gradient instead and leaves the tensor to the regular path."""
        from . import grid as G
        if owner._k4_split is not None and owner._k4_split.get('early', False):
            return self._split_late_step(param, owner)
        if param.grad is not None or not masked or (self.per_lr is not None and param.shape == self.per_lr.shape):
            G.sweep_pending_grad(owner)
            return
        owner._k4_sparse_pending = False
        _, C_, X, Y, Z = param.shape
        hit = G._GSB_WS.get(param.device)
        if hit is None or hit[0] != (C_, X, Y, Z):
            raise N.K4Error('MaskedAdam: the scratch image of the pending grid gradient is gone')
        state = self.state[param]
        if not state:
            state.update(step=0, exp_avg=torch.zeros_like(param, memory_format=torch.preserve_format),
                         exp_avg_sq=torch.zeros_like(param, memory_format=torch.preserve_format))
        state['step'] += 1
        for t in (param, state['exp_avg'], state['exp_avg_sq']):
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise ValueError('k4_masked_adam_upd_sparse_cl: tensors must be contiguous fp32 device tensors')
        cur, side = torch.cuda.current_stream(param.device), _side_stream(param.device)
        side.wait_stream(cur)                                  # the scatter (and everything that read the old values) is done
        hit[1].record_stream(side)
        with torch.cuda.stream(side):
            N.check(N.lib().k4_masked_adam_upd_sparse_cl(N.ptr(param), N.ptr(state['exp_avg']), N.ptr(state['exp_avg_sq']), N.ptr(hit[1]), C_, X, Y, Z,
                                                         int(state['step']), float(beta1), float(beta2), float(lr), float(eps), N.stream()),
                    'k4_masked_adam_upd_sparse_cl')
            ev = torch.cuda.Event()
            ev.record(side)
        hit[2] = ev                                            # the next scatter into the image waits for the sweep half of this kernel
        owner.note_pending_update(ev)
        for t in (param, state['exp_avg'], state['exp_avg_sq']):
            torch.autograd.graph.increment_version(t)

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            (beta1, beta2), lr, eps = group['betas'], group['lr'], group['eps']
            masked = group['skip_zero_grad']                   # KeyError without it, as upstream (masked_adam.py:45)
            for owner in self._side:
                if owner._k4_sparse_pending and any(p is owner.grid for p in group['params']):
                    self._sparse_step(owner.grid, owner, masked, beta1, beta2, lr, eps)
            small = {}
            fast_ids = self._fast[id(group)]['ids'] if self._fast_step(group, masked, beta1, beta2, lr, eps) else ()
            for param in (p for p in group['params'] if p.grad is not None):
                if id(param) in fast_ids:
                    continue
                state = self.state[param]
                if not state:                                   # lazy state, zeros in the parameter's memory format
                    state.update(step=0, exp_avg=torch.zeros_like(param, memory_format=torch.preserve_format),
                                 exp_avg_sq=torch.zeros_like(param, memory_format=torch.preserve_format))
                state['step'] += 1
                grad = param.grad.contiguous()
                moments = (state['exp_avg'], state['exp_avg_sq'])
                hyper = (state['step'], beta1, beta2, lr, eps)
                owner = next((o for o in self._side if o.grid is param), None) if param.is_cuda and param.numel() >= _MULTI_BELOW else None
                if owner is not None:
                    cur, side = torch.cuda.current_stream(param.device), _side_stream(param.device)
                    side.wait_stream(cur)                       # the gradient (and everything that read the old values) is done
                    # (no grad.record_stream: the owner's readers -- and JointTrainer before zero_grad -- make the current stream wait for this
                    # update, so the gradient's block returns to the allocator only after it; a recorded 1.4 GB block would sit in limbo)
                    if grad is not param.grad:
                        grad.record_stream(side)                # a contiguous COPY made above: a temporary the current stream's allocator would recycle
                    ctx = torch.cuda.stream(side)
                else:
                    ctx = contextlib.nullcontext()
                with ctx:
                    # kernel selection order of lib/masked_adam.py:58-71: per-voxel lr first, then the masked update
                    if self.per_lr is not None and param.shape == self.per_lr.shape:
                        adam_upd_with_perlr(param, grad, *moments, self.per_lr, *hyper)
                    elif param.numel() < _MULTI_BELOW:
                        small.setdefault((bool(masked), state['step']), []).append((param, grad) + moments)
                    elif masked:
                        masked_adam_upd(param, grad, *moments, *hyper)
                    else:
                        adam_upd(param, grad, *moments, *hyper)
                    if owner is not None:
                        ev = torch.cuda.Event()
                        ev.record(side)
                        owner.note_pending_update(ev)
            # the group's small tensors (the decoder: 458 of them) share hyper-parameters: a handful of launches instead of one each
            for (msk, step), items in small.items():
                adam_upd_multi(items, msk, step, beta1, beta2, lr, eps)
            if not fast_ids:
                if len(small) == 1 and self.per_lr is None:     # one step count, one mask flag: the next step can take the fast path
                    self._fast_plan(group, masked, next(iter(small.values())))
                else:
                    self._fast.pop(id(group), None)
            small.clear()
