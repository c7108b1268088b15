"""Training path of the VC-Decoder: ``SFTNet`` forward + backward with every convolution on the gfx950
matrix cores, and the gradient exchange of patch-parallel data parallelism.

The reference back-propagates its losses through ``SFTNet`` with PyTorch autograd over cuDNN convolutions
(/root/reference/run_sr.py:869-1014; modules lib/sr_esrnet.py:112-182, forward :446-465).  Here the graph is evaluated on NHWC
``[H, W, C]`` tensors; each ``nn.Conv2d`` call (3x3 and 1x1: all of the network's FLOPs) is the autograd Function ``K4Conv2d``:
    forward : k4_conv2d_nhwc_bf16x6           (csrc/k4_sr.hip: exact 3-term bf16 splits, 6 MFMA products, fp32-equivalent)
    dgrad   : the same kernel on W'[ci][co][2-dy][2-dx] = W[co][ci][dy][dx]   (a "same" convolution of dY with the flipped,
              transposed filter; packed once per weight version)
    wgrad   : k4_conv2d_wgrad_dbias_bf16x6    (csrc/k4_sr_bwd.hip: MFMA GEMM over the pixels, split-K + fp32 atomics; the bias
              gradient is summed by the same launch)
A LeakyReLU that follows a convolution runs in its epilogue; an SFTLayer is one Function (``K4SFTLayer``: fused exact-fp32 kernels
forward and backward); a ResidualDenseBlock with its two SFT layers is one Function (``K4RDB``: the dense block and its gradient live
in one [H, W, 192] image each, no concatenations); all weight operands are re-packed by ceil(n/64) launches per iteration
(``_WeightCache.prepack``).  What stays on PyTorch ops and their autograd: the RRDB / trunk residual adds and nearest x2 upsampling.
There is no CPU path.

``allreduce_gradients`` is the data-parallel exchange: each rank back-propagates its own 64x64 patch (run_sr.py:829-835), the
458 gradient tensors (15.8 MB) are flattened into one bucket and summed with ONE all-reduce (RCCL over xGMI on the GPUs:
ring all-reduce of 15.8 MB moves 2*(N-1)/N * 15.8 MB per rank = 27.7 MB at N=8, ~0.2 ms at the per-link rate) and averaged.
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import _native as N
from .sr_esrnet import _Packed, _PackPlan, SFTNet, EPI_LRELU, EPI_RES, CONV_SMALL


class _WeightCache:
    """Packed forward / transposed (dgrad) weights per (tensor, version)."""

    def __init__(self):
        self._c = {}
        self._plan = None          # prepack(): (signature, _PackPlan, modules, [weight versions], [bias versions])
        self.always = False        # True: pack on every call (hipGraph capture: the packing kernels must be part of the graph, see GraphedDecoder)

    @staticmethod
    def _key(kind, weight, bias):
        return (kind, weight.data_ptr(), weight._version, None if bias is None else bias._version, weight.device)

    def _get(self, kind, weight, bias, make):
        hit = self._c.get((kind, weight.data_ptr()))
        slot = (kind, weight.data_ptr())
        if hit is not None and type(hit[0]) is int:       # an operand of the prepack plan: current iff prepack() saw these versions
            q, plan = hit[0], self._plan
            if plan is not None and (self.always or (plan[3][q] == weight._version and (bias is None or plan[4][q] == bias._version))):
                return hit[1]
            # stale plan operand (a step without prepack()): pack for this use under a SEPARATE key -- the plan's slot stays, so the
            # next prepack() brings the operand back onto the persistent buffers
            hit, slot = self._c.get(('stale',) + slot), ('stale',) + slot
        if self.always:
            return make()
        key = self._key(kind, weight, bias)
        if hit is None or hit[0] != key:
            hit = (key, make())
            self._c[slot] = hit
        return hit[1]

    def fwd(self, weight, bias):
        return self._get('f', weight, bias, lambda: _Packed.native(weight, bias))

    def bwd(self, weight):
        # dgrad operand: the filter flipped in both taps axes and transposed to [cin, cout, k, k], packed by the same kernel
        return self._get('b', weight, None, lambda: _Packed.native(weight, None, dgrad=True))

    def prepack(self, convs, dgrad=True):
        """Pack every operand of `convs` (modules with .weight / .bias) that an optimizer step made stale, now, in ceil(n / 64) launches
        into buffers that persist (sr_esrnet._PackPlan) instead of one launch + two allocations per operand on first use.  The plan is
        rebuilt when the module list or a parameter's storage changes; under ``always`` every call re-packs."""
        sig = (dgrad,) + tuple(m.weight.data_ptr() for m in convs)
        plan = self._plan
        if plan is None or plan[0] != sig:
            items = [(m.weight, m.bias if kind == 'f' else None, kind == 'b') for m in convs for kind in (('f', 'b') if dgrad else ('f',))]
            pp = _PackPlan(items)
            if not pp.live:                                # non-fp32 / non-contiguous parameters: the per-use path handles them
                if self._plan is not None:                 # ... and the old plan's integer-tagged slots must not outlive it (_get reads plan[3])
                    for slot in self._plan[5]:
                        self._c.pop(slot, None)
                self._plan = None
                return
            slots = [(kind, m.weight.data_ptr()) for m in convs for kind in (('f', 'b') if dgrad else ('f',))]
            if self._plan is not None:
                for slot in self._plan[5]:
                    self._c.pop(slot, None)
            n = len(slots) // len(convs)
            for q, (slot, pk) in enumerate(zip(slots, pp.packed)):
                self._c[slot] = (q // n, pk)                # index of the module: its versions live in plan[3] / plan[4]
            plan = self._plan = [sig, pp, list(convs), None, None, slots]
        wv = [m.weight._version for m in convs]
        bv = [-1 if m.bias is None else m.bias._version for m in convs]
        if self.always or wv != plan[3] or bv != plan[4]:
            plan[1].run()
            plan[3], plan[4] = wv, bv

    def invalidate(self):
        self._c.clear()
        self._plan = None


def _wgrad(x, x_off, cin, x_stride, gy, gy_off, cout, gy_stride, k, H, W, weight_shape, with_bias):
    """dW (and dbias) of a stride-1 "same" convolution from channel windows of NHWC images: one zero-fill + one launch."""
    L = N.lib()
    nw = cout * cin * k * k
    buf = torch.empty([nw + (cout if with_bias else 0)], dtype=torch.float32, device=x.device)         # zeroed by the entry point
    xp, gp = N.C.c_void_p(x.data_ptr() + 4 * x_off), N.C.c_void_p(gy.data_ptr() + 4 * gy_off)
    if with_bias:
        N.check(L.k4_conv2d_wgrad_dbias_bf16x6(xp, cin, x_stride, gp, cout, gy_stride, k, H, W, N.f32(buf), N.stream()), 'k4_conv2d_wgrad_dbias_bf16x6')
        return buf[:nw].view(weight_shape), buf[nw:]
    N.check(L.k4_conv2d_wgrad_bf16x6(xp, cin, x_stride, gp, cout, gy_stride, k, H, W, N.f32(buf), N.stream()), 'k4_conv2d_wgrad_bf16x6')
    return buf.view(weight_shape), None


def _lrelu_bwd(g, g_off, g_stride, y, y_off, y_stride, n_pix, channels, out, out_off, out_stride):
    N.check(N.lib().k4_lrelu_bwd(N.C.c_void_p(g.data_ptr() + 4 * g_off), g_stride, N.C.c_void_p(y.data_ptr() + 4 * y_off), y_stride, n_pix, channels,
                                 0.2, N.C.c_void_p(out.data_ptr() + 4 * out_off), out_stride, N.stream()), 'k4_lrelu_bwd')


class K4Conv2d(torch.autograd.Function):
    """stride-1 "same" convolution of an NHWC [H, W, Cin] image with an nn.Conv2d weight [Cout, Cin, k, k] (k = 1 | 3); with ``act``
    the LeakyReLU(0.2) that follows runs in the kernel's epilogue (its backward: one k4_lrelu_bwd on the incoming gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, act=False):
        if not x.is_cuda:
            raise N.K4Error('K4Conv2d: the MI355X-native decoder has no CPU path')
        x = x.contiguous()
        H, W, cin = x.shape
        cout, cin_w, k, _ = weight.shape
        assert cin == cin_w and x.dtype == torch.float32
        y = torch.empty([H, W, cout], dtype=torch.float32, device=x.device)
        small = CONV_SMALL if k == 3 else 0                # (the K-split kernel on small images: include/k4nerf.h K4_CONV_SMALL)
        SFTNet._conv(cache.fwd(weight, bias), x, 0, cin, y, 0, cout, cout, H, W, flags=(EPI_LRELU if act else 0) | small)
        if act:
            ctx.save_for_backward(x, weight, y)
        else:
            ctx.save_for_backward(x, weight)
        ctx.cache, ctx.has_bias, ctx.act = cache, bias is not None, act
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors[:2]
        gy = gy.contiguous().float()
        H, W, cin = x.shape
        cout, _, k, _ = weight.shape
        if ctx.act:
            g2 = torch.empty_like(gy)
            _lrelu_bwd(gy, 0, cout, ctx.saved_tensors[2], 0, cout, H * W, cout, g2, 0, cout)
            gy = g2
        gx = gw = gb = None
        L = N.lib()
        if ctx.needs_input_grad[0]:
            gx = torch.empty([H, W, cin], dtype=torch.float32, device=x.device)
            SFTNet._conv(ctx.cache.bwd(weight), gy, 0, cout, gx, 0, cin, cin, H, W, flags=CONV_SMALL if k == 3 else 0)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw, gb = _wgrad(x, 0, cin, cin, gy, 0, cout, cout, k, H, W, weight.shape, want_b)
        elif want_b:
            gb = torch.empty([cout], dtype=torch.float32, device=x.device)
            N.check(L.k4_conv2d_bias_grad(N.f32(gy), cout, cout, H * W, N.f32(gb), N.stream()), 'k4_conv2d_bias_grad')
        return gx, gw, gb, None, None


class _CondFan(torch.autograd.Function):
    """The condition map on its way to the decoder's SFT layers (21 consumers in SFTNet: 15 dense blocks, 6 SFT layers).  Forward: the map
    itself.  The consumers ADD their condition gradients into `acc` inside their backward kernels (k4_sft_train_bwd_ex) and return None;
    this node, which the engine runs after all of them, returns the sum -- instead of 35 elementwise additions per backward pass (20 by the
    autograd engine, 15 inside the dense blocks), each a launch on the host path that paces the joint training iteration.
    `acc` must be zero when the backward pass starts (``forward_train`` allocates it zeroed, one per forward)."""

    @staticmethod
    def forward(ctx, c, acc):
        ctx.acc = acc
        ctx.set_materialize_grads(False)
        return c.view_as(c)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        acc = ctx.acc
        acc._k4_spent = True                              # a second backward pass over the same graph would add to a gradient already handed out
        return (acc if g is None else acc + g), None      # g: consumers outside the fused Functions (none in SFTNet)


class K4SFTLayer(torch.autograd.Function):
    """SFTLayer (lib/sr_esrnet.py:112-123) of an NHWC image: ``x * (scale(cond) + 1) + shift(cond)`` with both 1x1-convolution pairs,
    LeakyReLU and the modulation in ONE launch forward (k4_sft_train_fwd) and two backward (k4_sft_train_bwd: grad_x, grad_cond, the
    eight weight / bias gradients).  As four K4Conv2d Functions + elementwise autograd a layer was ~45 launches per iteration.
    `acc` (None | [H, W, 32]): the layer's condition gradient is ADDED to it and None returned for `cond` (see _CondFan).
    `res` (None | [H, W, C]), `res_scale`: the layer's output goes through ``* res_scale + res`` in the forward kernel's store (the RRDB's
    skip connection, lib/sr_esrnet.py:181: two elementwise launches less forward, one less backward; same roundings)."""

    @staticmethod
    def forward(ctx, x, cond, acc, res, res_scale, *params):
        if not x.is_cuda:
            raise N.K4Error('K4SFTLayer: the MI355X-native decoder has no CPU path')
        ctx.direct = None
        if len(params) == 1 and isinstance(params[0], (list, tuple)):      # direct mode, as K4RDB
            ctx.direct = params = list(params[0])
        w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h = params
        x, cond = x.contiguous().float(), cond.contiguous().float()
        H, W, C = x.shape
        assert cond.shape == (H, W, 32) and w0s.shape[:2] == (32, 32) and w1s.shape[:2] == (C, 32)
        assert acc is None or (acc.shape == cond.shape and acc.is_contiguous() and acc.dtype == torch.float32)
        ws = [t.detach().contiguous() for t in (w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h)]
        y = torch.empty_like(x)
        if res is not None:
            res = res.contiguous().float()
            assert res.shape == x.shape
        N.check(N.lib().k4_sft_train_fwd_ex(N.f32(x), C, N.f32(cond), 32, H * W, C, *[N.f32(t) for t in ws], 0.2, N.f32(y), C,
                                            None if res is None else N.f32(res), C, float(res_scale), N.stream()), 'k4_sft_train_fwd_ex')
        ctx.save_for_backward(x, cond, *ws)
        ctx.acc = acc
        ctx.gy_scale = float(res_scale) if res is not None else 1.0
        ctx.has_res = res is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, cond, w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h = ctx.saved_tensors
        acc = ctx.acc
        _check_acc(acc)
        gy = gy.contiguous().float()
        H, W, C = x.shape
        n = H * W
        L = N.lib()
        gx = torch.empty_like(x)
        gc = acc if acc is not None else torch.empty_like(cond)
        g = [torch.empty_like(t) for t in (w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h)]
        nbytes = int(L.k4_sft_train_bwd_workspace_bytes(n, C))
        ws = torch.empty([nbytes // 4], dtype=torch.float32, device=x.device)
        N.check(L.k4_sft_train_bwd_ex(N.f32(x), C, N.f32(cond), 32, N.f32(gy), C, n, C, N.f32(w0s), N.f32(b0s), N.f32(w1s), N.f32(b1s),
                                      N.f32(w0h), N.f32(b0h), N.f32(w1h), 0.2, N.f32(gx), N.f32(gc), *[N.f32(t) for t in g],
                                      N.f32(ws), nbytes, None, 0, int(acc is not None), 0, ctx.gy_scale, N.stream()), 'k4_sft_train_bwd_ex')
        g_res = gy if ctx.has_res else None                # the skip connection's gradient is the incoming one itself
        if ctx.direct is not None:
            _hand_over_grads(ctx.direct, g)
            return gx, None if acc is not None else gc, None, g_res, None, None
        return (gx, None if acc is not None else gc, None, g_res, None, *g)


def _sft_bwd(x, x_stride, C, cond, gy, gy_off, gy_stride, n_pix, ws):
    """k4_sft_train_bwd on channel windows: (grad_x [n_pix, C], grad_cond [n_pix, 32], the eight weight / bias gradients)."""
    L = N.lib()
    w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h = ws
    gx = torch.empty([n_pix, C], dtype=torch.float32, device=x.device)
    gc = torch.empty([n_pix, 32], dtype=torch.float32, device=x.device)
    g = [torch.empty_like(t) for t in ws]
    nbytes = int(L.k4_sft_train_bwd_workspace_bytes(n_pix, C))
    wk = torch.empty([nbytes // 4], dtype=torch.float32, device=x.device)
    N.check(L.k4_sft_train_bwd(N.f32(x), x_stride, N.f32(cond), 32, N.C.c_void_p(gy.data_ptr() + 4 * gy_off), gy_stride, n_pix, C,
                               N.f32(w0s), N.f32(b0s), N.f32(w1s), N.f32(b1s), N.f32(w0h), N.f32(b0h), N.f32(w1h), 0.2, N.f32(gx), N.f32(gc),
                               *[N.f32(t) for t in g], N.f32(wk), nbytes, N.stream()), 'k4_sft_train_bwd')
    return gx, gc, g


_NATIVE_RDB = True      # False: a dense block's launches issued one by one from Python (A/B)
_WGRAD_STREAM = True    # False: the block's weight gradients on the chain's own stream (A/B)
_SIDE_LOW_PRIORITY = False      # True: the weight gradients' stream at the device's lowest priority (A/B: no effect)
_FUSED_LRELU = True     # False: a dense block's four LeakyReLU backward passes as launches of their own (A/B, tests)
_DIRECT_GRADS = os.environ.get('K4_TRAIN_DIRECT_GRADS', '1') != '0'  # 0: every parameter an autograd input of its Function (torch.autograd.grad, parameter hooks)
_COND_ACC = True        # False: every SFT consumer returns its condition gradient, autograd adds them (A/B)
_TAIL_SPLIT = True      # False: the last weight gradients of the backward pass (conv_first, CondNet) in one queue behind the chain's last launch (A/B)
_AUX_WGRAD = True       # False: all five weight gradients of a dense block on the weight gradients' stream (A/B)
_SFT_SPLIT = True       # False: the SFT layers' whole backward on the chain's stream, as one launch each (A/B, tests)
_TAPE = os.environ.get('K4_TRAIN_TAPE', '1') != '0'     # 0: the decoder as ~20 autograd nodes per RRDB (below) instead of ONE node on two launch tapes (lib/sr_tape.py)


def _side_stream(device):
    """The second HIP stream (per device) a dense block's weight-gradient launches go to (k4_rdb_train_bwd: forked / joined inside the call)."""
    if not _WGRAD_STREAM:
        return None
    # (one per MAIN stream -- callers on different streams do not share one -- and verified to run BESIDE it: _native.overlapping_stream)
    st = N.overlapping_stream(device, 'decoder weight gradients', low_priority=_SIDE_LOW_PRIORITY)
    return st.cuda_stream


def _aux_stream(device):
    """The third stream of the decoder's backward pass (lib/sr_tape.py): the SFT layers' deferred backward (k4_sft_train_bwd_rest) and their reductions.  It is
    the stream the dense total-variation term uses (lib/grid.py: that work is done long before the decoder's backward pass starts, and stream order keeps any
    overlap correct) -- main + weight gradients + grid optimizer step + this one are the four hardware queues a process gets (_native.overlapping_stream)."""
    if not (_WGRAD_STREAM and _SFT_SPLIT):
        return None
    return N.overlapping_stream(device, 'dense total variation', low_priority=_SIDE_LOW_PRIORITY).cuda_stream


def _hand_over_grads(params, grads):
    """Direct mode of the fused Functions (K4_TRAIN_DIRECT_GRADS, see forward_train): the parameter gradients go to ``.grad`` here instead of
    through 26 (8) AccumulateGrad nodes per Function -- what those do for a leaf: assign when there is no gradient yet, add otherwise."""
    for prm, g in zip(params, grads):
        if g is None or not prm.requires_grad:
            continue
        if prm.grad is None:
            prm.grad = g
        else:
            prm.grad.add_(g)


def _check_acc(acc):
    if acc is not None and getattr(acc, '_k4_spent', False):
        raise N.K4Error('second backward pass through a decoder graph whose condition-gradient accumulator was already consumed '
                        '(retain_graph use: sr_train._COND_ACC = False)')


def _rdb_desc(t, c, buf, x4, P, H, W, nf, g):
    d = N.RdbTrain()
    d.H, d.W, d.nf, d.g = H, W, nf, g
    d.t, d.c, d.buf, d.x4 = t.data_ptr(), c.data_ptr(), buf.data_ptr(), x4.data_ptr()
    for i in range(8):
        d.sft0[i] = P[i].data_ptr()
        d.sft1[i] = P[18 + i].data_ptr()
    return d


_RDB_LAYOUTS = {}


def _rdb_bwd_layout(P, n, nf, g, with_gc):
    """Float offsets of a dense block's backward buffers inside TWO allocations (per (n_pix, nf, g): computed once).
    param grads: [dW1 | db1 | ... | dW5 | db5] (the span one launch zeroes for the split-K weight gradients) | sft0's eight | sft1's eight;
    scratch:     gx0 [n, nf] | G [n, bw] | gx4 [n, g] | sft0 workspace | sft1 workspace | g5 [n, nf] | (gc0 | gc1 [n, 32] without an accumulator).
    As ~31 torch.empty calls + one zero-fill per layer this was a third of K4RDB.backward's host time, which paces the joint iteration."""
    key = (n, nf, g, with_gc)
    lay = _RDB_LAYOUTS.get(key)
    if lay is None:
        L = N.lib()
        order = [8, 9, 10, 11, 12, 13, 14, 15, 16, 17] + list(range(8)) + list(range(18, 26))        # indices into P, in buffer order
        sizes = [P[i].numel() for i in order]
        offs = [0]
        for q in sizes:
            offs.append(offs[-1] + q)
        span = offs[10]
        nb0, nb1 = int(L.k4_sft_train_bwd_workspace_bytes(n, nf)), int(L.k4_sft_train_bwd_workspace_bytes(n, g))
        bw = nf + 4 * g
        ssz = [n * nf, n * bw, n * g, nb0 // 4, nb1 // 4, n * nf] + ([n * 32, n * 32] if with_gc else [])
        soff = [0]
        for q in ssz:
            soff.append(soff[-1] + q)
        lay = _RDB_LAYOUTS[key] = (order, sizes, offs, span, soff, nb0, nb1, [tuple(P[i].shape) if P[i].dim() > 1 else None for i in order])
    return lay


class K4RDB(torch.autograd.Function):
    """ResidualDenseBlock_5C with its two SFT layers (lib/sr_esrnet.py:126-158) as ONE autograd node.

    Forward: the dense block lives in one [H, W, nf + 4g] image (xc0 | x1 | x2 | x3 | xc1) that every convolution reads a channel
    prefix of and writes its slice of (as the inference path does) -- no torch.cat, LeakyReLU and ``x5 * 0.2 + x`` in the epilogues.
    Backward: ONE gradient image of the same shape; the dgrad of conv_k ACCUMULATES into the channel prefix it read (residual epilogue
    with the output as its own residual), so the five-way sums of the concatenations need no kernels of their own.
    7 launches forward, 21 backward (+ the weight packers, batched by _WeightCache.prepack); as separate Functions a block was ~70.
    `acc` (None | [H, W, 32]): the block's condition gradient is ADDED to it and None returned for `c` (see _CondFan).
    params: sft0 (w0s b0s w1s b1s w0h b0h w1h b1h), conv1..conv5 (weight, bias), sft1 (8)."""

    @staticmethod
    def forward(ctx, t, c, cache, acc, *P):
        if not t.is_cuda:
            raise N.K4Error('K4RDB: the MI355X-native decoder has no CPU path')
        L = N.lib()
        ctx.direct = None
        if len(P) == 1 and isinstance(P[0], (list, tuple)):       # direct mode: the 26 parameters as ONE opaque argument (no autograd edges to them)
            ctx.direct = P = list(P[0])
        t, c = t.contiguous(), c.contiguous()
        H, W, nf = t.shape
        g = P[8].shape[0]
        bw, n = nf + 4 * g, H * W
        assert c.shape == (H, W, 32) and len(P) == 26 and g == 32 and nf in (32, 64)
        assert acc is None or (acc.shape == c.shape and acc.is_contiguous() and acc.dtype == torch.float32)
        for q in P:
            if not q.is_contiguous():
                P = [q.contiguous() for q in P]
                break
        f32 = dict(dtype=torch.float32, device=t.device)
        buf, x4, out = torch.empty([H, W, bw], **f32), torch.empty([H, W, g], **f32), torch.empty([H, W, nf], **f32)
        ctx.acc = acc
        if _NATIVE_RDB:                                   # the seven launches below, issued by ONE native call (include/k4nerf.h, k4_rdb_train)
            d = _rdb_desc(t, c, buf, x4, P, H, W, nf, g)
            d.out = out.data_ptr()
            packs = [cache.fwd(P[8 + 2 * k], P[9 + 2 * k]) for k in range(5)]
            for k, pk in enumerate(packs):
                assert pk.mode == 'bf16x6' and pk.k == 3 and pk.flags_extra == 0
                d.w_fwd[k], d.b_fwd[k] = pk.w.data_ptr(), pk.b.data_ptr()
            N.check(L.k4_rdb_train_fwd(N.C.byref(d), N.stream()), 'k4_rdb_train_fwd')
            if ctx.direct is not None:
                ctx.save_for_backward(t, c, buf, x4)
                ctx.P = P
            else:
                ctx.save_for_backward(t, c, buf, x4, *P)
            ctx.cache, ctx.desc = cache, d                # the backward fills in its own fields of the same descriptor
            return out
        N.check(L.k4_sft_train_fwd(N.f32(t), nf, N.f32(c), 32, n, nf, *[N.f32(q) for q in P[0:8]], 0.2, N.f32(buf), bw, N.stream()), 'k4_sft_train_fwd')
        for k in (1, 2, 3):
            SFTNet._conv(cache.fwd(P[6 + 2 * k], P[7 + 2 * k]), buf, 0, bw, buf, nf + (k - 1) * g, bw, g, H, W, flags=EPI_LRELU)
        SFTNet._conv(cache.fwd(P[14], P[15]), buf, 0, bw, x4, 0, g, g, H, W, flags=EPI_LRELU)
        N.check(L.k4_sft_train_fwd(N.f32(x4), g, N.f32(c), 32, n, g, *[N.f32(q) for q in P[18:26]], 0.2,
                                   N.C.c_void_p(buf.data_ptr() + 4 * (nf + 3 * g)), bw, N.stream()), 'k4_sft_train_fwd')
        SFTNet._conv(cache.fwd(P[16], P[17]), buf, 0, bw, out, 0, nf, nf, H, W, flags=EPI_RES, res=(t, 0, nf, 0.2))
        if ctx.direct is not None:
            ctx.save_for_backward(t, c, buf, x4)
            ctx.P = P
        else:
            ctx.save_for_backward(t, c, buf, x4, *P)
        ctx.cache = cache
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        t, c, buf, x4 = ctx.saved_tensors[:4]
        P = ctx.P if ctx.direct is not None else ctx.saved_tensors[4:]
        cache, acc = ctx.cache, ctx.acc
        _check_acc(acc)
        H, W, nf = t.shape
        g = P[8].shape[0]
        bw, n = nf + 4 * g, H * W
        go = go.contiguous().float()
        grads = [None] * 26
        if _NATIVE_RDB:                                   # the 19 launches of the host form below, issued by ONE native call; the weight gradients on a second stream
            L, dev = N.lib(), t.device
            d = ctx.desc
            order, sizes, offs, span, soff, nb0, nb1, shapes = _rdb_bwd_layout(P, n, nf, g, acc is None)
            pg = torch.empty([offs[-1]], dtype=torch.float32, device=dev)
            scr = torch.empty([soff[-1]], dtype=torch.float32, device=dev)
            pb, sb = pg.data_ptr(), scr.data_ptr()
            for q, (i, piece, shape) in enumerate(zip(order, pg.split(sizes), shapes)):
                grads[i] = piece if shape is None else piece.view(shape)
                ptr = pb + 4 * offs[q]
                if q < 10:
                    if q % 2 == 0:
                        d.dwdb[q // 2] = ptr                                          # [dW | dbias] of conv q/2 + 1
                elif q < 18:
                    d.gsft0[q - 10] = ptr
                else:
                    d.gsft1[q - 18] = ptr
            d.dwdb_span, d.dwdb_span_floats = pb, span
            d.gx0_add = go.data_ptr()
            d.gx0, d.G, d.gx4, d.ws0, d.ws1, d.g5 = (sb + 4 * o for o in soff[:6])
            d.ws0_bytes, d.ws1_bytes = nb0, nb1
            d.g5_from_gx0_add, d.fused_lrelu = 1, int(_FUSED_LRELU)          # g5 = 0.2 grad_out inside the call; the four LeakyReLU backward passes in epilogues
            if acc is None:
                d.gc0, d.gc1, d.gc_acc = sb + 4 * soff[6], sb + 4 * soff[7], None
            else:
                d.gc_acc = acc.data_ptr()
            keep = []
            for k in range(5):
                pk = cache.bwd(P[8 + 2 * k])
                assert pk.mode == 'bf16x6' and pk.k == 3 and pk.flags_extra == 0
                d.w_bwd[k], d.b_bwd[k] = pk.w.data_ptr(), pk.b.data_ptr()
                keep.append(pk)
            d.side_stream = _side_stream(dev)
            N.check(L.k4_rdb_train_bwd(N.C.byref(d), N.stream()), 'k4_rdb_train_bwd')
            gt = scr[:n * nf].view(H, W, nf)                                          # = go + the gradient through sft0 (added in the kernel's store)
            gc = None if acc is not None else (scr[soff[6]:soff[7]] + scr[soff[7]:soff[8]]).view(H, W, 32)
            if ctx.direct is not None:
                _hand_over_grads(ctx.direct, grads)
                return gt, gc, None, None, None
            return (gt, gc, None, None, *grads)

        G = torch.empty([H, W, bw], dtype=torch.float32, device=t.device)

        def accum(pk, src, s_off, s_stride, cout):                      # G[..., :cout] += dgrad: the output is its own residual
            SFTNet._conv(pk, src, s_off, s_stride, G, 0, bw, cout, H, W, flags=EPI_RES, res=(G, 0, bw, 1.0))

        # conv5: out = 0.2 conv5(buf) + t
        g5 = go * 0.2
        SFTNet._conv(cache.bwd(P[16]), g5, 0, nf, G, 0, bw, bw, H, W)                                                 # G = dgrad (every channel)
        grads[16], grads[17] = _wgrad(buf, 0, bw, bw, g5, 0, nf, nf, 3, H, W, P[16].shape, True)
        # xc1 = sft1(x4), x4 = lrelu(conv4(buf[:nf+3g]))
        gx4, gc1, grads[18:26] = _sft_bwd(x4, g, g, c, G, nf + 3 * g, bw, n, P[18:26])
        _lrelu_bwd(gx4, 0, g, x4, 0, g, n, g, gx4, 0, g)
        cin4 = nf + 3 * g
        accum(cache.bwd(P[14]), gx4, 0, g, cin4)                                                                      # G[:cin4] += dgrad
        grads[14], grads[15] = _wgrad(buf, 0, cin4, bw, gx4, 0, g, g, 3, H, W, P[14].shape, True)
        for k in (3, 2, 1):                                                                                         # x_k = lrelu(conv_k(buf[:off]))
            off = nf + (k - 1) * g
            _lrelu_bwd(G, off, bw, buf, off, bw, n, g, G, off, bw)
            accum(cache.bwd(P[6 + 2 * k]), G, off, bw, off)
            grads[6 + 2 * k], grads[7 + 2 * k] = _wgrad(buf, 0, off, bw, G, off, g, bw, 3, H, W, P[6 + 2 * k].shape, True)
        gx0, gc0, grads[0:8] = _sft_bwd(t, nf, nf, c, G, 0, bw, n, P[0:8])                                            # xc0 = sft0(t)
        gt = go + gx0.view(H, W, nf)
        gc = (gc0 + gc1).view(H, W, 32)
        if acc is not None:
            acc.add_(gc)
            gc = None
        if ctx.direct is not None:
            _hand_over_grads(ctx.direct, grads)
            return gt, gc, None, None, None
        return (gt, gc, None, None, *grads)


_TAP = None        # tests/debug/sr_rdb_debug3.py: callback(block, input, cond, output) per dense block


def _up2(t):
    """F.interpolate(scale_factor=2, mode='nearest') of an NHWC image (lib/sr_esrnet.py:461-463)."""
    return t.repeat_interleave(2, 0).repeat_interleave(2, 1)


def _params_hooked(net):
    """A parameter with a tensor hook or a post-accumulate-grad hook (DDP-style reducers, clipping hooks) needs its autograd edge: the direct
    gradient hand-over would never fire the hook."""
    for p in (net.k4_parameters() if hasattr(net, 'k4_parameters') else net.parameters()):
        if p._backward_hooks or getattr(p, '_post_accumulate_grad_hooks', None):
            return True
    return False


def forward_train(net, x, cond):
    """SFTNet.forward (lib/sr_esrnet.py:446-465) with autograd, every convolution on the HIP kernels.
    x [1,C,h,w], cond [1,num_cond,h,w] -> [1,3,s*h,s*w]."""
    assert x.shape[0] == 1 and cond.shape[0] == 1, 'batch 1 (as every call site of the reference)'
    cache = net._k4.setdefault('train_cache', _WeightCache())

    def conv(m, t, act=False):
        return K4Conv2d.apply(t, m.weight, m.bias, cache, act)

    def lrelu(t):
        return F.leaky_relu(t, 0.2)

    fused = os.environ.get('K4_TRAIN_SFT', 'fused') != 'convs'               # 'convs': one Function per convolution + elementwise autograd (A/B, tests)
    # The whole decoder as ONE autograd node whose forward and backward are launch tapes replayed by one native call each (lib/sr_tape.py): what the
    # direct gradient hand-over below needs (no parameter hooks, no graph capture), on the shapes the fused kernels cover.
    if (fused and _TAPE and _DIRECT_GRADS and _NATIVE_RDB and _FUSED_LRELU and _COND_ACC and _TAP is None and x.is_cuda
            and not torch.cuda.is_current_stream_capturing() and not _params_hooked(net)):
        from . import sr_tape
        anchor = next((p for p in net.k4_parameters() if p.requires_grad), None)
        if (anchor is not None or x.requires_grad or cond.requires_grad) and sr_tape.eligible(net, x, cond):
            prog = sr_tape.program_for(net, cache, x, cond)
            if prog is not None:
                return sr_tape.K4DecoderTape.apply(x, cond, prog, anchor)
    convs = net._k4.get(('train_convs', fused))
    if convs is None:                                                     # the SFT layers' 1x1 convolutions are not packed when fused
        convs = net._k4[('train_convs', fused)] = [m for name, m in net.named_modules()
                                                   if isinstance(m, torch.nn.Conv2d) and not (fused and '.SFT_' in '.' + name)]
    cache.prepack(convs)

    xi = x[0].permute(1, 2, 0).contiguous().float()
    ci = cond[0].permute(1, 2, 0).contiguous().float()
    feat = conv(net.conv_first, xi)
    cn = net.CondNet
    c = conv(cn[6], conv(cn[4], conv(cn[2], conv(cn[0], ci, True), True), True))
    # the fused SFT layers / dense blocks add their condition gradients into one buffer inside their kernels (see _CondFan)
    acc = torch.zeros_like(c) if fused and _COND_ACC and c.requires_grad and c.shape[2] == 32 else None
    # direct mode: the fused Functions take their parameters as one opaque list and write ``.grad`` themselves (no autograd edges to 438 of
    # the decoder's 458 parameter tensors: ~1.5 ms of host time per iteration in Function.apply and AccumulateGrad).  loss.backward() sees no
    # difference; torch.autograd.grad(..., params), parameter hooks and graph capture need the edges: K4_TRAIN_DIRECT_GRADS=0 / automatic.
    direct = fused and _DIRECT_GRADS and feat.requires_grad and not torch.cuda.is_current_stream_capturing() and not _params_hooked(net)
    if acc is not None:
        c = _CondFan.apply(c, acc)

    def sft_params(layer):
        return (layer.SFT_scale_conv0.weight, layer.SFT_scale_conv0.bias, layer.SFT_scale_conv1.weight, layer.SFT_scale_conv1.bias,
                layer.SFT_shift_conv0.weight, layer.SFT_shift_conv0.bias, layer.SFT_shift_conv1.weight, layer.SFT_shift_conv1.bias)

    def sft(layer, t, res=None, res_scale=1.0):                                # lib/sr_esrnet.py:120-123 (+ `* res_scale + res` when given)
        if fused and t.shape[2] in (32, 64) and c.shape[2] == 32:
            if direct and t.requires_grad:
                return K4SFTLayer.apply(t, c, acc, res, res_scale, list(sft_params(layer)))
            return K4SFTLayer.apply(t, c, acc, res, res_scale, *sft_params(layer))
        scale = conv(layer.SFT_scale_conv1, lrelu(conv(layer.SFT_scale_conv0, c)))
        shift = conv(layer.SFT_shift_conv1, lrelu(conv(layer.SFT_shift_conv0, c)))
        y = t * (scale + 1) + shift
        return y if res is None else y * res_scale + res

    def rdb(blk, t):                                                          # lib/sr_esrnet.py:149-158
        if fused and t.shape[2] in (32, 64) and c.shape[2] == 32 and blk.conv1.weight.shape[0] == 32:
            convs = [q for m in (blk.conv1, blk.conv2, blk.conv3, blk.conv4, blk.conv5) for q in (m.weight, m.bias)]
            if direct and t.requires_grad:
                return K4RDB.apply(t, c, cache, acc, [*sft_params(blk.sft0), *convs, *sft_params(blk.sft1)])
            return K4RDB.apply(t, c, cache, acc, *sft_params(blk.sft0), *convs, *sft_params(blk.sft1))
        xc0 = sft(blk.sft0, t)
        x1 = lrelu(conv(blk.conv1, xc0))
        x2 = lrelu(conv(blk.conv2, torch.cat((xc0, x1), 2)))
        x3 = lrelu(conv(blk.conv3, torch.cat((xc0, x1, x2), 2)))
        x4 = lrelu(conv(blk.conv4, torch.cat((xc0, x1, x2, x3), 2)))
        xc1 = sft(blk.sft1, x4)
        x5 = conv(blk.conv5, torch.cat((xc0, x1, x2, x3, xc1), 2))
        return x5 * 0.2 + t

    body = feat
    if _TAP is not None:
        _rdb = rdb

        def rdb(blk, t):
            o = _rdb(blk, t)
            _TAP(blk, t, c, o)
            return o
    for rr in net.body:                                                       # lib/sr_esrnet.py:176-182
        out = rdb(rr.rdb3, rdb(rr.rdb2, rdb(rr.rdb1, body)))
        body = sft(rr.sft0, out, body, 0.2)                                   # lib/sr_esrnet.py:181: out * 0.2 + x
    body = conv(net.conv_body, sft(net.sftbody, body)) + feat
    if net.scale > 1:
        body = conv(net.conv_up1, _up2(body), True)
        if _TAP is not None:
            _TAP('conv_up1', None, None, body)
        if net.scale == 4:
            body = conv(net.conv_up2, _up2(body), True)
            if _TAP is not None:
                _TAP('conv_up2', None, None, body)
    hr = conv(net.conv_hr, body, True)
    if _TAP is not None:
        _TAP('conv_hr', None, None, hr)
    out = conv(net.conv_last, hr)
    return out.permute(2, 0, 1).unsqueeze(0)


class GraphedDecoder:
    """``forward_train`` + its backward for ONE input shape as two hipGraphs (``torch.cuda.make_graphed_callables``).

    The decoder's training graph on a 64x64 patch is ~2500 launches of 4-30 us (229 convolutions x (pack, forward | pack, dgrad,
    wgrad, dbias) + the elementwise glue): 25 of the joint iteration's 42 ms were host time spent issuing them.  Every launch goes
    to torch's current stream and every buffer comes from torch's allocator, so the whole thing is capturable; the weight packers
    are kernels (k4_pack_conv_weight_bf16x6) and run INSIDE the graphs (``_WeightCache.always``): a replay after an optimizer step
    re-packs the updated weights.  Shapes other than the captured one fall back to the eager path (the caller keeps both)."""

    def __init__(self, net, x_shape, cond_shape):
        dev = next(net.parameters()).device
        cache = net._k4.setdefault('train_cache', _WeightCache())

        class _Fwd(torch.nn.Module):
            def __init__(self, net):
                super().__init__()
                self.net = net

            def forward(self, x, cond):
                return forward_train(self.net, x, cond)
        self.x_shape, self.cond_shape = tuple(x_shape), tuple(cond_shape)
        sample = (torch.rand(self.x_shape, device=dev, requires_grad=True), torch.rand(self.cond_shape, device=dev))
        cache.always = True
        try:
            self.fn = torch.cuda.make_graphed_callables(_Fwd(net), sample, allow_unused_input=True)    # scale 2 / dswise leave parameters unused
        finally:
            cache.always = False

    def matches(self, x, cond):
        return tuple(x.shape) == self.x_shape and tuple(cond.shape) == self.cond_shape and x.requires_grad and not cond.requires_grad

    def __call__(self, x, cond):
        return self.fn(x.contiguous(), cond.contiguous())


# ---------------------------------------------------------------------------------------------------------------------
# data parallelism: one bucketed all-reduce of the decoder's gradients
# ---------------------------------------------------------------------------------------------------------------------
def allreduce_gradients(params, group=None, average=True):
    """Sum (and average) ``p.grad`` of `params` over the process group with ONE all-reduce of a flat fp32 bucket.
    Parameters without a gradient on this rank contribute zeros (every rank must pass the same parameter list).
    Returns the number of bytes reduced."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    dev = params[0].device
    sizes = [p.numel() for p in params]
    # the bucket in a handful of launches: one concatenation in, one multi-tensor copy out (the decoder alone has 458 parameters: a copy per tensor each way was
    # ~900 launches per iteration, more than the decoder's own forward + backward pass)
    pieces = [p.grad.reshape(-1) if p.grad is not None else torch.zeros([n], dtype=torch.float32, device=dev) for p, n in zip(params, sizes)]
    if any(t.dtype != torch.float32 for t in pieces):
        pieces = [t.float() for t in pieces]
    flat = torch.cat(pieces)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 or (N.FORCE_COLLECTIVES and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)          # backend "nccl" = RCCL on the GPUs, gloo in the CPU tests
        if average:
            flat /= world
    have, views = [], []
    for p, g in zip(params, flat.split(sizes)):
        if p.grad is None:
            p.grad = g.view_as(p).clone()
        else:
            have.append(p.grad)
            views.append(g.view_as(p.grad))
    if have:
        torch._foreach_copy_(have, views)
    return flat.numel() * 4


def patch_parallel_step(net, optimizer, loss_fn, group=None):
    """One data-parallel SR step: loss_fn() evaluates this rank's patch (forward through `net`), gradients are exchanged with
    one all-reduce, the optimizer steps on identical averaged gradients on every rank (replicas stay bit-identical)."""
    optimizer.zero_grad(set_to_none=True)
    loss = loss_fn()
    loss.backward()
    allreduce_gradients(list(net.parameters()), group=group)
    optimizer.step()
    return loss.detach()
