"""Training path of the VC-Decoder: ``SFTNet`` forward + backward with every convolution on the gfx950
matrix cores, and the gradient exchange of patch-parallel data parallelism.

The reference back-propagates its losses through ``SFTNet`` with PyTorch autograd over cuDNN convolutions
(/root/reference/run_sr.py:869-1014; modules lib/sr_esrnet.py:112-182, forward :446-465).  Here the graph is evaluated on NHWC
``[H, W, C]`` tensors; each ``nn.Conv2d`` call (3x3 and 1x1: all of the network's FLOPs) is the autograd Function ``K4Conv2d``:
    forward : k4_conv2d_nhwc_bf16x6           (csrc/k4_sr.hip: exact 3-term bf16 splits, 6 MFMA products, fp32-equivalent)
    dgrad   : the same kernel on W'[ci][co][2-dy][2-dx] = W[co][ci][dy][dx]   (a "same" convolution of dY with the flipped,
              transposed filter; packed once per weight version)
    wgrad   : k4_conv2d_wgrad_bf16x6          (csrc/k4_sr_bwd.hip: MFMA GEMM over the pixels, split-K + fp32 atomics)
    dbias   : k4_conv2d_bias_grad
The elementwise glue between the convolutions (LeakyReLU, SFT modulation x*(scale+1)+shift, residual scaling, channel concat,
nearest x2 upsampling) stays on PyTorch ops and their autograd.  There is no CPU path.

``allreduce_gradients`` is the data-parallel exchange: each rank back-propagates its own 64x64 patch (run_sr.py:829-835), the
458 gradient tensors (15.8 MB) are flattened into one bucket and summed with ONE all-reduce (RCCL over xGMI on the GPUs:
ring all-reduce of 15.8 MB moves 2*(N-1)/N * 15.8 MB per rank = 27.7 MB at N=8, ~0.2 ms at the per-link rate) and averaged.
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .. import _native as N
from .sr_esrnet import _Packed, SFTNet


class _WeightCache:
    """Packed forward / transposed (dgrad) weights per (tensor, version)."""

    def __init__(self):
        self._c = {}
        self.always = False        # True: pack on every call (hipGraph capture: the packing kernels must be part of the graph, see GraphedDecoder)

    def _get(self, kind, weight, bias, make):
        if self.always:
            return make()
        key = (kind, weight.data_ptr(), weight._version, None if bias is None else bias._version, str(weight.device))
        hit = self._c.get((kind, weight.data_ptr()))
        if hit is None or hit[0] != key:
            hit = (key, make())
            self._c[(kind, weight.data_ptr())] = hit
        return hit[1]

    def fwd(self, weight, bias):
        return self._get('f', weight, bias, lambda: _Packed.native(weight, bias))

    def bwd(self, weight):
        # dgrad operand: the filter flipped in both taps axes and transposed to [cin, cout, k, k], packed by the same kernel
        return self._get('b', weight, None, lambda: _Packed.native(weight, None, dgrad=True))


class K4Conv2d(torch.autograd.Function):
    """stride-1 "same" convolution of an NHWC [H, W, Cin] image with an nn.Conv2d weight [Cout, Cin, k, k] (k = 1 | 3)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache):
        if not x.is_cuda:
            raise N.K4Error('K4Conv2d: the MI355X-native decoder has no CPU path')
        x = x.contiguous()
        H, W, cin = x.shape
        cout, cin_w, k, _ = weight.shape
        assert cin == cin_w and x.dtype == torch.float32
        y = torch.empty([H, W, cout], dtype=torch.float32, device=x.device)
        SFTNet._conv(cache.fwd(weight, bias), x, 0, cin, y, 0, cout, cout, H, W)
        ctx.save_for_backward(x, weight)
        ctx.cache, ctx.has_bias = cache, bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous().float()
        H, W, cin = x.shape
        cout, _, k, _ = weight.shape
        gx = gw = gb = None
        L = N.lib()
        if ctx.needs_input_grad[0]:
            gx = torch.empty([H, W, cin], dtype=torch.float32, device=x.device)
            SFTNet._conv(ctx.cache.bwd(weight), gy, 0, cout, gx, 0, cin, cin, H, W)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and want_b:
            # dW and dbias from one buffer: one zero-fill + one launch (the wgrad workgroups of tap 0 / input block 0 also sum dY)
            nw = weight.numel()
            buf = torch.empty([nw + cout], dtype=torch.float32, device=x.device)
            N.check(L.k4_conv2d_wgrad_dbias_bf16x6(N.f32(x), cin, cin, N.f32(gy), cout, cout, k, H, W, N.f32(buf), N.stream()),
                    'k4_conv2d_wgrad_dbias_bf16x6')
            gw, gb = buf[:nw].view(weight.shape), buf[nw:]
        elif ctx.needs_input_grad[1]:
            gw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)          # zeroed by the entry point
            N.check(L.k4_conv2d_wgrad_bf16x6(N.f32(x), cin, cin, N.f32(gy), cout, cout, k, H, W, N.f32(gw), N.stream()),
                    'k4_conv2d_wgrad_bf16x6')
        elif want_b:
            gb = torch.empty([cout], dtype=torch.float32, device=x.device)
            N.check(L.k4_conv2d_bias_grad(N.f32(gy), cout, cout, H * W, N.f32(gb), N.stream()), 'k4_conv2d_bias_grad')
        return gx, gw, gb, None


class K4SFTLayer(torch.autograd.Function):
    """SFTLayer (lib/sr_esrnet.py:112-123) of an NHWC image: ``x * (scale(cond) + 1) + shift(cond)`` with both 1x1-convolution pairs,
    LeakyReLU and the modulation in ONE launch forward (k4_sft_train_fwd) and two backward (k4_sft_train_bwd: grad_x, grad_cond, the
    eight weight / bias gradients).  As four K4Conv2d Functions + elementwise autograd a layer was ~45 launches per iteration."""

    @staticmethod
    def forward(ctx, x, cond, w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h):
        if not x.is_cuda:
            raise N.K4Error('K4SFTLayer: the MI355X-native decoder has no CPU path')
        x, cond = x.contiguous().float(), cond.contiguous().float()
        H, W, C = x.shape
        assert cond.shape == (H, W, 32) and w0s.shape[:2] == (32, 32) and w1s.shape[:2] == (C, 32)
        ws = [t.detach().contiguous() for t in (w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h)]
        y = torch.empty_like(x)
        N.check(N.lib().k4_sft_train_fwd(N.f32(x), C, N.f32(cond), 32, H * W, C, *[N.f32(t) for t in ws], 0.2, N.f32(y), C, N.stream()),
                'k4_sft_train_fwd')
        ctx.save_for_backward(x, cond, *ws)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, cond, w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h = ctx.saved_tensors
        gy = gy.contiguous().float()
        H, W, C = x.shape
        n = H * W
        L = N.lib()
        gx, gc = torch.empty_like(x), torch.empty_like(cond)
        g = [torch.empty_like(t) for t in (w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h)]
        nbytes = int(L.k4_sft_train_bwd_workspace_bytes(n, C))
        ws = torch.empty([nbytes // 4], dtype=torch.float32, device=x.device)
        N.check(L.k4_sft_train_bwd(N.f32(x), C, N.f32(cond), 32, N.f32(gy), C, n, C, N.f32(w0s), N.f32(b0s), N.f32(w1s), N.f32(b1s),
                                   N.f32(w0h), N.f32(b0h), N.f32(w1h), 0.2, N.f32(gx), N.f32(gc), *[N.f32(t) for t in g],
                                   N.f32(ws), nbytes, N.stream()), 'k4_sft_train_bwd')
        return (gx, gc, *g)


def _up2(t):
    """F.interpolate(scale_factor=2, mode='nearest') of an NHWC image (lib/sr_esrnet.py:461-463)."""
    return t.repeat_interleave(2, 0).repeat_interleave(2, 1)


def forward_train(net, x, cond):
    """SFTNet.forward (lib/sr_esrnet.py:446-465) with autograd, every convolution on the HIP kernels.
    x [1,C,h,w], cond [1,num_cond,h,w] -> [1,3,s*h,s*w]."""
    assert x.shape[0] == 1 and cond.shape[0] == 1, 'batch 1 (as every call site of the reference)'
    cache = net._k4.setdefault('train_cache', _WeightCache())

    def conv(m, t):
        return K4Conv2d.apply(t, m.weight, m.bias, cache)

    def lrelu(t):
        return F.leaky_relu(t, 0.2)

    xi = x[0].permute(1, 2, 0).contiguous().float()
    ci = cond[0].permute(1, 2, 0).contiguous().float()
    feat = conv(net.conv_first, xi)
    cn = net.CondNet
    c = conv(cn[6], lrelu(conv(cn[4], lrelu(conv(cn[2], lrelu(conv(cn[0], ci)))))))

    fused_sft = os.environ.get('K4_TRAIN_SFT', 'fused') != 'convs'            # 'convs': the four-convolution form (A/B, tests)

    def sft(layer, t):                                                        # lib/sr_esrnet.py:120-123
        if fused_sft and t.shape[2] in (32, 64) and c.shape[2] == 32:
            return K4SFTLayer.apply(t, c, layer.SFT_scale_conv0.weight, layer.SFT_scale_conv0.bias, layer.SFT_scale_conv1.weight,
                                    layer.SFT_scale_conv1.bias, layer.SFT_shift_conv0.weight, layer.SFT_shift_conv0.bias,
                                    layer.SFT_shift_conv1.weight, layer.SFT_shift_conv1.bias)
        scale = conv(layer.SFT_scale_conv1, lrelu(conv(layer.SFT_scale_conv0, c)))
        shift = conv(layer.SFT_shift_conv1, lrelu(conv(layer.SFT_shift_conv0, c)))
        return t * (scale + 1) + shift

    def rdb(blk, t):                                                          # lib/sr_esrnet.py:149-158
        xc0 = sft(blk.sft0, t)
        x1 = lrelu(conv(blk.conv1, xc0))
        x2 = lrelu(conv(blk.conv2, torch.cat((xc0, x1), 2)))
        x3 = lrelu(conv(blk.conv3, torch.cat((xc0, x1, x2), 2)))
        x4 = lrelu(conv(blk.conv4, torch.cat((xc0, x1, x2, x3), 2)))
        xc1 = sft(blk.sft1, x4)
        x5 = conv(blk.conv5, torch.cat((xc0, x1, x2, x3, xc1), 2))
        return x5 * 0.2 + t

    body = feat
    for rr in net.body:                                                       # lib/sr_esrnet.py:176-182
        out = rdb(rr.rdb3, rdb(rr.rdb2, rdb(rr.rdb1, body)))
        body = sft(rr.sft0, out) * 0.2 + body
    body = conv(net.conv_body, sft(net.sftbody, body)) + feat
    if net.scale > 1:
        body = lrelu(conv(net.conv_up1, _up2(body)))
        if net.scale == 4:
            body = lrelu(conv(net.conv_up2, _up2(body)))
    out = conv(net.conv_last, lrelu(conv(net.conv_hr, body)))
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
            self.fn = torch.cuda.make_graphed_callables(_Fwd(net), sample)
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
    flat = torch.zeros([sum(sizes)], dtype=torch.float32, device=dev)
    off = 0
    for p, n in zip(params, sizes):
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)          # backend "nccl" = RCCL on the GPUs, gloo in the CPU tests
        if average:
            flat /= world
    off = 0
    for p, n in zip(params, sizes):
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
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
