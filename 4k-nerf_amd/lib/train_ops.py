"""Training-graph operators of the marcher on HIP kernels (csrc/k4_train.hip; SURVEY.md 8f rank 1 "MLP bwd", 3.4).

``rgbnet_sigmoid``      the colour MLP + sigmoid of ``DirectMPIGO.forward`` / ``DirectVoxGO.forward`` under autograd
                        (/root/reference/lib/dmpigo.py:375-379, lib/dvgo.py:407-412: ``torch.sigmoid(self.rgbnet(feat))``,
                        differentiated by PyTorch autograd over library GEMMs): ONE launch forward (activations saved),
                        two launches backward (everything + the ordered sum of the per-workgroup weight-gradient partials).
``flatten_eff_distloss`` the distortion loss of the joint training step (run_sr.py:976-988; third-party
                        ``torch_efficient_distloss``, not vendored upstream): value and gradient in one launch.

``rgbnet_sigmoid_layers`` the same expression for Linear-ReLU stacks OUTSIDE those kernels' shapes (width not in {32, 64, 128}, more than one
                        hidden->hidden layer, dim0 > 64; ``rgbnet_supported`` tells which): layer by layer on the exact-fp32 1x1 MFMA
                        convolution, INFERENCE ONLY -- the models route such shapes here (lib/dvgo.py::_k4_rgbnet_sigmoid).

There is no CPU path and no PyTorch path: CPU tensors raise ``K4Error``; so do these out-of-shape stacks under autograd, and anything that is
not a biased Linear-ReLU stack ending in 3 outputs (another activation, a bias-free Linear, a layer wider than 128 / more than 192 inputs).
"""

import torch
from torch import nn

from .. import _native as N


def _linears(rgbnet):
    return [m for m in rgbnet.modules() if isinstance(m, nn.Linear)]


def rgbnet_supported(rgbnet):
    """True when `rgbnet` is Linear-ReLU-[Linear-ReLU]-Linear(->3) with a shape k4_rgbnet_* covers."""
    lins = _linears(rgbnet)
    if len(lins) not in (2, 3) or lins[-1].out_features != 3:
        return False
    acts = [m for m in rgbnet.modules() if not isinstance(m, (nn.Linear, nn.Sequential))]
    if not all(isinstance(a, nn.ReLU) for a in acts):
        return False
    width = lins[0].out_features
    if any(l.bias is None for l in lins) or any(l.in_features != width for l in lins[1:]) or any(l.out_features != width for l in lins[1:-1]):
        return False
    return N.lib().k4_rgbnet_bwd_workspace_bytes(1, lins[0].in_features, width, len(lins) - 2) >= 0


class RgbNetSigmoid(torch.autograd.Function):
    """rgb = sigmoid(W3 relu(W2 relu(W1 x + b1) + b2) + b3 (+ add)); w2/b2 None for a 2-layer rgbnet."""

    @staticmethod
    def forward(ctx, x, add, w1, b1, w2, b2, w3, b3):
        x = x.detach().float().contiguous()
        n, dim0 = x.shape
        width = w1.shape[0]
        nh = 0 if w2 is None else 1
        dev = x.device
        ws = [t.detach().contiguous() if t is not None else None for t in (w1, b1, w2, b2, w3, b3)]
        addc = None if add is None else add.detach().float().contiguous()
        need = any(ctx.needs_input_grad)
        h1 = torch.empty([n, width], dtype=torch.float32, device=dev) if need else None
        h2 = torch.empty([n, width], dtype=torch.float32, device=dev) if need and nh else None
        rgb = torch.empty([n, 3], dtype=torch.float32, device=dev)
        N.check(N.lib().k4_rgbnet_fwd(N.f32(x), n, dim0, width, nh, N.f32(ws[0]), N.f32(ws[1]),
                                      None if ws[2] is None else N.f32(ws[2]), None if ws[3] is None else N.f32(ws[3]),
                                      N.f32(ws[4]), N.f32(ws[5]), None if addc is None else N.f32(addc),
                                      None if h1 is None else N.f32(h1), None if h2 is None else N.f32(h2), N.f32(rgb), N.stream()),
                'k4_rgbnet_fwd')
        if need:
            ctx.save_for_backward(x, h1, h2, rgb, ws[0], ws[2], ws[4])
            ctx.has_add = add is not None
        return rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_rgb):
        x, h1, h2, rgb, w1, w2, w3 = ctx.saved_tensors
        n, dim0 = x.shape
        width = w1.shape[0]
        nh = 0 if w2 is None else 1
        dev = x.device
        g = grad_rgb.float().contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gl = torch.empty([n, 3], dtype=torch.float32, device=dev) if ctx.has_add and ctx.needs_input_grad[1] else None
        gw1, gb1 = torch.empty_like(w1), torch.empty([width], dtype=torch.float32, device=dev)
        gw2, gb2 = (torch.empty_like(w2), torch.empty([width], dtype=torch.float32, device=dev)) if nh else (None, None)
        gw3, gb3 = torch.empty_like(w3), torch.empty([3], dtype=torch.float32, device=dev)
        L = N.lib()
        wsb = int(L.k4_rgbnet_bwd_workspace_bytes(n, dim0, width, nh))
        work = torch.empty([max(wsb, 4) // 4], dtype=torch.float32, device=dev)
        N.check(L.k4_rgbnet_bwd(N.f32(x), n, dim0, width, nh, N.f32(w1), None if w2 is None else N.f32(w2), N.f32(w3),
                                N.f32(h1), None if h2 is None else N.f32(h2), N.f32(rgb), N.f32(g),
                                None if gx is None else N.f32(gx), None if gl is None else N.f32(gl),
                                N.f32(gw1), N.f32(gb1), None if gw2 is None else N.f32(gw2), None if gb2 is None else N.f32(gb2),
                                N.f32(gw3), N.f32(gb3), N.f32(work), wsb, N.stream()), 'k4_rgbnet_bwd')
        return gx, gl, gw1, gb1, gw2, gb2, gw3, gb3


def rgbnet_sigmoid(rgbnet, x, add=None):
    """``torch.sigmoid(rgbnet(x) [+ add])`` on the HIP kernels (x [n, dim0]; add [n, 3] = k0_diffuse or None)."""
    if not x.is_cuda:
        raise N.K4Error('rgbnet_sigmoid: tensor must be on the GPU (no CPU path exists for this op)')
    lins = _linears(rgbnet)
    w2, b2 = (lins[1].weight, lins[1].bias) if len(lins) == 3 else (None, None)
    return RgbNetSigmoid.apply(x, add, lins[0].weight, lins[0].bias, w2, b2, lins[-1].weight, lins[-1].bias)


_GENERIC_PACKS = {}
CHECK_DISTLOSS = False      # True: check the two preconditions of the distortion loss' fast path on every call (costs a host sync; debugging)


def rgbnet_sigmoid_layers(rgbnet, x, add=None):
    """``torch.sigmoid(rgbnet(x) [+ add])`` for Linear-ReLU stacks OUTSIDE the shapes of k4_rgbnet_fwd (more hidden layers, other widths
    up to 128, dim0 up to 192): every Linear is one launch of the exact-fp32 MFMA convolution kernel as a 1x1 layer over the samples
    (k4_conv2d_nhwc, ReLU = its LeakyReLU epilogue with slope 0).  Inference only: under autograd these shapes raise."""
    # (checked in the CALLER's grad mode: as a `torch.no_grad()`-decorated function this test could never fire and a training loop would have
    #  received a constant -- found by tests/test_train_ops_gpu.py::test_rgbnet_layer_by_layer_path_and_rejected_shapes in round 5)
    if torch.is_grad_enabled() and (x.requires_grad or (add is not None and add.requires_grad) or any(p.requires_grad for p in rgbnet.parameters())):
        raise N.K4Error('training an rgbnet of this shape is outside k4_rgbnet_fwd / k4_rgbnet_bwd (Linear-ReLU stack of <= 3 layers, width 32 / 64 / 128)')
    with torch.no_grad():
        return _rgbnet_sigmoid_layers(rgbnet, x, add)


def _rgbnet_sigmoid_layers(rgbnet, x, add):
    from .sr_esrnet import _Packed, EPI_LRELU
    def flat(m):                                                   # execution order; a shared activation instance counts every time it is applied
        return [q for c in m.children() for q in flat(c)] if isinstance(m, nn.Sequential) else [m]
    mods = flat(rgbnet)
    lins = [m for m in mods if isinstance(m, nn.Linear)]
    ok = bool(lins) and all(isinstance(m, (nn.Linear, nn.ReLU)) for m in mods) and lins[-1].out_features == 3 and mods[-1] is lins[-1] \
        and all(l.bias is not None and l.out_features <= 128 and l.in_features <= 192 for l in lins) \
        and len(mods) == 2 * len(lins) - 1 and all(isinstance(m, nn.Linear) == (i % 2 == 0) for i, m in enumerate(mods))
    if not ok:
        raise N.K4Error('rgbnet is not a Linear-ReLU stack the HIP kernels cover (no PyTorch fallback)')
    n = x.shape[0]
    dev = x.device
    if n == 0:
        return torch.empty([0, 3], dtype=torch.float32, device=dev)
    key = tuple((l.weight.data_ptr(), l.weight._version, l.bias._version) for l in lins)
    hit = _GENERIC_PACKS.get(id(rgbnet))
    if hit is None or hit[0] != key:
        hit = (key, [_Packed(l.weight.detach()[:, :, None, None], l.bias, 'fp32') for l in lins])
        if len(_GENERIC_PACKS) > 8:
            _GENERIC_PACKS.clear()
        _GENERIC_PACKS[id(rgbnet)] = hit
    Wd = 256                                                       # the samples as an image of 256-pixel rows (1x1 layer: any shape does)
    Hh = (n + Wd - 1) // Wd
    cur = torch.zeros([Hh * Wd, lins[0].in_features], dtype=torch.float32, device=dev)
    cur[:n] = x.detach().float()
    L = N.lib()
    for i, (l, pk) in enumerate(zip(lins, hit[1])):
        cout = l.out_features
        nxt = torch.empty([Hh * Wd, cout], dtype=torch.float32, device=dev)
        N.check(L.k4_conv2d_nhwc(N.f32(cur), l.in_features, l.in_features, N.f32(pk.w), N.f32(pk.b), 1, N.f32(nxt), cout, cout, Hh, Wd,
                                 EPI_LRELU if i + 1 < len(lins) else 0, 0.0, None, 0, 0.0, None, 0, N.stream()), 'k4_conv2d_nhwc (rgbnet layer)')
        cur = nxt
    logit = cur[:n]
    return torch.sigmoid(logit if add is None else logit + add.detach().float())


class RgbnetInputMPI(torch.autograd.Function):
    """The colour MLP's input of DirectMPIGO's training forward, ``cat([vox_emb, pe_emb, viewdirs_emb[ray_id]], -1)`` with both embeddings built from ray_pts /
    viewdirs as lib/dmpigo.py:360-374 does, in ONE launch (k4_rgbnet_input_mpi) instead of 16 PyTorch ops.  Only vox_emb carries a gradient."""

    @staticmethod
    def forward(ctx, vox_emb, ray_pts, viewdirs, ray_id, xyz_min, xyz_max, posfreq, viewfreq):
        n, C = vox_emb.shape
        P, V = int(posfreq.numel()), int(viewfreq.numel())
        dim0 = C + 3 + 6 * P + 3 + 6 * V
        x = torch.empty([n, dim0], dtype=torch.float32, device=vox_emb.device)
        N.check(N.lib().k4_rgbnet_input_mpi(N.f32(vox_emb.contiguous()), C, N.f32(ray_pts.contiguous()), N.f32(viewdirs.contiguous()), N.ptr(ray_id.contiguous()), n,
                                            N.f32(xyz_min.contiguous()), N.f32(xyz_max.contiguous()), N.f32(posfreq.contiguous()) if P else None, P,
                                            N.f32(viewfreq.contiguous()) if V else None, V, N.f32(x), dim0, N.stream()), 'k4_rgbnet_input_mpi')
        ctx.C = C
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gx):
        return gx[:, :ctx.C].contiguous(), None, None, None, None, None, None, None


def rgbnet_input_mpi(vox_emb, ray_pts, viewdirs, ray_id, xyz_min, xyz_max, posfreq, viewfreq):
    """-> [n, dim0] MLP input, or None when the fused form does not apply (CPU tensors, other dtypes): the caller keeps the op-by-op form."""
    ts = (vox_emb, ray_pts, viewdirs, xyz_min, xyz_max, posfreq, viewfreq)
    if not all(t.is_cuda and t.dtype == torch.float32 for t in ts) or ray_id.dtype != torch.int64 or vox_emb.dim() != 2:
        return None
    return RgbnetInputMPI.apply(vox_emb, ray_pts, viewdirs, ray_id, xyz_min, xyz_max, posfreq, viewfreq)


class FlattenEffDistLoss(torch.autograd.Function):
    """sum_rays [ sum_ij w_i w_j |s_i - s_j| + 1/3 sum_i w_i^2 interval ] / (ray_id.max() + 1), gradient w.r.t. w only.
    ray_id: int64, on the device, ASCENDING (samples grouped by ray in marching order, as the marcher emits them): the kernel finds a
    ray's segment by binary search."""

    @staticmethod
    def forward(ctx, w, s, interval, ray_id, n_rays_bound):
        if not w.is_cuda:
            raise N.K4Error('flatten_eff_distloss: tensor must be on the GPU (no CPU path exists for this op)')
        if torch.is_tensor(interval):
            raise NotImplementedError('per-sample interval tensors: run_sr.py:985 passes the scalar 1/n_max')
        if ray_id.dtype != torch.int64 or not ray_id.is_cuda or ray_id.shape != w.shape:
            raise ValueError('flatten_eff_distloss: ray_id must be an int64 device tensor of the shape of w (sorted ascending)')
        wc, sc, idx = w.detach().float().contiguous(), s.detach().float().contiguous(), ray_id.contiguous()
        n = wc.shape[0]
        if n == 0:
            ctx.empty = True
            return wc.new_zeros([])
        ctx.empty = False
        n_rays_t = (idx[-1] + 1).to(torch.float32)               # the package's normaliser ray_id.max() + 1 (sorted: the last entry); stays on the device
        # the launch bound must be a host integer: any bound > max(ray_id) is correct (rays without samples contribute 0).  Without one
        # from the caller it costs a host synchronisation per step.
        n_rays = int(n_rays_bound) if n_rays_bound is not None else int(idx[-1]) + 1
        if CHECK_DISTLOSS:              # debug check of the two preconditions of the fast path (costs a host sync)
            if n > 1 and bool((idx[1:] < idx[:-1]).any()):
                raise ValueError('flatten_eff_distloss: ray_id is not ascending')
            if int(idx[-1]) >= n_rays:
                raise ValueError(f'flatten_eff_distloss: n_rays={n_rays} <= max(ray_id)={int(idx[-1])}')
        ray_loss = torch.empty([n_rays], dtype=torch.float32, device=wc.device)
        grad = torch.zeros_like(wc)                                # zeros: samples of rays beyond a too-small bound get no gradient, not garbage
        N.check(N.lib().k4_distortion_loss(N.f32(wc), N.f32(sc), N.ptr(idx), n, n_rays, float(interval), N.f32(ray_loss), N.f32(grad),
                                           N.stream()), 'k4_distortion_loss')
        ctx.save_for_backward(grad, n_rays_t)
        return ray_loss.sum() / n_rays_t

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        if ctx.empty:
            return None, None, None, None, None
        grad, n_rays_t = ctx.saved_tensors
        return grad * (grad_back / n_rays_t), None, None, None, None


_JL_ACC = {}         # device -> 8 fp64 accumulators of k4_joint_losses_fwd (all zero between calls: the finishing kernel clears them)


class JointSmallLosses(torch.autograd.Function):
    """The elementwise loss terms of the joint iteration (run_sr.py:877-995: photo L1, decoder L1 + its PSNR, last-transmittance entropy, per-sample colour)
    as ONE node: k4_joint_losses_fwd -> (their sum, the five terms [photo, l1, psnr_sr, entropy_last, rgbper]); k4_joint_losses_bwd -> the gradients w.r.t.
    rgb_feature, rgb_sr, alphainv_last and raw_rgb in one launch.  ``alphainv_last`` / ``raw_rgb`` None switch their term off (weight 0 in the config)."""

    @staticmethod
    def forward(ctx, rgb_feature, rgb_sr, alphainv_last, raw_rgb, target, target_4x, weights, ray_id, w_main, w_ent, w_per):
        dev = rgb_feature.device
        feat, tgt, t4 = rgb_feature.detach().contiguous(), target.contiguous(), target_4x.contiguous()
        n_rays, n_hr = int(feat.shape[0]), int(t4.shape[0])
        sr = rgb_sr.detach()
        if tuple(sr.shape[:2]) != (1, 3) or sr.shape[2] * sr.shape[3] != n_hr or feat.shape != tgt.shape or feat.shape[1] != 3 or t4.shape[1] != 3:
            raise ValueError('joint losses: rgb_sr [1, 3, H, W] against target_4x [H * W, 3], rgb_feature / target [n_rays, 3]')
        W = int(sr.shape[3])
        if not ((sr.stride(1) == 1 and sr.stride(3) == 3 and sr.stride(2) == 3 * W) or sr.is_contiguous()):
            sr = sr.contiguous()
        cs, ps = (1, 3) if (sr.stride(1) == 1 and sr.stride(3) == 3) else (n_hr, 1)
        d = N.JointLosses()
        d.rgb_feature, d.target, d.n_rays = feat.data_ptr(), tgt.data_ptr(), n_rays
        d.rgb_sr, d.target_4x, d.n_hr, d.sr_cstride, d.sr_pstride = sr.data_ptr(), t4.data_ptr(), n_hr, cs, ps
        keep = [feat, tgt, t4, sr]
        if alphainv_last is not None:
            a = alphainv_last.detach().contiguous()
            d.alphainv_last = a.data_ptr()
            keep.append(a)
        if raw_rgb is not None:
            raw, w, rid = raw_rgb.detach().contiguous(), weights.detach().contiguous(), ray_id.contiguous()
            if rid.dtype != torch.int64 or raw.shape != (w.shape[0], 3) or rid.shape != w.shape:
                raise ValueError('joint losses: raw_rgb [n, 3], weights [n], ray_id int64 [n]')
            d.raw_rgb, d.weights, d.ray_id, d.n_pts = raw.data_ptr() or None, w.data_ptr() or None, rid.data_ptr() or None, int(w.shape[0])
            if d.n_pts == 0:
                d.raw_rgb = None
            keep += [raw, w, rid]
        for t in keep:
            if not t.is_cuda or t.dtype != (torch.int64 if t is ray_id or (raw_rgb is not None and t is keep[-1]) else torch.float32):
                raise N.K4Error('joint losses: fp32 device tensors (no CPU path exists for this op)')
        d.weight_main, d.weight_entropy_last, d.weight_rgbper = float(w_main), float(w_ent), float(w_per)
        acc = _JL_ACC.get(dev)
        if acc is None:
            acc = _JL_ACC[dev] = torch.zeros([8], dtype=torch.float64, device=dev)
        terms = torch.empty([5], dtype=torch.float32, device=dev)
        total = torch.empty([], dtype=torch.float32, device=dev)
        try:
            N.check(N.lib().k4_joint_losses_fwd(N.C.byref(d), N.ptr(acc), N.f32(terms), N.f32(total), N.stream()), 'k4_joint_losses_fwd')
        except Exception:
            _JL_ACC.pop(dev, None)
            raise
        ctx.desc, ctx.keep, ctx.sr_like = d, keep, sr
        ctx.shapes = (rgb_feature.shape, alphainv_last.shape if alphainv_last is not None else None, raw_rgb.shape if raw_rgb is not None else None)
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_total, _g_terms):
        d, sr = ctx.desc, ctx.sr_like
        need = ctx.needs_input_grad
        dev = sr.device
        g = g_total.detach().to(torch.float32).contiguous()
        g_feat = torch.empty(ctx.shapes[0], dtype=torch.float32, device=dev) if need[0] else None
        g_sr = torch.empty_strided(sr.shape, sr.stride(), dtype=torch.float32, device=dev) if need[1] else None
        g_a = torch.empty(ctx.shapes[1], dtype=torch.float32, device=dev) if (need[2] and ctx.shapes[1] is not None) else None
        g_raw = None
        if need[3] and ctx.shapes[2] is not None:
            g_raw = torch.empty(ctx.shapes[2], dtype=torch.float32, device=dev) if d.n_pts > 0 and d.raw_rgb else torch.zeros(ctx.shapes[2], dtype=torch.float32, device=dev)
        vp = lambda t: None if t is None else N.C.c_void_p(t.data_ptr())          # (g_sr carries rgb_sr's strides: NHWC for the training tape's result)
        N.check(N.lib().k4_joint_losses_bwd(N.C.byref(d), N.f32(g), vp(g_feat), vp(g_sr), vp(g_a), vp(g_raw) if d.raw_rgb else None, N.stream()), 'k4_joint_losses_bwd')
        return g_feat, g_sr, g_a, g_raw, None, None, None, None, None, None, None


def joint_small_losses(rgb_feature, rgb_sr, alphainv_last, raw_rgb, target, target_4x, weights, ray_id, w_main, w_ent, w_per):
    """-> (photo + l1 + entropy_last + rgbper as one differentiable scalar, detached [photo, l1, psnr_sr, entropy_last, rgbper])."""
    return JointSmallLosses.apply(rgb_feature, rgb_sr, alphainv_last, raw_rgb, target, target_4x, weights, ray_id, w_main, w_ent, w_per)


def flatten_eff_distloss(w, s, interval, ray_id, n_rays=None):
    """Drop-in for ``torch_efficient_distloss.flatten_eff_distloss`` as run_sr.py:985 calls it.  ``n_rays`` (optional, not in the
    package's signature): the number of rays of the batch -- any integer > max(ray_id) -- saves the host synchronisation that reading
    ``ray_id.max()`` back costs."""
    return FlattenEffDistLoss.apply(w, s, interval, ray_id, n_rays)
