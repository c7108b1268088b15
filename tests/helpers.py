"""Shared test helpers (golden loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_march_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    kw = json.loads(str(z['model_kwargs_json']))
    for k in ('xyz_min', 'xyz_max'):
        kw[k] = np.asarray(kw[k], dtype=np.float32)
    sd, inp, out = {}, {}, {}
    for k in z.files:
        if k.startswith('sd/'):
            sd[k[3:]] = torch.from_numpy(z[k])
        elif k.startswith('in/'):
            inp[k[3:]] = torch.from_numpy(z[k])
        elif k.startswith('out/'):
            out[k[4:]] = torch.from_numpy(z[k])
    return {'model_class': str(z['model_class']), 'model_kwargs': kw, 'model_state_dict': sd,
            'render_kwargs': json.loads(str(z['render_kwargs_json'])), 'rays': inp, 'out': out}


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


# ---------------------------------------------------------------------------------------------------------------------
# Reference-made vectors of the native kernels (tests/golden/native_*.npz, optim_ref.npz): outputs of the reference's OWN
# lib/cuda/*.cu compiled for gfx950 (oracle/build_ref.py) and run on an MI355X (oracle/gen_native_golden.py).
# Contract: integer / boolean outputs and everything that feeds a mask decision are BIT-EXACT; floating-point values that pass
# through compiler-chosen FMA contraction, division or libm (the reference was built with hipcc's defaults, nvcc would choose
# its own) agree to a few ulp -- the tolerance of each key is spelled out below.
# ---------------------------------------------------------------------------------------------------------------------
NATIVE_EXACT = ['t_min', 't_max', 'n_samples', 'rays_start', 'aabb/mask_outbbox', 'aabb/ray_id', 'aabb/step_id', 'aabb/N_steps',
                'aabb/sp_t_min', 'aabb/sp_t_max', 'ndc37/pts', 'ndc37/mask_outbbox', 'ndc256/pts', 'ndc256/mask_outbbox', 'mask/hit',
                'a2w/weight', 'a2w/T', 'a2w/alphainv_last', 'a2w/i_start', 'a2w/i_end']
NATIVE_TOL = {  # key -> (atol, rtol)
    'rays_dir': (2.5e-7, 0.0),                 # d / |d|: 2 ulp (contraction of dx*dx+dy*dy+dz*dz is the compiler's choice)
    'aabb/pts': (6e-7, 0.0),                   # start + dir * (stepdist * k) with that dir
    'r2a_a/exp': (0.0, 1.3e-7), 'r2a_b/exp': (0.0, 1.3e-7), 'r2a_c/exp': (0.0, 1.3e-7), 'r2a_nonuni/exp': (0.0, 1.3e-7),   # expf: 1 ulp
    'r2a_a/alpha': (2.4e-7, 0.0), 'r2a_b/alpha': (2.4e-7, 0.0), 'r2a_c/alpha': (2.4e-7, 0.0), 'r2a_nonuni/alpha': (2.4e-7, 0.0),
    'r2a_a/grad': (2.4e-7, 1e-6), 'r2a_b/grad': (2.4e-7, 1e-6), 'r2a_c/grad': (2.4e-7, 1e-6), 'r2a_nonuni/grad': (2.4e-7, 1e-6),
    'a2w/grad': (1e-6, 1e-5),                  # reverse scan with a division by (1 - alpha + 1e-10)
}


def load_native_golden():
    out = {}
    for name in ('native_sampler', 'native_mask', 'native_alpha'):
        z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
        out[name] = {k: z[k] for k in z.files}
    return out


def replay_native(impl, G, device):
    """Feed the fixtures' inputs to `impl` (an object with the 13 render_utils_cuda entry points) -> {key: numpy array}."""
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    N_ = lambda v: v.detach().cpu().numpy()
    S, M, A = G['native_sampler'], G['native_mask'], G['native_alpha']
    near, far, sd = float(S['in/near']), float(S['in/far']), float(S['in/stepdist'])
    ro, rd, lo, hi = T(S['in/rays_o']), T(S['in/rays_d']), T(S['in/xyz_min']), T(S['in/xyz_max'])
    r = {}
    t_min, t_max = impl.infer_t_minmax(ro, rd, lo, hi, near, far)
    r['t_min'], r['t_max'] = N_(t_min), N_(t_max)
    r['n_samples'] = N_(impl.infer_n_samples(rd, T(S['t_min']), T(S['t_max']), sd))
    st, di = impl.infer_ray_start_dir(ro, rd, T(S['t_min']))
    r['rays_start'], r['rays_dir'] = N_(st), N_(di)
    for k, v in zip(['pts', 'mask_outbbox', 'ray_id', 'step_id', 'N_steps', 'sp_t_min', 'sp_t_max'],
                    impl.sample_pts_on_rays(ro, rd, lo, hi, near, far, sd)):
        r['aabb/' + k] = N_(v)
    for ns in (37, 256):
        pts, m = impl.sample_ndc_pts_on_rays(T(S['in/ndc_o']), T(S['in/ndc_d']), lo, hi, ns)
        r[f'ndc{ns}/pts'], r[f'ndc{ns}/mask_outbbox'] = N_(pts), N_(m)
    r['mask/hit'] = N_(impl.maskcache_lookup(T(M['in/world']), T(M['in/xyz']), T(M['in/scale']), T(M['in/shift'])))
    d, gb = T(A['in/density']), T(A['in/grad_back'])
    for tag in ('a', 'b', 'c'):
        sh, iv = float(A[f'r2a_{tag}/shift']), float(A[f'r2a_{tag}/interval'])
        e, a = impl.raw2alpha(d, sh, iv)
        r[f'r2a_{tag}/exp'], r[f'r2a_{tag}/alpha'] = N_(e), N_(a)
        r[f'r2a_{tag}/grad'] = N_(impl.raw2alpha_backward(T(A[f'r2a_{tag}/exp']), gb, iv))
    ipp = T(A['in/interval_pp'])
    e, a = impl.raw2alpha_nonuni(d, -2.0, ipp)
    r['r2a_nonuni/exp'], r['r2a_nonuni/alpha'] = N_(e), N_(a)
    r['r2a_nonuni/grad'] = N_(impl.raw2alpha_nonuni_backward(T(A['r2a_nonuni/exp']), gb, ipp))
    al, rid, nr = T(A['in/alpha']), T(A['in/ray_id']), int(A['in/n_rays'])
    for k, v in zip(('weight', 'T', 'alphainv_last', 'i_start', 'i_end'), impl.alpha2weight(al, rid, nr)):
        r['a2w/' + k] = N_(v)
    r['a2w/grad'] = N_(impl.alpha2weight_backward(al, T(A['a2w/weight']), T(A['a2w/T']), T(A['a2w/alphainv_last']), T(A['a2w/i_start']),
                                                  T(A['a2w/i_end']), nr, T(A['in/grad_weights']), T(A['in/grad_last'])))
    return r


def native_reference_value(G, key):
    if key == 'mask/hit':
        return G['native_mask']['hit']
    for name in ('native_sampler', 'native_alpha'):
        if key in G[name]:
            return G[name][key]
    raise KeyError(key)


def check_native(got, G, what):
    """Assert the replayed outputs against the reference-made values with the tolerances of NATIVE_EXACT / NATIVE_TOL."""
    assert set(got) == set(NATIVE_EXACT) | set(NATIVE_TOL), sorted(set(got) ^ (set(NATIVE_EXACT) | set(NATIVE_TOL)))
    for key in NATIVE_EXACT:
        want = native_reference_value(G, key)
        assert got[key].shape == want.shape, (what, key, got[key].shape, want.shape)
        assert np.array_equal(got[key], want, equal_nan=True), (what, key, 'must be bit-exact', int((got[key] != want).sum()))
    for key, (atol, rtol) in NATIVE_TOL.items():
        want = native_reference_value(G, key)
        g = got[key]
        assert g.shape == want.shape, (what, key)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(g), fin) and np.array_equal(g[~fin], want[~fin], equal_nan=True), (what, key, 'non-finite pattern')
        err = np.abs(g[fin].astype(np.float64) - want[fin].astype(np.float64))
        bound = atol + rtol * np.abs(want[fin].astype(np.float64))
        assert (err <= bound).all(), (what, key, float(err.max()), float((err - bound).max()))


def sft_layer_torch(l, t, c):
    """SFTLayer.forward (lib/sr_esrnet.py:120-123) with PyTorch ops on the module's parameters: test-side checker."""
    import torch.nn.functional as F
    scale = l.SFT_scale_conv1(F.leaky_relu(l.SFT_scale_conv0(c), 0.2))
    shift = l.SFT_shift_conv1(F.leaky_relu(l.SFT_shift_conv0(c), 0.2))
    return t * (scale + 1) + shift


def rdb_torch(b, t, c):
    """ResidualDenseBlock_SFT.forward (lib/sr_esrnet.py:149-158) with PyTorch ops: test-side checker."""
    import torch.nn.functional as F
    lr = lambda v: F.leaky_relu(v, 0.2)
    xc0 = sft_layer_torch(b.sft0, t, c)
    x1 = lr(b.conv1(xc0))
    x2 = lr(b.conv2(torch.cat((xc0, x1), 1)))
    x3 = lr(b.conv3(torch.cat((xc0, x1, x2), 1)))
    x4 = lr(b.conv4(torch.cat((xc0, x1, x2, x3), 1)))
    xc1 = sft_layer_torch(b.sft1, x4, c)
    return b.conv5(torch.cat((xc0, x1, x2, x3, xc1), 1)) * 0.2 + t


def sftnet_forward_torch(net, x, cond):
    """The reference's op sequence (lib/sr_esrnet.py:112-182,446-465, fea=None) on the module's nn.Conv2d PARAMETERS with PyTorch-ROCm
    ops: a CHECKER for the HIP graphs.  The product has no PyTorch path: SFTNet.forward raises where the HIP kernels do not apply and the
    sub-blocks (SFTLayer, ResidualDenseBlock_SFT, RRDB_SFT) are parameter containers whose forward raises -- so the block bodies live here."""
    import torch.nn.functional as F
    sft, rdb = sft_layer_torch, rdb_torch
    feat = net.conv_first(x)
    c = net.CondNet(cond)
    body = feat
    for rr in net.body:                                                       # lib/sr_esrnet.py:176-182
        out = rdb(rr.rdb3, rdb(rr.rdb2, rdb(rr.rdb1, body, c), c), c)
        body = sft(rr.sft0, out, c) * 0.2 + body
    body_feat = net.conv_body(sft(net.sftbody, body, c)) + feat
    if net.scale > 1:
        body_feat = net.lrelu(net.conv_up1(F.interpolate(body_feat, scale_factor=2, mode='nearest')))
        if net.scale == 4:
            body_feat = net.lrelu(net.conv_up2(F.interpolate(body_feat, scale_factor=2, mode='nearest')))
    return net.conv_last(net.lrelu(net.conv_hr(body_feat)))


# ---------------------------------------------------------------------------------------------------------------------
# Pre-split activations ("p16", include/k4nerf.h): the format restated on the CPU, for the tests of its producers / consumers.
# ---------------------------------------------------------------------------------------------------------------------
def to_p16(x, E):
    """fp32 [..., C] (C % 16 == 0) -> int32 [..., C]: per 16-channel chunk the four units [hi 0-7][hi 8-15][lo 0-7][lo 8-15] of
    hi = RNE_fp16(x 2^E), lo = RNE_fp16(x 2^E - hi)."""
    xs = torch.ldexp(x.float().cpu(), torch.tensor(E))
    hi = xs.to(torch.float16)
    lo = (xs - hi.float()).to(torch.float16)
    lead, C = x.shape[:-1], x.shape[-1]
    units = torch.stack([hi.reshape(*lead, C // 16, 16), lo.reshape(*lead, C // 16, 16)], -2)        # [..., chunk, term, 16]
    return units.contiguous().view(torch.int16).reshape(*lead, C // 16, 32).view(torch.int32).reshape(*lead, C)


def from_p16(p, E):
    """Inverse of to_p16: (hi + lo) 2^-E as fp64."""
    lead, C = p.shape[:-1], p.shape[-1]
    h = p.cpu().contiguous().view(torch.int16).reshape(*lead, C // 16, 2, 16).view(torch.float16).double()
    return torch.ldexp(h[..., 0, :] + h[..., 1, :], torch.tensor(-E)).reshape(*lead, C)
