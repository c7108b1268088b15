"""Generate ``tests/golden/native_*.npz`` / ``optim_ref.npz`` by running the REFERENCE's compiled kernels on a GPU.
TEST INFRASTRUCTURE ONLY.

    python oracle/gen_native_golden.py [--out gpurun_out/native_golden]        (on the MI355X box, via gpurun)

The libraries are ``oracle/_ref/*_ref.so`` = the reference's own ``lib/cuda/*.cu|.cpp`` built by ``oracle/build_ref.py`` in
the build container (they travel with the repo snapshot; /root/reference does not exist on the GPU box).  Inputs are
seeded numpy arrays built here, stored in the fixture next to the outputs, so the CPU tests replay them without a GPU:

  native_sampler.npz   infer_t_minmax / infer_n_samples / infer_ray_start_dir / sample_pts_on_rays (ray-AABB: zero direction
                       components, rays that miss the box, origins inside) and sample_ndc_pts_on_rays
                       (render_utils_kernel.cu:12-293)
  native_mask.npz      maskcache_lookup incl. exact .5 ties, negative halves and out-of-range indices (:374-424)
  native_alpha.npz     raw2alpha / raw2alpha_nonuni (+backward) incl. +-inf, saturating and tiny densities (:431-574);
                       alpha2weight fwd/bwd incl. rays without points and rays that hit the T < 1e-3 stop (:577-707)
  optim_ref.npz        adam_upd / masked_adam_upd / adam_upd_with_perlr (adam_upd_kernel.cu) and total_variation_add_grad
                       dense + sparse (total_variation_kernel.cu)
  native_march_*.npz   the march_* fixtures' inputs re-evaluated by oracle/marcher.py with the REFERENCE's compiled kernels in
                       place of oracle/native_cpu.py (per-ray outputs only): end-to-end values in which every native step is
                       reference-made.

With ``--check`` (default on) the script also prints how oracle/native_cpu.py, oracle/optim.py and the product's staged
gfx950 kernels compare with the vectors it has just produced.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import build_ref, native_cpu, optim as ooptim, marcher   # noqa: E402

DEV = 'cuda'


def _np(v):
    return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sampler_inputs():
    rs = np.random.RandomState(1234)
    xyz_min = np.array([-1.3, -1.1, -1.0], np.float32)
    xyz_max = np.array([1.3, 1.1, 1.0], np.float32)
    n = 301                                                   # ragged against 64 and 256
    o = (rs.uniform(-2.0, 2.0, [n, 3])).astype(np.float32)
    d = rs.normal(0, 1, [n, 3]).astype(np.float32)
    o[:40] = rs.uniform(-0.9, 0.9, [40, 3]).astype(np.float32)            # origins inside the box
    d[40:60, 0] = 0.0                                                     # zero components -> 1e-6 (.cu:24-26)
    d[60:70, 1] = 0.0
    d[70:80, 2] = 0.0
    d[80:84, :2] = 0.0                                                    # axis-aligned rays
    o[84:100] = np.array([3.0, 3.0, 3.0], np.float32)                      # pointing away: miss
    d[84:100] = np.abs(d[84:100])
    d[100:120] *= 1e-3                                                    # short direction vectors
    d[120:140] *= 50.0
    # NDC-like rays: origin on the near plane z=-1 .. travelling to z=+1
    on = np.stack([rs.uniform(-1.5, 1.5, n), rs.uniform(-1.3, 1.3, n), -np.ones(n)], -1).astype(np.float32)
    dn = np.stack([rs.uniform(-0.4, 0.4, n), rs.uniform(-0.4, 0.4, n), 2 * np.ones(n)], -1).astype(np.float32)
    return dict(xyz_min=xyz_min, xyz_max=xyz_max, rays_o=o, rays_d=d, ndc_o=on, ndc_d=dn,
                near=np.float32(0.2), far=np.float32(1e9), stepdist=np.float32(0.0137))


def gen_sampler(ref):
    I = sampler_inputs()
    g = {k: _t(v).to(DEV) for k, v in I.items() if isinstance(v, np.ndarray) and v.ndim > 0}
    near, far, sd = float(I['near']), float(I['far']), float(I['stepdist'])
    out = {('in/' + k): v for k, v in I.items()}
    t_min, t_max = ref.infer_t_minmax(g['rays_o'], g['rays_d'], g['xyz_min'], g['xyz_max'], near, far)
    out['t_min'], out['t_max'] = _np(t_min), _np(t_max)
    out['n_samples'] = _np(ref.infer_n_samples(g['rays_d'], t_min, t_max, sd))
    start, rdir = ref.infer_ray_start_dir(g['rays_o'], g['rays_d'], t_min)
    out['rays_start'], out['rays_dir'] = _np(start), _np(rdir)
    names = ['pts', 'mask_outbbox', 'ray_id', 'step_id', 'N_steps', 'sp_t_min', 'sp_t_max']
    for k, v in zip(names, ref.sample_pts_on_rays(g['rays_o'], g['rays_d'], g['xyz_min'], g['xyz_max'], near, far, sd)):
        out['aabb/' + k] = _np(v)
    for ns in (37, 256):
        pts, m = ref.sample_ndc_pts_on_rays(g['ndc_o'], g['ndc_d'], g['xyz_min'], g['xyz_max'], ns)
        out[f'ndc{ns}/pts'], out[f'ndc{ns}/mask_outbbox'] = _np(pts), _np(m)
    return out


def mask_inputs():
    rs = np.random.RandomState(77)
    world = rs.rand(13, 11, 17) < 0.4
    scale = np.array([2.0, 4.0, 0.5], np.float32)
    shift = np.array([0.5, -1.0, 0.25], np.float32)
    n = 4096
    # coordinates on a 1/16 lattice: xyz*scale+shift lands on exact multiples of 1/32 incl. every k + 0.5 tie
    xyz = np.stack([rs.randint(-24, 140, n) / 16.0, rs.randint(-8, 60, n) / 16.0, rs.randint(-40, 600, n) / 16.0], -1).astype(np.float32)
    # explicit ties / edges (index = x*scale+shift): -0.5 -> -1 (out), 0.5 -> 1, 12.5 -> 13 (out for dim 13), 11.5 -> 12
    xyz[:8, 0] = np.array([-0.5, 0.0, 6.0, 5.5, -0.25, 5.75, 6.25, 0.25], np.float32)
    xyz[8:16, 1] = np.array([0.125, 0.375, 2.625, 2.875, 0.25, 2.75, 0.0, 3.0], np.float32)
    xyz_rand = rs.uniform(-1, 35, [n, 3]).astype(np.float32)
    return dict(world=world, scale=scale, shift=shift, xyz=np.concatenate([xyz, xyz_rand], 0))


def gen_mask(ref):
    I = mask_inputs()
    out = {('in/' + k): v for k, v in I.items()}
    out['hit'] = _np(ref.maskcache_lookup(_t(I['world']).to(DEV), _t(I['xyz']).to(DEV), _t(I['scale']).to(DEV), _t(I['shift']).to(DEV)))
    return out


def alpha_inputs():
    rs = np.random.RandomState(99)
    n = 5000
    dens = rs.normal(-2.0, 4.0, n).astype(np.float32)
    dens[:12] = np.array([np.inf, -np.inf, 100.0, -100.0, 88.0, 89.0, -87.0, 0.0, 1e-8, -1e-8, 16.0, -16.0], np.float32)
    interval_pp = rs.uniform(0.1, 2.0, n).astype(np.float32)
    grad_back = rs.normal(0, 1, n).astype(np.float32)
    # alpha2weight: 200 rays; some empty, some opaque early (stop), some long and faint
    n_rays = 200
    counts = rs.randint(0, 60, n_rays)
    counts[[3, 17, 18, 199]] = 0
    counts[5] = 256
    ray_id = np.repeat(np.arange(n_rays), counts).astype(np.int64)
    m = len(ray_id)
    alpha = (rs.beta(0.4, 3.0, m)).astype(np.float32)
    sel = np.isin(ray_id, [7, 8, 9, 40])
    alpha[sel] = rs.uniform(0.5, 0.999, sel.sum()).astype(np.float32)          # T crosses 1e-3 quickly
    alpha[ray_id == 5] = np.float32(0.03)                                     # crosses the stop near the end of 256 samples
    alpha[ray_id == 11] = 0.0
    first = np.searchsorted(ray_id, 12)
    if first < m:
        alpha[first] = 1.0                                                     # fully opaque first sample: T becomes 0
    gw = rs.normal(0, 1, m).astype(np.float32)
    gl = rs.normal(0, 1, n_rays).astype(np.float32)
    return dict(density=dens, interval_pp=interval_pp, grad_back=grad_back, alpha=alpha, ray_id=ray_id,
                n_rays=np.int64(n_rays), grad_weights=gw, grad_last=gl)


def gen_alpha(ref):
    I = alpha_inputs()
    out = {('in/' + k): v for k, v in I.items()}
    d = _t(I['density']).to(DEV)
    gb = _t(I['grad_back']).to(DEV)
    ipp = _t(I['interval_pp']).to(DEV)
    for tag, shift, interval in (('a', -4.0, 0.5), ('b', 0.0, 1.0), ('c', -13.8, 0.25)):
        e, a = ref.raw2alpha(d, shift, interval)
        out[f'r2a_{tag}/shift'], out[f'r2a_{tag}/interval'] = np.float32(shift), np.float32(interval)
        out[f'r2a_{tag}/exp'], out[f'r2a_{tag}/alpha'] = _np(e), _np(a)
        out[f'r2a_{tag}/grad'] = _np(ref.raw2alpha_backward(e, gb, interval))
    e, a = ref.raw2alpha_nonuni(d, -2.0, ipp)
    out['r2a_nonuni/exp'], out['r2a_nonuni/alpha'] = _np(e), _np(a)
    out['r2a_nonuni/grad'] = _np(ref.raw2alpha_nonuni_backward(e, gb, ipp))
    alpha, rid = _t(I['alpha']).to(DEV), _t(I['ray_id']).to(DEV)
    nr = int(I['n_rays'])
    w, T, ainv, i0, i1 = ref.alpha2weight(alpha, rid, nr)
    for k, v in zip(('weight', 'T', 'alphainv_last', 'i_start', 'i_end'), (w, T, ainv, i0, i1)):
        out['a2w/' + k] = _np(v)
    out['a2w/grad'] = _np(ref.alpha2weight_backward(alpha, w, T, ainv, i0, i1, nr, _t(I['grad_weights']).to(DEV), _t(I['grad_last']).to(DEV)))
    return out


def optim_inputs():
    rs = np.random.RandomState(5)
    shape = (1, 3, 9, 10, 33)
    p = rs.normal(0, 1, shape).astype(np.float32)
    g = rs.normal(0, 1, shape).astype(np.float32)
    g[rs.rand(*shape) < 0.6] = 0.0
    m = (rs.normal(0, 0.1, shape)).astype(np.float32)
    v = (rs.uniform(0, 0.01, shape)).astype(np.float32)
    perlr = rs.uniform(0, 1, shape).astype(np.float32)
    return dict(param=p, grad=g, exp_avg=m, exp_avg_sq=v, perlr=perlr)


def gen_optim(adam, tv):
    I = optim_inputs()
    out = {('in/' + k): v for k, v in I.items()}
    hyp = dict(beta1=0.9, beta2=0.99, lr=0.1, eps=1e-8)
    out['hyper_json'] = np.array(json.dumps(hyp))
    for name in ('adam_upd', 'masked_adam_upd', 'adam_upd_with_perlr'):
        for step in (1, 7):
            p, g, m, v = [_t(I[k]).to(DEV).clone() for k in ('param', 'grad', 'exp_avg', 'exp_avg_sq')]
            if name == 'adam_upd_with_perlr':
                adam.adam_upd_with_perlr(p, g, m, v, _t(I['perlr']).to(DEV), step, hyp['beta1'], hyp['beta2'], hyp['lr'], hyp['eps'])
            else:
                getattr(adam, name)(p, g, m, v, step, hyp['beta1'], hyp['beta2'], hyp['lr'], hyp['eps'])
            out[f'{name}/{step}/param'], out[f'{name}/{step}/exp_avg'], out[f'{name}/{step}/exp_avg_sq'] = _np(p), _np(m), _np(v)
    for dense in (True, False):
        p, g = _t(I['param']).to(DEV).clone(), _t(I['grad']).to(DEV).clone()
        tv.total_variation_add_grad(p, g, 0.3, 0.2, 0.7, dense)
        out[f'tv/{"dense" if dense else "sparse"}/grad'] = _np(g)
    return out


class _RefNativeOnGpu:
    """oracle.native_cpu's interface served by the reference's compiled kernels (CPU tensors in / out)."""

    def __init__(self, ref):
        self.ref = ref

    def __getattr__(self, name):
        fn = getattr(self.ref, name)

        def call(*args):
            a = [x.to(DEV).contiguous() if torch.is_tensor(x) else x for x in args]
            r = fn(*a)
            torch.cuda.synchronize()
            if torch.is_tensor(r):
                return r.cpu()
            return [x.cpu() for x in r]
        return call


def gen_march(ref, golden_dir):
    """march_* fixtures re-evaluated with reference-made native steps."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    outs = {}
    saved = marcher.nat
    marcher.nat = _RefNativeOnGpu(ref)
    try:
        for f in sorted(os.listdir(golden_dir)):
            if not (f.startswith('march_') and f.endswith('.npz')):
                continue
            g = helpers.load_march_golden(f[:-4])
            r = g['rays']
            o = marcher.forward(g['model_class'], g['model_kwargs'], g['model_state_dict'], r['rays_o'], r['rays_d'], r['viewdirs'],
                                **g['render_kwargs'])
            res = {k: _np(v) for k, v in o.items() if torch.is_tensor(v)}
            res['source'] = np.array(f)
            diffs = {k: float(np.abs(res[k].astype(np.float64) - _np(g['out'][k]).astype(np.float64)).max()) if res[k].shape == tuple(g['out'][k].shape) else 'shape'
                     for k in res if k in g['out'] and res[k].dtype != np.dtype('<U1') and k != 'source'}
            print(f'[march] {f}: reference-native vs committed golden (native_cpu underneath): {diffs}')
            outs['native_' + f] = res
    finally:
        marcher.nat = saved
    return outs


def _maxdiff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return f'SHAPE {a.shape} vs {b.shape}'
    if a.dtype == bool or np.issubdtype(a.dtype, np.integer):
        return int((a != b).sum())
    fin = np.isfinite(a) & np.isfinite(b)
    same_nonfinite = np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~fin & ~np.isnan(a)], b[~fin & ~np.isnan(b)])
    d = np.abs(a[fin].astype(np.float64) - b[fin].astype(np.float64))
    ulp = np.spacing(np.maximum(np.abs(a[fin]), np.abs(b[fin])).astype(np.float32)).astype(np.float64)
    return {'max_abs': float(d.max()) if d.size else 0.0, 'max_ulp': float((d / ulp).max()) if d.size else 0.0,
            'nonfinite_equal': bool(same_nonfinite), 'n_diff': int((d > 0).sum())}


def check_against(outs, impl, tag, device):
    """Replay the fixtures through `impl` (oracle.native_cpu on CPU, or the product's render_utils_cuda shim on the GPU)."""
    T = lambda a: _t(np.asarray(a)).to(device)
    S, M, A = outs['native_sampler'], outs['native_mask'], outs['native_alpha']
    near, far, sd = float(S['in/near']), float(S['in/far']), float(S['in/stepdist'])
    ro, rd, lo, hi = T(S['in/rays_o']), T(S['in/rays_d']), T(S['in/xyz_min']), T(S['in/xyz_max'])
    rep = {}
    t_min, t_max = impl.infer_t_minmax(ro, rd, lo, hi, near, far)
    rep['t_min'], rep['t_max'] = _maxdiff(_np(t_min), S['t_min']), _maxdiff(_np(t_max), S['t_max'])
    rep['n_samples'] = _maxdiff(_np(impl.infer_n_samples(rd, T(S['t_min']), T(S['t_max']), sd)), S['n_samples'])
    st, di = impl.infer_ray_start_dir(ro, rd, T(S['t_min']))
    rep['rays_start'], rep['rays_dir'] = _maxdiff(_np(st), S['rays_start']), _maxdiff(_np(di), S['rays_dir'])
    r = impl.sample_pts_on_rays(ro, rd, lo, hi, near, far, sd)
    for k, v in zip(['pts', 'mask_outbbox', 'ray_id', 'step_id', 'N_steps', 'sp_t_min', 'sp_t_max'], r):
        rep['aabb/' + k] = _maxdiff(_np(v), S['aabb/' + k])
    for ns in (37, 256):
        pts, m = impl.sample_ndc_pts_on_rays(T(S['in/ndc_o']), T(S['in/ndc_d']), lo, hi, ns)
        rep[f'ndc{ns}/pts'], rep[f'ndc{ns}/mask'] = _maxdiff(_np(pts), S[f'ndc{ns}/pts']), _maxdiff(_np(m), S[f'ndc{ns}/mask_outbbox'])
    rep['mask/hit'] = _maxdiff(_np(impl.maskcache_lookup(T(M['in/world']), T(M['in/xyz']), T(M['in/scale']), T(M['in/shift']))), M['hit'])
    d, gb = T(A['in/density']), T(A['in/grad_back'])
    for tag2 in ('a', 'b', 'c'):
        sh, iv = float(A[f'r2a_{tag2}/shift']), float(A[f'r2a_{tag2}/interval'])
        e, a = impl.raw2alpha(d, sh, iv)
        rep[f'r2a_{tag2}/exp'], rep[f'r2a_{tag2}/alpha'] = _maxdiff(_np(e), A[f'r2a_{tag2}/exp']), _maxdiff(_np(a), A[f'r2a_{tag2}/alpha'])
        rep[f'r2a_{tag2}/grad'] = _maxdiff(_np(impl.raw2alpha_backward(T(A[f'r2a_{tag2}/exp']), gb, iv)), A[f'r2a_{tag2}/grad'])
    e, a = impl.raw2alpha_nonuni(d, -2.0, T(A['in/interval_pp']))
    rep['r2a_nonuni/alpha'] = _maxdiff(_np(a), A['r2a_nonuni/alpha'])
    al, rid, nr = T(A['in/alpha']), T(A['in/ray_id']), int(A['in/n_rays'])
    w, Tt, ainv, i0, i1 = impl.alpha2weight(al, rid, nr)
    for k, v in zip(('weight', 'T', 'alphainv_last', 'i_start', 'i_end'), (w, Tt, ainv, i0, i1)):
        rep['a2w/' + k] = _maxdiff(_np(v), A['a2w/' + k])
    g = impl.alpha2weight_backward(al, T(A['a2w/weight']), T(A['a2w/T']), T(A['a2w/alphainv_last']), T(A['a2w/i_start']), T(A['a2w/i_end']),
                                   nr, T(A['in/grad_weights']), T(A['in/grad_last']))
    rep['a2w/grad'] = _maxdiff(_np(g), A['a2w/grad'])
    print(f'==== {tag} vs reference-compiled kernels ====')
    for k, v in rep.items():
        print(f'  {k:24s} {v}')
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'native_golden'))
    ap.add_argument('--no-check', action='store_true')
    args = ap.parse_args()
    assert torch.cuda.is_available(), 'needs a GPU: the reference extensions reject CPU tensors (render_utils.cpp:46-48)'
    ref = build_ref.load('render_utils_cuda_ref')
    adam = build_ref.load('adam_upd_cuda_ref')
    tv = build_ref.load('total_variation_cuda_ref')
    os.makedirs(args.out, exist_ok=True)
    outs = {'native_sampler': gen_sampler(ref), 'native_mask': gen_mask(ref), 'native_alpha': gen_alpha(ref),
            'optim_ref': gen_optim(adam, tv)}
    for name, arrs in outs.items():
        np.savez_compressed(os.path.join(args.out, name + '.npz'), **arrs)
        print('wrote', name, len(arrs), 'arrays')
    march = gen_march(ref, os.path.join(ROOT, 'tests', 'golden'))
    for name, arrs in march.items():
        np.savez_compressed(os.path.join(args.out, name), **arrs)
        print('wrote', name)
    if not args.no_check:
        check_against(outs, native_cpu, 'oracle/native_cpu.py (CPU)', 'cpu')
        import nerf4k_amd  # noqa: F401
        from nerf4k_amd.lib import render_utils_cuda as shim
        check_against(outs, shim, 'product staged gfx950 kernels (k4_staged.hip)', DEV)
        O = outs['optim_ref']
        hyp = json.loads(str(O['hyper_json']))
        print('==== oracle/optim.py vs reference-compiled kernels ====')
        for name in ('adam_upd', 'masked_adam_upd', 'adam_upd_with_perlr'):
            for step in (1, 7):
                p, m, v = ooptim.adam_upd(O['in/param'], O['in/grad'], O['in/exp_avg'], O['in/exp_avg_sq'], step, hyp['beta1'], hyp['beta2'],
                                          hyp['lr'], hyp['eps'], perlr=O['in/perlr'] if name.endswith('perlr') else None,
                                          masked=name == 'masked_adam_upd')
                print(f'  {name}/{step}: param {_maxdiff(p, O[f"{name}/{step}/param"])} exp_avg {_maxdiff(m, O[f"{name}/{step}/exp_avg"])} '
                      f'exp_avg_sq {_maxdiff(v, O[f"{name}/{step}/exp_avg_sq"])}')
        for dense in (True, False):
            g = ooptim.total_variation_add_grad(O['in/param'], O['in/grad'], 0.3, 0.2, 0.7, dense)
            key = f'tv/{"dense" if dense else "sparse"}/grad'
            print(f'  {key}: {_maxdiff(g, O[key])}')


if __name__ == '__main__':
    main()
