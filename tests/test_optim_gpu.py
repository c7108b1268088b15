"""GPU parity of the training-step streaming kernels (csrc/k4_opt.hip) against oracle/optim.py.

Tolerance: the kernels follow the reference's fp32 op order; the oracle emulates FMA through float64, which can differ
from a true FMA by one rounding, and a parameter update p - num/den adds an ulp of p.  rtol 2e-6 / atol 1e-7 on
moments, atol 2e-7 on parameters of magnitude ~1 (written at each check)."""
import os
import numpy as np
import pytest
import torch

import nerf4k_amd  # noqa: F401
from nerf4k_amd.lib import masked_adam as MA
from nerf4k_amd.lib import grid as G
from nerf4k_amd.lib.masked_adam import MaskedAdam
from oracle import optim as O

pytestmark = pytest.mark.gpu


def _case(n, seed, zero_frac=0.0):
    rng = np.random.default_rng(seed)
    p = rng.standard_normal(n).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    if zero_frac:
        g[rng.random(n) < zero_frac] = 0
    m = (rng.standard_normal(n) * 0.1).astype(np.float32)
    v = (rng.random(n) * 0.1).astype(np.float32)
    return p, g, m, v


def _close(got, want, rtol, atol):
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=rtol, atol=atol)


@pytest.mark.parametrize('n', [0, 1, 3, 4, 255, 1024, 100003])
@pytest.mark.parametrize('variant', ['plain', 'masked', 'perlr'])
def test_adam_kernels_match_oracle(n, variant):
    p, g, m, v = _case(n, 10 + n % 7, zero_frac=0.5 if variant == 'masked' else 0.0)
    perlr = np.random.default_rng(5).random(n).astype(np.float32) if variant == 'perlr' else None
    tp, tg, tm, tv = (torch.from_numpy(a.copy()).cuda() for a in (p, g, m, v))
    step, b1, b2, lr, eps = 7, 0.9, 0.99, 0.1, 1e-8
    if variant == 'plain':
        MA.adam_upd(tp, tg, tm, tv, step, b1, b2, lr, eps)
    elif variant == 'masked':
        MA.masked_adam_upd(tp, tg, tm, tv, step, b1, b2, lr, eps)
    else:
        MA.adam_upd_with_perlr(tp, tg, tm, tv, torch.from_numpy(perlr).cuda(), step, b1, b2, lr, eps)
    torch.cuda.synchronize()
    wp, wm, wv = O.adam_upd(p, g, m, v, step, b1, b2, lr, eps, perlr=perlr, masked=(variant == 'masked'))
    _close(tm, wm, 2e-6, 1e-7)
    _close(tv, wv, 2e-6, 1e-7)
    _close(tp, wp, 2e-6, 2e-7)
    if variant == 'masked':                                 # untouched voxels are BIT-identical
        z = g == 0
        assert np.array_equal(tp.cpu().numpy()[z], p[z]) and np.array_equal(tm.cpu().numpy()[z], m[z]) \
            and np.array_equal(tv.cpu().numpy()[z], v[z])
    assert torch.equal(tg.cpu(), torch.from_numpy(g))       # gradient is read-only


def test_adam_misaligned_views_take_the_scalar_path_with_identical_results():
    n = 4099
    p, g, m, v = _case(n + 1, 3, zero_frac=0.3)
    big = [torch.from_numpy(a.copy()).cuda() for a in (p, g, m, v)]
    off = [t[1:] for t in big]                              # 4-byte offset: not 16-byte aligned
    MA.masked_adam_upd(*off, 2, 0.9, 0.99, 0.05, 1e-8)
    al = [torch.from_numpy(a[1:].copy()).cuda() for a in (p, g, m, v)]
    MA.masked_adam_upd(*al, 2, 0.9, 0.99, 0.05, 1e-8)
    torch.cuda.synchronize()
    for a, b in zip(off, al):
        assert torch.equal(a, b)                            # vector and scalar kernels are bit-identical
    assert float(big[0][0]) == float(p[0])                  # element before the view untouched


def test_adam_rejects_bad_arguments():
    t = torch.zeros(8, device='cuda')
    with pytest.raises(ValueError):
        MA.adam_upd(t, t.double(), t, t, 1, 0.9, 0.99, 0.1, 1e-8)
    with pytest.raises(ValueError):
        MA.adam_upd(t, t[:4], t, t, 1, 0.9, 0.99, 0.1, 1e-8)
    with pytest.raises(Exception):
        MA.adam_upd(t, t.clone(), t.clone(), t.clone(), 0, 0.9, 0.99, 0.1, 1e-8)   # step must be >= 1


def test_masked_adam_optimizer_trajectory_matches_oracle():
    rng = np.random.default_rng(0)
    shape_d, shape_k = (1, 1, 6, 5, 8), (1, 4, 6, 5, 8)
    d0, k0 = rng.standard_normal(shape_d).astype(np.float32), rng.standard_normal(shape_k).astype(np.float32)
    dens = torch.nn.Parameter(torch.from_numpy(d0.copy()).cuda())
    k0p = torch.nn.Parameter(torch.from_numpy(k0.copy()).cuda())
    opt = MaskedAdam([{'params': [dens], 'lr': 0.1, 'skip_zero_grad': True, 'kname': 'density'},
                      {'params': [k0p], 'lr': 0.05, 'skip_zero_grad': False, 'kname': 'k0'}])
    count = rng.integers(0, 9, shape_d).astype(np.float32)
    sd = [d0.copy(), np.zeros_like(d0), np.zeros_like(d0)]
    sk = [k0.copy(), np.zeros_like(k0), np.zeros_like(k0)]
    for step in range(1, 5):
        if step == 3:                                       # per-voxel lr applies to tensors of the count's shape
            opt.set_pervoxel_lr(torch.from_numpy(count).cuda())
        gd = rng.standard_normal(shape_d).astype(np.float32)
        gd[rng.random(shape_d) < 0.5] = 0
        gk = rng.standard_normal(shape_k).astype(np.float32)
        dens.grad, k0p.grad = torch.from_numpy(gd).cuda(), torch.from_numpy(gk).cuda()
        opt.step()
        if step >= 3:
            sd = list(O.adam_upd(*sd[:1], gd, *sd[1:], step, 0.9, 0.99, 0.1, 1e-8, perlr=count / count.max()))
        else:
            sd = list(O.adam_upd(*sd[:1], gd, *sd[1:], step, 0.9, 0.99, 0.1, 1e-8, masked=True))
        sk = list(O.adam_upd(*sk[:1], gk, *sk[1:], step, 0.9, 0.99, 0.05, 1e-8))
        _close(dens.detach(), sd[0], 5e-6, 5e-7)
        _close(k0p.detach(), sk[0], 5e-6, 5e-7)
    assert opt.state[dens]['step'] == 4 and set(opt.state[dens]) == {'step', 'exp_avg', 'exp_avg_sq'}


@pytest.mark.parametrize('shape', [(1, 1, 5, 6, 8), (1, 3, 4, 7, 5), (1, 2, 1, 3, 4), (1, 2, 9, 1, 12), (1, 1, 1, 1, 1),
                                   (1, 12, 10, 11, 256), (1, 3, 8, 5, 20), (1, 2, 9, 13, 128), (1, 2, 17, 9, 260)])
@pytest.mark.parametrize('dense', [True, False])
def test_total_variation_matches_oracle(shape, dense):
    rng = np.random.default_rng(sum(shape))
    p = (rng.standard_normal(shape) * 1.5).astype(np.float32)
    g = rng.standard_normal(shape).astype(np.float32)
    if not dense:
        g[rng.random(shape) < 0.6] = 0
    tp, tg = torch.from_numpy(p).cuda(), torch.from_numpy(g.copy()).cuda()
    G.total_variation_add_grad(tp, tg, 0.7, 1.3, 2.1, dense)
    torch.cuda.synchronize()
    want = O.total_variation_add_grad(p, g, 0.7, 1.3, 2.1, dense)
    _close(tg, want, 2e-6, 5e-7)
    if not dense:
        z = g == 0
        assert np.array_equal(tg.cpu().numpy()[z], g[z])
    assert torch.equal(tp.cpu(), torch.from_numpy(p))


@pytest.mark.parametrize('shape', [(1, 2, 9, 13, 128), (1, 3, 4, 7, 5), (1, 1, 8, 300, 260)])
def test_written_dense_tv_term_is_the_added_one(shape):
    """dense_mode 2 ('write': the term computed ahead of the backward pass, lib/grid.py total_variation_seed_grad) must produce exactly what
    adding the term to zeros produces, whatever the buffer held before."""
    g = torch.Generator().manual_seed(sum(shape))
    p = (torch.randn(shape, generator=g) * 1.5).cuda()
    w = torch.full(shape, 7.0, device='cuda')
    G.total_variation_add_grad(p, w, 0.7, 1.3, 2.1, 'write')
    z = torch.zeros(shape, device='cuda')
    G.total_variation_add_grad(p, z, 0.7, 1.3, 2.1, True)
    assert torch.equal(w, z)


def test_dense_grid_and_model_tv_entry_points():
    from nerf4k_amd.lib.grid import DenseGrid
    gr = DenseGrid(2, [6, 5, 8], [0, 0, 0], [1, 1, 1]).cuda()
    with torch.no_grad():
        gr.grid.copy_(torch.randn_like(gr.grid))
    gr.grid.grad = torch.zeros_like(gr.grid)
    gr.total_variation_add_grad(1.0, 1.0, 2.0, True)
    want = O.total_variation_add_grad(gr.grid.detach().cpu().numpy(), np.zeros(gr.grid.shape, np.float32), 1., 1., 2., True)
    _close(gr.grid.grad, want, 2e-6, 5e-7)
    with pytest.raises(ValueError):
        G.total_variation_add_grad(gr.grid.detach()[0], gr.grid.grad[0], 1, 1, 1, True)


def test_full_size_llff_grid_properties():
    """BASELINE-size density grid (417x353x256): size-independent properties instead of an oracle run.
    (1) masked Adam with an all-zero gradient is the identity; (2) TV of a constant grid adds exactly zero;
    (3) TV is odd: TV(-p) = -TV(p); (4) dense Adam == masked Adam where every grad is non-zero."""
    shape = (1, 1, 417, 353, 256)
    n = int(np.prod(shape))
    gen = torch.Generator(device='cuda').manual_seed(0)
    p = torch.randn(shape, device='cuda', generator=gen)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    p_ref = p.clone()
    MA.masked_adam_upd(p, torch.zeros_like(p), m, v, 1, 0.9, 0.99, 0.1, 1e-8)
    assert torch.equal(p, p_ref) and float(m.abs().max()) == 0 and float(v.abs().max()) == 0
    g = torch.zeros_like(p)
    G.total_variation_add_grad(torch.full_like(p, 0.37), g, 1., 1., 1., True)
    assert float(g.abs().max()) == 0
    ga, gb = torch.zeros_like(p), torch.zeros_like(p)
    G.total_variation_add_grad(p, ga, 0.5, 0.5, 2., True)
    G.total_variation_add_grad(-p, gb, 0.5, 0.5, 2., True)
    assert torch.equal(ga, -gb) and float(ga.abs().max()) > 0
    assert abs(float(ga.double().sum())) < 1e-3 * n ** 0.5          # pair terms cancel: sum of TV gradient ~ 0
    grad = torch.randn(shape, device='cuda', generator=gen)
    grad[grad == 0] = 1
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    MA.adam_upd(p, grad, m, v, 1, 0.9, 0.99, 0.1, 1e-8)
    MA.masked_adam_upd(p2, grad, m2, v2, 1, 0.9, 0.99, 0.1, 1e-8)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)


def test_optimizer_kernels_vs_reference_compiled_kernels():
    """k4_opt.hip against tests/golden/optim_ref.npz = the reference's adam_upd_cuda / total_variation_cuda compiled for gfx950 and
    run on an MI355X (oracle/gen_native_golden.py).  Same compiler family on both sides: atol 5e-7 covers contraction order."""
    import json
    import os
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'optim_ref.npz'))
    hyp = json.loads(str(z['hyper_json']))
    args = (hyp['beta1'], hyp['beta2'], hyp['lr'], hyp['eps'])
    T = lambda k: torch.from_numpy(z[k].copy()).cuda()
    for name in ('adam_upd', 'masked_adam_upd', 'adam_upd_with_perlr'):
        for step in (1, 7):
            p, g, m, v = T('in/param'), T('in/grad'), T('in/exp_avg'), T('in/exp_avg_sq')
            if name == 'adam_upd_with_perlr':
                MA.adam_upd_with_perlr(p, g, m, v, T('in/perlr'), step, *args)
            else:
                getattr(MA, name)(p, g, m, v, step, *args)
            for got, key in ((p, 'param'), (m, 'exp_avg'), (v, 'exp_avg_sq')):
                np.testing.assert_allclose(got.cpu().numpy(), z[f'{name}/{step}/{key}'], rtol=0, atol=5e-7, err_msg=f'{name}/{step}/{key}')
    for dense in (True, False):
        p, g = T('in/param'), T('in/grad')
        G.total_variation_add_grad(p, g, 0.3, 0.2, 0.7, dense)
        np.testing.assert_allclose(g.cpu().numpy(), z[f'tv/{"dense" if dense else "sparse"}/grad'], rtol=0, atol=5e-7)


@pytest.mark.parametrize('masked', [False, True])
def test_multi_tensor_adam_is_bit_identical_to_the_single_tensor_kernels(masked):
    """k4_adam_upd_multi (the decoder's optimizer step: 458 tensors in 8 launches) against k4_adam_upd / k4_masked_adam_upd per tensor:
    ragged sizes incl. empty, one element, chunk boundaries (1023 / 1024 / 1025) and more tensors than one launch holds."""
    from nerf4k_amd.lib import masked_adam as MA
    g = torch.Generator().manual_seed(3)
    sizes = [0, 1, 3, 1023, 1024, 1025, 5000, 64, 2048, 7] * 15
    items_a, items_b = [], []
    for n in sizes:
        p, gr = torch.randn([n], generator=g), torch.randn([n], generator=g)
        if masked and n:
            gr[torch.rand([n], generator=g) < 0.5] = 0
        m, v = torch.randn([n], generator=g) * 0.1, torch.rand([n], generator=g) * 0.1
        items_a.append(tuple(t.clone().cuda() for t in (p, gr, m, v)))
        items_b.append(tuple(t.clone().cuda() for t in (p, gr, m, v)))
    MA.adam_upd_multi(items_a, masked, 7, 0.9, 0.99, 2e-4, 1e-8)
    for p, gr, m, v in items_b:
        (MA.masked_adam_upd if masked else MA.adam_upd)(p, gr, m, v, 7, 0.9, 0.99, 2e-4, 1e-8)
    for a, b in zip(items_a, items_b):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert items_a[3][0]._version > 0                                   # versions bumped like the single-tensor wrappers do


@pytest.mark.parametrize('masked', [False, True])
def test_masked_adam_fast_path_for_small_tensors_is_the_regular_step(masked, monkeypatch):
    """MaskedAdam.step takes a cached plan for a group's small tensors from its second step on (_fast_step: one pointer refresh per tensor):
    five steps with fresh gradient tensors each time against an optimizer whose fast path is disabled -- identical parameters and moments, the
    same step counts; then a missing gradient, a re-allocated moment and load_state_dict each drop back to the regular path and stay identical."""
    g = torch.Generator().manual_seed(11)
    shapes = [(3, 3, 7), (64,), (32, 16, 3, 3), (1,), (17,)] * 12
    init = [torch.randn(sh, generator=g) for sh in shapes]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    oa = MaskedAdam([{'params': pa, 'lr': 1e-3, 'skip_zero_grad': masked}])
    ob = MaskedAdam([{'params': pb, 'lr': 1e-3, 'skip_zero_grad': masked}])
    monkeypatch.setattr(ob, '_fast_step', lambda *a, **k: False)
    took = []
    orig = oa._fast_step
    monkeypatch.setattr(oa, '_fast_step', lambda *a, **k: took.append(orig(*a, **k)) or took[-1])

    def give_grads(skip=None):
        for j, (x, y) in enumerate(zip(pa, pb)):
            gr = torch.randn(x.shape, generator=g)
            if masked:
                gr[torch.rand(x.shape, generator=g) < 0.3] = 0
            x.grad, y.grad = (None, None) if j == skip else (gr.clone().cuda(), gr.clone().cuda())

    def same():
        for x, y in zip(pa, pb):
            assert torch.equal(x, y)
            sa, sb = oa.state.get(x), ob.state.get(y)
            assert (sa is None) == (sb is None)
            if sa:
                assert sa['step'] == sb['step'] and torch.equal(sa['exp_avg'], sb['exp_avg']) and torch.equal(sa['exp_avg_sq'], sb['exp_avg_sq'])

    for it in range(5):
        give_grads()
        oa.step(); ob.step()
        same()
    assert took[0] is False and all(took[1:])                        # first step builds the plan, the next four take it
    v0 = pa[0]._version
    give_grads(skip=4)                                               # a tensor without a gradient: regular path, that tensor's step count stays behind
    oa.step(); ob.step()
    same()
    assert took[-1] is False and pa[0]._version > v0
    give_grads()
    oa.step(); ob.step()                                             # step counts now differ inside the group: still the regular path
    same()
    for opt, ps in ((oa, pa), (ob, pb)):
        opt.state[ps[7]]['exp_avg'] = opt.state[ps[7]]['exp_avg'].clone()     # re-allocated moment
    give_grads()
    oa.step(); ob.step()
    same()
    oa.load_state_dict(oa.state_dict()); ob.load_state_dict(ob.state_dict())
    give_grads()
    oa.step(); ob.step()
    same()


@pytest.mark.parametrize('C', [2, 9, 12])
def test_sparse_masked_adam_from_the_scatter_image_is_the_dense_masked_step(C):
    """k4_masked_adam_upd_sparse_cl (MaskedAdam's masked update of the touched voxels, straight from the channel-last scratch image of the lookups'
    backward) against the dense route -- sweep into a zero gradient, k4_masked_adam_upd (lib/masked_adam.py:58-71 with skip_zero_grad) -- from the
    SAME scratch image: parameters and both moments bit for bit, a channel whose sums are all zero untouched, the image all zero afterwards."""
    from nerf4k_amd import _native as N
    g = torch.Generator().manual_seed(40 + C)
    X, Y, Z = 11, 9, 33
    n = 1500
    pts = (torch.rand([n, 3], generator=g) * 2.2 - 1.1).cuda()
    gout = torch.randn([n, C], generator=g)
    gout[:, 1] = 0                                                        # one channel without gradient anywhere
    gout[::7] = 0                                                         # samples that flag voxels without contributing
    gout = gout.cuda()
    mn, mx = torch.tensor([-1., -1., -1.]).cuda(), torch.tensor([1., 1., 1.]).cuda()
    L = N.lib()
    nb = int(L.k4_grid_sample_3d_backward_workspace_bytes(C, X, Y, Z))
    ws = torch.zeros([nb // 4], dtype=torch.int32, device='cuda')
    N.check(L.k4_grid_sample_3d_backward_cl_scatter(N.f32(gout), C, X, Y, Z, N.f32(pts), N.f32(mn), N.f32(mx), n, N.ptr(ws), N.stream()), 'scatter')
    assert int(ws.count_nonzero()) > 0
    ws2 = ws.clone()
    shape = [1, C, X, Y, Z]
    p0 = torch.randn(shape, generator=g).cuda()
    m0 = (torch.randn(shape, generator=g) * 0.1).cuda()
    v0 = (torch.rand(shape, generator=g) * 0.1).cuda()
    hyper = (7, 0.9, 0.99, 1e-1, 1e-8)
    # dense route
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    grad = torch.zeros(shape, device='cuda')
    N.check(L.k4_grid_sample_3d_backward_cl_sweep(C, X, Y, Z, N.ptr(ws), N.f32(grad), N.stream()), 'sweep')
    assert int(ws.count_nonzero()) == 0 and int(grad.count_nonzero()) > 0
    N.check(L.k4_masked_adam_upd(N.ptr(pa), N.ptr(grad), N.ptr(ma), N.ptr(va), pa.numel(), *hyper, N.stream()), 'masked_adam_upd')
    # in place
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    N.check(L.k4_masked_adam_upd_sparse_cl(N.ptr(pb), N.ptr(mb), N.ptr(vb), N.ptr(ws2), C, X, Y, Z, *hyper, N.stream()), 'sparse')
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert int(ws2.count_nonzero()) == 0
    changed = pb != p0
    assert torch.equal(changed, grad != 0) or int((changed != (grad != 0)).sum()) <= 2        # (an update smaller than half an ulp of p leaves p's bits)
    assert not bool(changed[0, 1].any()) and torch.equal(mb[0, 1], m0[0, 1])
    assert int(L.k4_masked_adam_upd_sparse_cl(N.ptr(pb), N.ptr(mb), N.ptr(vb), N.ptr(ws2), 1, X, Y, Z, *hyper, N.stream())) == 10001
    assert int(L.k4_masked_adam_upd_sparse_cl(N.ptr(pb), N.ptr(mb), N.ptr(vb), N.ptr(ws2), C, X, Y, Z, 0, 0.9, 0.99, 1e-1, 1e-8, N.stream())) == 10001


@pytest.mark.parametrize('C', [3, 12])
def test_split_masked_adam_step_equals_the_one_pass_step(C):
    """The masked step of a grid in two parts -- k4_grid_flag_corners (the voxels a scatter of the lookup's points will touch), k4_masked_adam_upd_unflagged
    (every other voxel, gradient = the dense term written ahead) and, after the scatter, k4_masked_adam_upd_sparse_cl_seeded (the flagged voxels, gradient =
    seed + the scatter's sums) -- against the one-pass route from the SAME scratch image: sweep into the seed, k4_masked_adam_upd.  Parameters and both moments
    bit for bit; the flags equal the scatter's own; scratch image and flags all zero afterwards; seed zeros (skipped elements) on both sides of the flags."""
    from nerf4k_amd import _native as N
    g = torch.Generator().manual_seed(90 + C)
    X, Y, Z = 11, 9, 32                                                   # (X * Y * Z % 4 == 0: the vector form of the first part)
    n = 700
    pts = (torch.rand([n, 3], generator=g) * 2.2 - 1.1).cuda()           # some points outside the grid: corners out of range
    gout = torch.randn([n, C], generator=g)
    gout[::5] = 0                                                         # samples that flag voxels without contributing
    gout = gout.cuda()
    mn, mx = torch.tensor([-1., -1., -1.]).cuda(), torch.tensor([1., 1., 1.]).cuda()
    L = N.lib()
    nvox = X * Y * Z
    shape = [1, C, X, Y, Z]
    seed = torch.randn(shape, generator=g)
    seed[0, :, ::3, ::2] = 0                                              # TV terms that are exactly zero: skipped unless the scatter adds to them
    seed = seed.cuda()
    nb = int(L.k4_grid_sample_3d_backward_workspace_bytes(C, X, Y, Z))
    ws = torch.zeros([nb // 4], dtype=torch.int32, device='cuda')
    flags = torch.zeros([nvox], dtype=torch.uint8, device='cuda')
    N.check(L.k4_grid_flag_corners(X, Y, Z, N.f32(pts), N.f32(mn), N.f32(mx), n, N.ptr(flags), N.stream()), 'k4_grid_flag_corners')
    N.check(L.k4_grid_sample_3d_backward_cl_scatter(N.f32(gout), C, X, Y, Z, N.f32(pts), N.f32(mn), N.f32(mx), n, N.ptr(ws), N.stream()), 'scatter')
    sflags = ws.view(torch.uint8)[nvox * C * 4: nvox * C * 4 + nvox]
    assert torch.equal(sflags, flags) and 0 < int(flags.sum()) < nvox
    ws2 = ws.clone()
    p0 = torch.randn(shape, generator=g).cuda()
    m0 = (torch.randn(shape, generator=g) * 0.1).cuda()
    v0 = (torch.rand(shape, generator=g) * 0.1).cuda()
    hyper = (5, 0.9, 0.99, 1e-1, 1e-8)
    # one pass: the sweep adds the sums to the seed, the masked step reads the result
    pa, ma, va, grad = p0.clone(), m0.clone(), v0.clone(), seed.clone()
    N.check(L.k4_grid_sample_3d_backward_cl_sweep(C, X, Y, Z, N.ptr(ws), N.f32(grad), N.stream()), 'sweep')
    N.check(L.k4_masked_adam_upd(N.ptr(pa), N.ptr(grad), N.ptr(ma), N.ptr(va), pa.numel(), *hyper, N.stream()), 'masked_adam_upd')
    # two parts
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    N.check(L.k4_masked_adam_upd_unflagged(N.ptr(pb), N.f32(seed), N.ptr(mb), N.ptr(vb), C, nvox, N.ptr(flags), *hyper, 5, N.stream()), 'unflagged')
    fl = flags.bool().view(1, 1, X, Y, Z).expand(shape)
    assert torch.equal(pb[fl], p0[fl]) and torch.equal(pb[~fl], pa[~fl])                  # the first part stepped exactly the unflagged voxels
    N.check(L.k4_masked_adam_upd_sparse_cl_seeded(N.ptr(pb), N.ptr(mb), N.ptr(vb), N.ptr(ws2), N.f32(seed), N.ptr(flags), C, X, Y, Z, *hyper, N.stream()), 'seeded')
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert int(ws2.count_nonzero()) == 0 and int(flags.count_nonzero()) == 0
    # the vector form's preconditions
    assert int(L.k4_masked_adam_upd_unflagged(N.ptr(pb), N.f32(seed), N.ptr(mb), N.ptr(vb), C, nvox - 1, N.ptr(flags), *hyper, 0, N.stream())) == N.K4_ERR_UNSUPPORTED
