"""CPU tests pinning oracle/optim.py (the restated Adam / total-variation kernels) against independent PyTorch
formulations, and the host-side MaskedAdam / optimiser-factory logic.  No GPU needed."""
import types

import numpy as np
import pytest
import torch

from oracle import optim as O


def test_adam_oracle_matches_torch_adam_when_eps_is_negligible():
    # reference: p -= lr*sqrt(bc2)/bc1 * m / (sqrt(v)+eps); torch: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2)+eps).
    # Identical for eps -> 0, so pin with eps=1e-30 on gradients well away from 0.
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(4096).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-2, betas=(0.9, 0.99), eps=1e-30)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = (rng.standard_normal(4096) + 3.0).astype(np.float32)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v = O.adam_upd(p, g, m, v, step, 0.9, 0.99, 1e-2, 1e-30)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_masked_adam_oracle_skips_zero_grads_and_perlr_scales():
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(1000).astype(np.float32)
    g = rng.standard_normal(1000).astype(np.float32)
    g[::3] = 0
    m0 = rng.standard_normal(1000).astype(np.float32) * 0.1
    v0 = rng.random(1000).astype(np.float32) * 0.1
    p, m, v = O.adam_upd(p0, g, m0, v0, 3, 0.9, 0.99, 1e-1, 1e-8, masked=True)
    z = g == 0
    assert np.array_equal(p[z], p0[z]) and np.array_equal(m[z], m0[z]) and np.array_equal(v[z], v0[z])
    pd, md, vd = O.adam_upd(p0, g, m0, v0, 3, 0.9, 0.99, 1e-1, 1e-8)
    assert np.array_equal(p[~z], pd[~z]) and np.array_equal(m[~z], md[~z])
    assert not np.array_equal(m[z], md[z])                 # the dense variant decays the moments of zero-grad voxels
    perlr = rng.random(1000).astype(np.float32)
    pl, _, _ = O.adam_upd(p0, g, m0, v0, 3, 0.9, 0.99, 1e-1, 1e-8, perlr=perlr)
    np.testing.assert_allclose(pl - p0, (pd - p0) * perlr, rtol=1e-4, atol=5e-7)   # fp32 cancellation in p - p0


@pytest.mark.parametrize('shape', [(1, 1, 5, 6, 8), (1, 3, 4, 7, 5), (1, 2, 1, 3, 4)])
def test_tv_oracle_is_the_gradient_of_a_huber_neighbour_loss(shape):
    # clamp(p - p_nb, -1, 1) summed over both neighbours of an axis is d/dp of sum_pairs huber(p_a - p_b), delta = 1
    rng = np.random.default_rng(2)
    p0 = (rng.standard_normal(shape) * 1.5).astype(np.float32)
    g0 = rng.standard_normal(shape).astype(np.float32)
    wx, wy, wz = 0.7, 1.3, 2.1
    tp = torch.from_numpy(p0.astype(np.float64)).requires_grad_(True)
    hub = lambda d: torch.nn.functional.smooth_l1_loss(d, torch.zeros_like(d), reduction='sum', beta=1.0)
    loss = (wx / 6) * hub(tp[..., 1:] - tp[..., :-1]) + (wy / 6) * hub(tp[..., 1:, :] - tp[..., :-1, :]) \
        + (wz / 6) * hub(tp[:, :, 1:] - tp[:, :, :-1])
    loss.backward()
    want = g0.astype(np.float64) + tp.grad.numpy()
    got = O.total_variation_add_grad(p0, g0, wx, wy, wz, True)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-6)


def test_tv_oracle_sparse_mode_only_touches_nonzero_grads():
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal((1, 2, 6, 5, 4)).astype(np.float32)
    g0 = rng.standard_normal(p0.shape).astype(np.float32)
    g0[rng.random(p0.shape) < 0.6] = 0
    dense = O.total_variation_add_grad(p0, g0, 1., 1., 2., True)
    sparse = O.total_variation_add_grad(p0, g0, 1., 1., 2., False)
    z = g0 == 0
    assert np.array_equal(sparse[z], g0[z]) and np.array_equal(sparse[~z], dense[~z])


def test_masked_adam_host_logic_rejects_bad_hyperparameters_and_needs_skip_flag():
    import nerf4k_amd  # noqa: F401
    from nerf4k_amd.lib.masked_adam import MaskedAdam
    w = torch.nn.Parameter(torch.zeros(4))
    for kw in (dict(lr=-1.), dict(eps=-1.), dict(betas=(1.0, 0.9)), dict(betas=(0.9, -0.1))):
        with pytest.raises(ValueError):
            MaskedAdam([{'params': [w], 'skip_zero_grad': False}], **kw)
    opt = MaskedAdam([{'params': [w]}])                    # group without 'skip_zero_grad' fails at step(), as upstream
    w.grad = torch.ones(4)
    with pytest.raises(KeyError):
        opt.step()
    opt = MaskedAdam([{'params': [w], 'skip_zero_grad': True}])
    with pytest.raises((ValueError, RuntimeError)):        # CPU tensors: no CPU path, fails loudly
        opt.step()


def test_optimizer_factory_groups_and_freezing():
    import nerf4k_amd  # noqa: F401
    from nerf4k_amd.lib import utils

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.density = torch.nn.Linear(2, 2)
            self.k0 = torch.nn.Linear(2, 2)
            self.rgbnet = None
            self.frozen = torch.nn.Parameter(torch.zeros(3))

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    cfg = Cfg(lrate_decay=20, lrate_density=0.1, lrate_k0=0.2, lrate_rgbnet=1e-3, lrate_frozen=0, lrate_missing=1.0,
              skip_zero_grad_fields=['density'])
    m = M()
    opt = utils.create_optimizer_or_freeze_model(m, cfg, global_step=10000)
    groups = {g['kname']: g for g in opt.param_groups}
    assert set(groups) == {'density', 'k0'}
    assert groups['density']['skip_zero_grad'] and not groups['k0']['skip_zero_grad']
    assert abs(groups['k0']['lr'] - 0.2 * 0.1 ** 0.5) < 1e-12
    assert m.frozen.requires_grad is False
