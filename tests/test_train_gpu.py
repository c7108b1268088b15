"""GPU parity of the training-step marcher path: autograd through the staged HIP ops --
k4_grid_sample_3d(+_backward), k4_raw2alpha(+_backward), k4_alpha2weight(+_backward), k4_segment_sum(+_backward) -- against
(a) PyTorch's own CPU autograd of the same library ops and (b) gradients produced by the REFERENCE's unmodified Python
modules (tests/golden/grad_*.npz, oracle/gen_golden.py::gen_grad).

Tolerance: gradients are sums of up to a few hundred fp32 terms accumulated by atomics in arbitrary order:
|err| <= 2e-5 * max|grad| + 1e-9 per tensor (written at each check)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import nerf4k_amd  # noqa: F401
from nerf4k_amd.lib import grid as G, dvgo, utils
from nerf4k_amd.lib.masked_adam import MaskedAdam
from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _close(got, want, name, rel=2e-5):
    got, want = got.detach().cpu().double(), want.double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= rel * scale + 1e-9, (name, err, scale)


@pytest.mark.parametrize('C,dims', [(1, (7, 6, 9)), (3, (5, 8, 4)), (9, (6, 6, 6)), (12, (4, 5, 7))])
def test_grid_sample_backward_matches_torch_autograd(C, dims):
    g = torch.Generator().manual_seed(C)
    grid = torch.randn([1, C, *dims], generator=g)
    mn, mx = torch.tensor([-1., -0.5, 0.]), torch.tensor([1., 0.7, 2.])
    n = 4000
    pts = torch.rand([n, 3], generator=g) * (mx - mn) * 1.2 + mn - 0.1 * (mx - mn)      # some outside the box
    pts[:8] = torch.stack([mn, mx, mn, mx, (mn + mx) / 2, mn, mx, mn])                   # exact boundaries
    gout = torch.randn([n, C], generator=g)
    gt = grid.clone().requires_grad_(True)
    ind = ((pts.reshape(1, 1, 1, -1, 3) - mn) / (mx - mn)).flip((-1,)) * 2 - 1
    out_ref = F.grid_sample(gt, ind, mode='bilinear', align_corners=True).reshape(C, -1).T
    out_ref.backward(gout)
    gd = grid.cuda().requires_grad_(True)
    out = G.GridSample3D.apply(gd, pts.cuda(), mn.cuda(), mx.cuda())
    _close(out, out_ref.detach(), 'forward', rel=2e-6)
    out.backward(gout.cuda())
    _close(gd.grad, gt.grad, 'grad_grid')


@pytest.mark.parametrize('C', [2, 4, 9, 12, 16, 20, 32])
@pytest.mark.parametrize('coherent', [False, True])
def test_grid_sample_backward_channel_last_path_matches_the_atomic_scatter(C, coherent):
    """k4_grid_sample_3d_backward_cl (channel-last scratch + sweep) against the channel-major atomic scatter and against fp64
    index_add of the same trilinear weights; ray-coherent samples (the z-merge across lanes fires) and random ones; two calls
    accumulate; the workspace is all zero again afterwards (it is never cleared between calls)."""
    from nerf4k_amd import _native as N
    g = torch.Generator().manual_seed(C * 2 + int(coherent))
    X, Y, Z = 11, 9, 33
    mn, mx = torch.tensor([-1., -1., -1.]), torch.tensor([1., 1., 1.])
    if coherent:
        R, S = 37, 33                                                    # z step of exactly one voxel, slow lateral drift
        o = torch.rand([R, 1, 3], generator=g) * 1.6 - 0.8
        d = torch.cat([torch.rand([R, 1, 2], generator=g) * 0.3 - 0.15, torch.full([R, 1, 1], 2.0)], -1)
        o[..., 2] = -1.0
        pts = (o + d * torch.linspace(0, 1, S).reshape(1, S, 1)).reshape(-1, 3)
    else:
        pts = torch.rand([2999, 3], generator=g) * 2.4 - 1.2              # some outside the box
    n = pts.shape[0]
    gout = torch.randn([n, C], generator=g)
    # fp64 reference: scatter of the trilinear weights (align_corners=True, zero padding)
    u = (pts.double() + 1) / 2 * torch.tensor([X - 1, Y - 1, Z - 1], dtype=torch.float64)
    b = torch.floor(u)
    f = u - b
    want = torch.zeros([C, X * Y * Z], dtype=torch.float64)
    for c in range(8):
        dx, dy, dz = (c >> 2) & 1, (c >> 1) & 1, c & 1
        ix, iy, iz = b[:, 0].long() + dx, b[:, 1].long() + dy, b[:, 2].long() + dz
        ok = (ix >= 0) & (ix < X) & (iy >= 0) & (iy < Y) & (iz >= 0) & (iz < Z)
        w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
        lin = ((ix * Y + iy) * Z + iz)[ok]
        want.index_add_(1, lin, (gout.double()[ok] * w[ok, None]).T)
    want = want.reshape(1, C, X, Y, Z)
    L = N.lib()
    dv = [t.cuda().contiguous() for t in (gout, pts, mn, mx)]             # kept alive: the entry points take raw pointers
    args = (N.f32(dv[0]), C, X, Y, Z, N.f32(dv[1]), N.f32(dv[2]), N.f32(dv[3]), n)
    plain = torch.zeros([1, C, X, Y, Z], device='cuda')
    N.check(L.k4_grid_sample_3d_backward(*args, N.f32(plain), N.stream()), 'plain')
    nbytes = int(L.k4_grid_sample_3d_backward_workspace_bytes(C, X, Y, Z))
    assert nbytes >= X * Y * Z * (4 * C + 1) and int(L.k4_grid_sample_3d_backward_workspace_bytes(1, X, Y, Z)) < 0
    ws = torch.zeros([nbytes // 4], dtype=torch.int32, device='cuda')
    cl = torch.zeros([1, C, X, Y, Z], device='cuda')
    for _ in range(2):
        N.check(L.k4_grid_sample_3d_backward_cl(*args, N.f32(cl), N.ptr(ws), N.stream()), 'cl')
    assert int(ws.count_nonzero()) == 0
    _close(plain, want.float(), 'atomic scatter')
    _close(cl, 2 * want.float(), 'channel-last path, two calls')
    assert torch.equal(cl == 0, plain == 0)                              # the same set of touched voxels


def test_segment_sum_backward_is_a_gather():
    g = torch.Generator().manual_seed(1)
    index = torch.sort(torch.randint(0, 50, [3000], generator=g)).values
    for shape in ([3000], [3000, 3]):
        src = torch.randn(shape, generator=g)
        gout = torch.randn([50] + shape[1:], generator=g)
        s = src.cuda().requires_grad_(True)
        out = dvgo.segment_sum(s, index.cuda(), 50)
        want = torch.zeros([50] + shape[1:]).index_add_(0, index, src)
        _close(out, want, 'forward', rel=2e-6)
        out.backward(gout.cuda())
        assert torch.equal(s.grad.cpu(), gout[index])


def _load_grad_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    kw = json.loads(str(z['model_kwargs_json']))
    for k in ('xyz_min', 'xyz_max'):
        kw[k] = np.asarray(kw[k], dtype=np.float32)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}
    rays = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('in/')}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('grad/')}
    ck = {'model_class': str(z['model_class']), 'model_kwargs': kw, 'model_state_dict': sd}
    return ck, json.loads(str(z['render_kwargs_json'])), rays, torch.from_numpy(z['target']), float(z['loss']), grads


@pytest.mark.parametrize('name', ['grad_mpi', 'grad_dvgo'])
def test_training_step_gradients_match_the_reference(name):
    ck, rk, rays, target, loss_ref, grads = _load_grad_golden(name)
    model = utils.model_from_checkpoint_dict(ck).cuda()
    with torch.enable_grad():
        out = model(rays['rays_o'].cuda(), rays['rays_d'].cuda(), rays['viewdirs'].cuda(), global_step=0, **rk)
        loss = F.mse_loss(out['rgb_marched'], target.cuda())
        loss.backward()
    assert abs(float(loss.detach()) - loss_ref) <= 2e-6 * max(1.0, abs(loss_ref)), (float(loss.detach()), loss_ref)
    named = dict(model.named_parameters())
    assert set(grads) <= set(named), set(grads) - set(named)
    for k, want in grads.items():
        assert named[k].grad is not None, k
        _close(named[k].grad, want, k)


def test_a_few_masked_adam_steps_with_tv_reduce_the_loss():
    """End-to-end slice of the reference's inner loop (run_sr.py:990-1014): forward, mse, backward, TV gradients in
    place, MaskedAdam step -- all grid-sized work on the HIP kernels."""
    ck, rk, rays, target, _, _ = _load_grad_golden('grad_mpi')
    model = utils.model_from_checkpoint_dict(ck).cuda()

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    cfg = Cfg(lrate_decay=20, lrate_density=1e-1, lrate_k0=1e-1, lrate_rgbnet=1e-3, skip_zero_grad_fields=['density', 'k0'])
    opt = utils.create_optimizer_or_freeze_model(model, cfg, global_step=0)
    assert isinstance(opt, MaskedAdam)
    ro, rd, vd, tgt = rays['rays_o'].cuda(), rays['rays_d'].cuda(), rays['viewdirs'].cuda(), target.cuda()
    losses = []
    for step in range(6):
        opt.zero_grad(set_to_none=True)
        with torch.enable_grad():
            loss = F.mse_loss(model(ro, rd, vd, global_step=step, **rk)['rgb_marched'], tgt)
            loss.backward()
        model.density_total_variation_add_grad(1e-6, step < 3)
        model.k0_total_variation_add_grad(1e-6, step < 3)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize('name', ['occ_mpi', 'occ_dvgo'])
def test_occupancy_and_resolution_maintenance_vs_reference_classes(name):
    """update_occupancy_cache / update_occupancy_cache_lt_nviews / voxel_count_views / maskout_near_cam_vox / scale_volume_grid
    (lib/dmpigo.py:189-246, lib/dvgo.py:186-268) on the HIP kernels, against tests/golden/occ_*.npz produced by the reference's
    own classes (oracle/gen_golden.py gen_occ).  Masks: exact except voxels whose pooled alpha sits within float rounding of the
    threshold (<= 0.2 % may flip: device expf/powf vs libm); resampled grids: 1e-5."""
    import json
    from helpers import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    kw = json.loads(str(z['model_kwargs_json']))
    for k in ('xyz_min', 'xyz_max'):
        kw[k] = np.asarray(kw[k], dtype=np.float32)
    ck = {'model_class': str(z['model_class']), 'model_kwargs': kw,
          'model_state_dict': {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd/')}}
    model = utils.model_from_checkpoint_dict(ck).cuda()
    rk = json.loads(str(z['render_kwargs_json']))

    def same_mask(got, want, what):
        got = got.cpu().numpy()
        assert got.shape == want.shape, (what, got.shape, want.shape)
        assert float((got != want).mean()) <= 2e-3, (what, float((got != want).mean()))

    with torch.no_grad():
        model.density.grid += float(z['density_plus'])
    model.update_occupancy_cache()
    same_mask(model.mask_cache.mask, z['upd/mask'], 'update_occupancy_cache')
    if name == 'occ_mpi':
        model.update_occupancy_cache_lt_nviews(torch.from_numpy(z['lt/rays_o']), torch.from_numpy(z['lt/rays_d']), [384, 384],
                                               dict(near=0, far=1, stepsize=rk['stepsize']), 1)
        same_mask(model.mask_cache.mask, z['lt/mask'], 'update_occupancy_cache_lt_nviews')
        # the fused marcher sees the refreshed occupancy (summary re-keyed on the mask's version)
        ro, rd = torch.from_numpy(z['lt/rays_o'])[:200].cuda(), torch.from_numpy(z['lt/rays_d'])[:200].cuda()
        vd = rd / rd.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            a = model(ro, rd, vd, **rk)
            b = model(ro, rd, vd, k4_staged=True, **rk)
        assert torch.allclose(a['rgb_marched'], b['rgb_marched'], atol=2e-5)
    else:
        cnt = model.voxel_count_views(torch.from_numpy(z['cnt/rays_o']), torch.from_numpy(z['cnt/rays_d']), [1, 1], near=rk['near'],
                                      far=rk['far'], stepsize=rk['stepsize'], downrate=1)
        assert float((cnt.cpu().numpy() != z['cnt/count']).mean()) <= 2e-3
        model.maskout_near_cam_vox(torch.tensor([[0.4, 0.3, 0.2], [-0.5, 0.1, 0.0]]), 0.35)
        assert np.array_equal(model.density.grid.detach().cpu().numpy() == -100, z['near/density'] == -100)
    model.scale_volume_grid(*[int(v) for v in z['new_res']])
    assert model.world_size.tolist() == z['scale/world_size'].tolist()
    np.testing.assert_allclose(model.density.grid.detach().cpu().numpy(), z['scale/density'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(model.k0.grid.detach().cpu().numpy(), z['scale/k0'], rtol=0, atol=2e-5)
    same_mask(model.mask_cache.mask, z['scale/mask'], 'scale_volume_grid')
    # and the rescaled model still renders (fused == staged)
    g = torch.Generator().manual_seed(1)
    ro = torch.tensor([[0.1, -0.2, -1.0]]).repeat(64, 1).cuda() if name == 'occ_mpi' else torch.tensor([[0.0, 0.0, 3.5]]).repeat(64, 1).cuda()
    rd = (torch.rand([64, 3], generator=g) * 0.4 - 0.2).cuda()
    rd[:, 2] = 2.0 if name == 'occ_mpi' else -1.0
    vd = rd / rd.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        a = model(ro, rd, vd, **rk)
        b = model(ro, rd, vd, k4_staged=True, **rk)
    assert torch.allclose(a['rgb_marched'], b['rgb_marched'], atol=2e-5)


@pytest.mark.parametrize('cfg,hw', [(dict(seed=71, num_voxels=64 * 64 * 48, mpi_depth=48), (64, 64)),
                                    (dict(seed=72, num_voxels=56 * 56 * 64, mpi_depth=64, stepsize=0.5), (48, 52)),          # interval != 1: the powf form, 127 samples
                                    (dict(seed=73, num_voxels=40 * 40 * 32, mpi_depth=32, mask_cache_world_size=[33, 29, 23]), (40, 40))])
def test_preselected_training_forward_equals_the_filter_by_filter_form(cfg, hw):
    """Round 5: the training forward of DirectMPIGO with its three sample filters decided up front by ONE launch
    (k4_train_select_mpi + k4_train_compact + k4_ndc_points_of, one read-back) against the op-for-op mirror of lib/dmpigo.py:300-333
    (four read-backs): every key of the returned dict bit for bit, the loss, and every gradient (the grids' up to the order of the atomic
    scatter)."""
    from nerf4k_amd import scene
    from oracle import marcher
    ck = scene.make_llff_checkpoint(**cfg)
    H, W = hw
    K = scene.LLFF_K.copy()
    K[:2] *= W / scene.LLFF_HW[1]
    rays = [x.cuda().reshape(-1, 3) for x in marcher.get_rays_of_a_view(H, W, K, scene.llff_spiral_poses()[6], ndc=True)]
    rk = dict(ck['render_kwargs'], render_depth=True)
    tgt = torch.rand([H * W, 3], generator=torch.Generator().manual_seed(3)).cuda()
    outs, grads, losses = [], [], []
    for presel in (False, True):
        model = utils.model_from_checkpoint_dict(ck).cuda()
        with torch.enable_grad():
            out = model(*rays, global_step=0, k4_presel=presel, **rk)
            loss = F.mse_loss(out['rgb_marched'], tgt) + 1e-2 * (out['weights'] * out['raw_rgb'].sum(-1)).sum() / (H * W)
            loss.backward()
        outs.append(out)
        losses.append(float(loss.detach()))
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    a, b = outs
    assert set(a) == set(b)
    assert a['ray_id'].numel() > 1000                                  # a non-trivial batch
    for k in a:
        if torch.is_tensor(a[k]):
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k
    assert losses[0] == losses[1]
    assert set(grads[0]) == set(grads[1]) and 'density.grid' in grads[0] and 'k0.grid' in grads[0]
    for k in grads[0]:
        _close(grads[1][k], grads[0][k].cpu(), k)


def test_preselected_training_forward_on_an_empty_scene_and_on_zero_rays():
    """Edge cases of the pre-selected training forward: a scene in which no sample passes the alpha threshold (empty lists end to end: the
    loss is the background term alone and every gradient is zero or absent) and a batch of zero rays -- same results as the filter-by-filter
    form, no crash."""
    from nerf4k_amd import scene
    from oracle import marcher
    ck = scene.make_llff_checkpoint(seed=74, num_voxels=40 * 40 * 32, mpi_depth=32)
    ck['model_state_dict']['density.grid'] = ck['model_state_dict']['density.grid'] - 60.0          # alpha ~ 0 everywhere
    H, W = 24, 32
    K = scene.LLFF_K.copy()
    K[:2] *= W / scene.LLFF_HW[1]
    rays = [x.cuda().reshape(-1, 3) for x in marcher.get_rays_of_a_view(H, W, K, scene.llff_spiral_poses()[2], ndc=True)]
    rk = dict(ck['render_kwargs'], render_depth=True)
    outs = []
    for presel in (False, True):
        model = utils.model_from_checkpoint_dict(ck).cuda()
        with torch.enable_grad():
            out = model(*rays, global_step=0, k4_presel=presel, **rk)
            loss = out['rgb_marched'].sum() + out['alphainv_last'].sum()
            if loss.requires_grad:
                loss.backward()
        outs.append(out)
        empty = model(rays[0][:0], rays[1][:0], rays[2][:0], global_step=0, k4_presel=presel, **rk)
        assert empty['rgb_marched'].shape == (0, 3) and empty['ray_id'].numel() == 0
    a, b = outs
    assert a['ray_id'].numel() == 0 and b['ray_id'].numel() == 0
    for k in a:
        if torch.is_tensor(a[k]):
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    assert torch.equal(b['alphainv_last'], torch.ones_like(b['alphainv_last']))
