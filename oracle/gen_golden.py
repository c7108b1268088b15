"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own Python in the build container.
TEST INFRASTRUCTURE ONLY.   Usage:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Needs ``/root/reference`` (absent on the GPU box -- the fixtures travel instead).

  march_*.npz : reference ``DirectMPIGO`` / ``DirectVoxGO`` (lib/dmpigo.py, lib/dvgo.py,
                lib/grid.py imported unmodified, see oracle/ref_import.py) evaluated on small
                seeded scenes.  Pins the Python-level control flow of oracle/marcher.py.  The 13
                native kernels underneath are oracle/native_cpu.py (the .cu files cannot run here).
  rays_*.npz  : reference ``get_rays_of_a_view`` (lib/dvgo.py:577-582).
  sr_*.npz    : UNMODIFIED reference ``SFTNet.forward`` / ``tile_process`` (lib/sr_esrnet.py);
                weights from oracle.sr.make_state_dict(seed) (stored as the seed, not the tensors).
"""
import io
import contextlib
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import, marcher, sr as osr   # noqa: E402
import nerf4k_amd                                    # noqa: E402,F401
from nerf4k_amd import scene                         # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _np(v):
    if torch.is_tensor(v):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def _kwargs_json(kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, np.ndarray):
            out[k] = v.tolist()
        elif torch.is_tensor(v):
            out[k] = v.tolist()
        else:
            out[k] = v
    return json.dumps(out)


def _save_march(name, ck, rays, ref_out):
    arrs = {'model_class': np.array(ck['model_class']),
            'model_kwargs_json': np.array(_kwargs_json(ck['model_kwargs'])),
            'render_kwargs_json': np.array(json.dumps(ck['render_kwargs']))}
    for k, v in ck['model_state_dict'].items():
        arrs['sd/' + k] = _np(v)
    for k, v in zip(('rays_o', 'rays_d', 'viewdirs'), rays):
        arrs['in/' + k] = _np(v)
    for k, v in ref_out.items():
        arrs['out/' + k] = _np(v)
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB,',
          {k: tuple(_np(v).shape) for k, v in ref_out.items()})


def _ref_model(ref, ck):
    cls = {'DirectMPIGO': ref.dmpigo.DirectMPIGO, 'DirectVoxGO': ref.dvgo.DirectVoxGO}[ck['model_class']]
    with contextlib.redirect_stdout(io.StringIO()):
        model = cls(**ck['model_kwargs'])
    model.load_state_dict(ck['model_state_dict'])
    return model.eval()


def _llff_rays(ref, H, W, pose, subsample):
    K = scene.LLFF_K.copy()
    K[:2] *= (W / scene.LLFF_HW[1])
    ro, rd, vd = ref.dvgo.get_rays_of_a_view(H, W, K, torch.Tensor(pose), True,
                                             inverse_y=False, flip_x=False, flip_y=False)
    sel = slice(None, None, subsample)
    return [x.flatten(0, -2)[sel].contiguous() for x in (ro, rd, vd)]


def gen_march(ref):
    poses = scene.llff_spiral_poses()
    cases = {
        'march_mpi_base': dict(seed=11, num_voxels=14 * 14 * 12, mpi_depth=12),
        'march_mpi_pe': dict(seed=12, num_voxels=12 * 12 * 10, mpi_depth=10, viewbase_pe=2, spatial_pe=1,
                             rgbnet_dim=5, rgbnet_width=32, rgbnet_depth=4),
        # (rgbnet_dim=0 is unreachable in the reference: DirectMPIGO.forward reads self.dim_rend,
        #  which __init__ only sets when rgbnet_dim>0 -- lib/dmpigo.py:69-88,385 -> AttributeError)
        'march_mpi_half': dict(seed=14, num_voxels=14 * 14 * 12, mpi_depth=12, stepsize=0.5),
    }
    for name, cfg in cases.items():
        ck = scene.make_llff_checkpoint(**cfg)
        rays = _llff_rays(ref, 24, 32, poses[5], subsample=3)
        with torch.no_grad():
            out = _ref_model(ref, ck)(*rays, **ck['render_kwargs'])
        _save_march(name, ck, rays, out)

    cases = {
        'march_dvgo_base': dict(seed=21, num_voxels=12 ** 3),
        'march_dvgo_nodirect': dict(seed=22, num_voxels=11 ** 3, rgbnet_direct=False, rgbnet_dim=9,
                                    rgbnet_width=32, viewbase_pe=2),
        'march_dvgo_coarse': dict(seed=23, num_voxels=11 ** 3, rgbnet_dim=0, fast_color_thres=1e-7,
                                  alpha_init=1e-6),
    }
    for name, cfg in cases.items():
        ck = scene.make_lego_checkpoint(**cfg)
        H = W = 20
        ro, rd, vd = ref.dvgo.get_rays_of_a_view(H, W, scene.lego_K(H, W), torch.Tensor(scene.lego_pose()),
                                                 False, inverse_y=False, flip_x=False, flip_y=False)
        rays = [x.flatten(0, -2)[::2].contiguous() for x in (ro, rd, vd)]
        # one ray that misses the box and one with a zero direction component (render_utils_kernel.cu:23-25,53)
        rays[0] = torch.cat([rays[0], torch.tensor([[5., 5., 5.], [0., 0., 4.]])])
        rays[1] = torch.cat([rays[1], torch.tensor([[1., 0.2, 0.1], [0., 0., -1.]])])
        rays[2] = torch.cat([rays[2], torch.tensor([[0.97, 0.2, 0.1], [0., 0., -1.]])])
        with torch.no_grad():
            out = _ref_model(ref, ck)(*rays, **ck['render_kwargs'])
        _save_march(name, ck, rays, out)


def gen_rays(ref):
    poses = scene.llff_spiral_poses()
    arrs = {}
    H, W = 9, 12
    K = scene.LLFF_K.copy()
    K[:2] *= W / scene.LLFF_HW[1]
    for tag, ndc, pose, Kc in (('ndc', True, poses[3], K), ('persp', False, scene.lego_pose()[:3, :4],
                                                            scene.lego_K(H, W))):
        ro, rd, vd = ref.dvgo.get_rays_of_a_view(H, W, Kc, torch.Tensor(pose), ndc,
                                                 inverse_y=False, flip_x=False, flip_y=False)
        arrs.update({f'{tag}/K': Kc, f'{tag}/c2w': pose, f'{tag}/rays_o': _np(ro),
                     f'{tag}/rays_d': _np(rd), f'{tag}/viewdirs': _np(vd)})
    arrs['H'], arrs['W'] = np.array(H), np.array(W)
    np.savez_compressed(os.path.join(GOLDEN, 'rays_views.npz'), **arrs)
    print('rays_views: ok')


def gen_sr(ref):
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    for name, nb, hw, tile in (('sr_full5', 5, (12, 16), None), ('sr_tiles', 2, (23, 30), 12),
                               ('sr_tiles510geom', 1, (26, 37), 16)):
        sd = osr.make_state_dict(seed=100 + nb, num_block=nb)
        net = ref.sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=nb, num_grow_ch=32, num_cond=1)
        missing = set(net.state_dict().keys()) ^ set(sd.keys())
        assert not missing, missing
        assert list(net.state_dict().keys()) == [k for k, _ in osr.state_dict_spec(num_block=nb)]
        net.load_state_dict(sd)
        net.eval()
        x = torch.rand([1, 3, *hw], generator=g)
        cond = torch.rand([1, *hw], generator=g)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            if tile is None:
                y = net(x, cond.unsqueeze(0))
            else:
                y = net.tile_process(x, cond, tile_size=tile)
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), seed=np.array(100 + nb), num_block=np.array(nb),
                            tile=np.array(-1 if tile is None else tile), x=_np(x), cond=_np(cond), y=_np(y))
        print(name, tuple(y.shape), float(y.abs().mean()))


def gen_grad(ref):
    """Training-step gradients of the reference's own modules (autograd through Raw2Alpha / Alphas2Weights with the restated
    backward kernels, torch's grid_sample backward, index_add): d(mse(rgb_marched, target))/d(parameters)."""
    poses = scene.llff_spiral_poses()
    cases = [('grad_mpi', scene.make_llff_checkpoint(seed=15, num_voxels=14 * 14 * 12, mpi_depth=12), None),
             ('grad_dvgo', scene.make_lego_checkpoint(seed=25, num_voxels=12 ** 3), 20)]
    for name, ck, hw in cases:
        if hw is None:
            rays = _llff_rays(ref, 24, 32, poses[5], subsample=3)
        else:
            ro, rd, vd = ref.dvgo.get_rays_of_a_view(hw, hw, scene.lego_K(hw, hw), torch.Tensor(scene.lego_pose()),
                                                     False, inverse_y=False, flip_x=False, flip_y=False)
            rays = [x.flatten(0, -2)[::2].contiguous() for x in (ro, rd, vd)]
        model = _ref_model(ref, ck)
        g = torch.Generator().manual_seed(77)
        target = torch.rand([rays[0].shape[0], 3], generator=g)
        out = model(*rays, global_step=0, **ck['render_kwargs'])
        loss = torch.nn.functional.mse_loss(out['rgb_marched'], target)
        loss.backward()
        arrs = {'model_class': np.array(ck['model_class']),
                'model_kwargs_json': np.array(_kwargs_json(ck['model_kwargs'])),
                'render_kwargs_json': np.array(json.dumps(ck['render_kwargs'])),
                'target': _np(target), 'loss': _np(loss.detach())}
        for k, v in ck['model_state_dict'].items():
            arrs['sd/' + k] = _np(v)
        for k, v in zip(('rays_o', 'rays_d', 'viewdirs'), rays):
            arrs['in/' + k] = _np(v)
        ng = 0
        for k, prm in model.named_parameters():
            if prm.grad is not None:
                arrs['grad/' + k] = _np(prm.grad)
                ng += 1
        path = os.path.join(GOLDEN, name + '.npz')
        np.savez_compressed(path, **arrs)
        print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, loss {float(loss):.6f}, {ng} parameter gradients')


def gen_grad_sr(ref):
    """Training-step gradients of the UNMODIFIED reference SFTNet (lib/sr_esrnet.py, PyTorch CPU autograd): L1 loss against a
    random target (run_sr.py trains the decoder with an L1 term), gradients w.r.t. every parameter and w.r.t. the inputs (the joint
    loop back-propagates into the marcher).  Full tensors are stored for a selection of parameters, (sum, L2 norm) for all."""
    nb = 2
    sd = osr.make_state_dict(seed=300, num_block=nb)
    net = ref.sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=nb, num_grow_ch=32, num_cond=1)
    net.load_state_dict(sd)
    net.train()
    g = torch.Generator().manual_seed(9)
    h, w = 17, 22
    x = torch.rand([1, 3, h, w], generator=g, requires_grad=True)
    cond = torch.rand([1, 1, h, w], generator=g, requires_grad=True)
    target = torch.rand([1, 3, 4 * h, 4 * w], generator=g)
    out = net(x, cond)
    loss = torch.nn.functional.l1_loss(out, target)
    loss.backward()
    full = ['conv_first.weight', 'conv_first.bias', 'CondNet.0.weight', 'CondNet.6.weight', 'CondNet.6.bias', 'conv_last.weight',
            'conv_last.bias', 'conv_hr.bias', 'conv_up1.bias', 'body.0.rdb1.conv1.weight', 'body.1.rdb3.conv5.bias',
            'body.0.rdb2.sft1.SFT_shift_conv1.weight', 'body.1.sft0.SFT_scale_conv0.weight', 'sftbody.SFT_scale_conv1.bias']
    arrs = {'seed': np.array(300), 'num_block': np.array(nb), 'x': _np(x), 'cond': _np(cond), 'target': _np(target), 'loss': _np(loss.detach()),
            'out': _np(out.detach()), 'grad_x': _np(x.grad), 'grad_cond': _np(cond.grad)}
    names, stats = [], []
    for k, p in net.named_parameters():
        names.append(k)
        stats.append([float(p.grad.double().sum()), float(p.grad.double().norm())])
        if k in full:
            arrs['grad/' + k] = _np(p.grad)
    arrs['names'] = np.array(names)
    arrs['stats'] = np.array(stats, dtype=np.float64)
    path = os.path.join(GOLDEN, 'grad_sr.npz')
    np.savez_compressed(path, **arrs)
    print(f'grad_sr: {os.path.getsize(path) / 1024:.1f} KiB, loss {float(loss):.6f}, {len(names)} parameters, |grad_x| {float(x.grad.abs().max()):.3e}')


def gen_occ(ref):
    """Occupancy / resolution maintenance of the training loop, run on the reference's own classes: update_occupancy_cache,
    scale_volume_grid (lib/dmpigo.py:189-226, lib/dvgo.py:200-233), update_occupancy_cache_lt_nviews / voxel_count_views."""
    poses = scene.llff_spiral_poses()
    for name, ck, new_res in (('occ_mpi', scene.make_llff_checkpoint(seed=16, num_voxels=14 * 14 * 12, mpi_depth=12), (18 * 18 * 12, 12)),       # mpi_depth stays fixed upstream (act_shift is per plane)
                              ('occ_dvgo', scene.make_lego_checkpoint(seed=26, num_voxels=12 ** 3), (15 ** 3,))):
        arrs = {'model_class': np.array(ck['model_class']), 'model_kwargs_json': np.array(_kwargs_json(ck['model_kwargs'])),
                'render_kwargs_json': np.array(json.dumps(ck['render_kwargs'])), 'new_res': np.array(new_res)}
        for k, v in ck['model_state_dict'].items():
            arrs['sd/' + k] = _np(v)
        model = _ref_model(ref, ck)
        with torch.no_grad():
            model.density.grid += (0.0 if ck['model_class'] == 'DirectMPIGO' else -4.0)   # make the density disagree with the stored mask: the update must prune
        arrs['density_plus'] = np.array(0.0 if ck['model_class'] == 'DirectMPIGO' else -4.0)
        with contextlib.redirect_stdout(io.StringIO()):
            model.update_occupancy_cache()
        arrs['upd/mask'] = _np(model.mask_cache.mask).copy()
        # views-per-voxel utilities on a handful of rays per "image"
        if ck['model_class'] == 'DirectMPIGO':
            rays = _llff_rays(ref, 24, 32, poses[5], subsample=1)
            ro, rd = rays[0], rays[1]
            with contextlib.redirect_stdout(io.StringIO()):
                model.update_occupancy_cache_lt_nviews(ro, rd, [384, 384], dict(near=0, far=1, stepsize=ck['render_kwargs']['stepsize']), 1)
            arrs['lt/rays_o'], arrs['lt/rays_d'] = _np(ro), _np(rd)
            arrs['lt/mask'] = _np(model.mask_cache.mask).copy()
        else:
            H = W = 20
            ro, rd, _ = ref.dvgo.get_rays_of_a_view(H, W, scene.lego_K(H, W), torch.Tensor(scene.lego_pose()), False,
                                                    inverse_y=False, flip_x=False, flip_y=False)
            ro2 = torch.stack([ro, ro]); rd2 = torch.stack([rd, rd * torch.tensor([1.0, 0.9, 1.1])])
            with contextlib.redirect_stdout(io.StringIO()):
                cnt = model.voxel_count_views(ro2, rd2, [1, 1], near=ck['render_kwargs']['near'], far=ck['render_kwargs']['far'],
                                              stepsize=ck['render_kwargs']['stepsize'], downrate=1)
            arrs['cnt/rays_o'], arrs['cnt/rays_d'], arrs['cnt/count'] = _np(ro2), _np(rd2), _np(cnt)
            with torch.no_grad():
                model.maskout_near_cam_vox(torch.tensor([[0.4, 0.3, 0.2], [-0.5, 0.1, 0.0]]), 0.35)
            arrs['near/density'] = _np(model.density.grid).copy()
        with contextlib.redirect_stdout(io.StringIO()):
            model.scale_volume_grid(*new_res)
        arrs['scale/world_size'] = _np(model.world_size)
        arrs['scale/density'], arrs['scale/k0'], arrs['scale/mask'] = _np(model.density.grid), _np(model.k0.grid), _np(model.mask_cache.mask)
        path = os.path.join(GOLDEN, name + '.npz')
        np.savez_compressed(path, **arrs)
        print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, mask {float(arrs["upd/mask"].mean()):.3f} -> scale {arrs["scale/world_size"].tolist()} '
              f'mask {float(arrs["scale/mask"].mean()):.3f}')


def gen_patch(ref):
    """The 'patch_mimg' ray sampler of the joint loop (lib/dvgo.py:822-880) under a fixed numpy seed: the first draws of the
    reference's own generator, for a non-divisible image (LLFF-like remainders) and an exactly divisible one (empty remainder
    entries upstream)."""
    arrs = {}
    for tag, imsz, num_im, BS, szp, sr, draws in (('a', (20, 27), 3, 32, 8, 4, 60), ('b', (16, 16), 2, 64, 16, 2, 24), ('c', (756, 1008), 2, 4096, 64, 4, 6)):
        np.random.seed(1234)
        gen = ref.dvgo.mimg_patch_indices_generator(np.array(imsz), num_im, BS, szp, sr)
        rows = []
        for _ in range(draws):
            im, r, c, r4, c4, ps = next(gen)
            r, c, r4, c4 = (np.asarray(v, dtype=np.int64) for v in (r, c, r4, c4))
            # a draw is summarised by sizes + position-weighted checksums (the index lists of a 64x64 / 256x256 patch pair are 70k numbers)
            chk = [int((v * (np.arange(v.size) % 97 + 1)).sum()) for v in (r, c, r4, c4)]
            rows.append([int(im), ps[0], ps[1], r.size, r4.size] + chk + [int(r.min()) if r.size else -1, int(c.min()) if c.size else -1])
        arrs[tag + '/args'] = np.array([imsz[0], imsz[1], num_im, BS, szp, sr])
        arrs[tag + '/draws'] = np.array(rows, dtype=np.int64)
    np.savez_compressed(os.path.join(GOLDEN, 'patch_sampler.npz'), **arrs)
    print('patch_sampler:', {k: v.shape for k, v in arrs.items()})


def gen_grad_joint(ref):
    """ONE iteration of the joint loop (run_sr.py:869-1000) on the reference's own modules: DirectMPIGO.forward (train) ->
    SFTNet(rgb_feature, depth) -> L1(LR) + L1(HR) + entropy + distortion + per-point rgb (oracle/train_ops.joint_losses restates the
    loss lines; the distortion term is the literal O(n^2) definition, torch_efficient_distloss is not installed) -> backward.
    Stores every loss term, the marcher's parameter gradients in full and (sum, L2 norm) of every decoder gradient."""
    from oracle import train_ops as oto
    poses = scene.llff_spiral_poses()
    ck = scene.make_llff_checkpoint(seed=17, num_voxels=14 * 14 * 12, mpi_depth=12)
    model = _ref_model(ref, ck).train()
    H, W, pr, pc, r0, c0 = 24, 32, 8, 10, 9, 13
    K = scene.LLFF_K.copy()
    K[:2] *= (W / scene.LLFF_HW[1])
    ro, rd, vd = ref.dvgo.get_rays_of_a_view(H, W, K, torch.Tensor(poses[7]), True, inverse_y=False, flip_x=False, flip_y=False)
    rays = [x[r0:r0 + pr, c0:c0 + pc].reshape(-1, 3).contiguous() for x in (ro, rd, vd)]
    nb = 1
    sd = osr.make_state_dict(seed=400, num_block=nb)
    net = ref.sr_esrnet.SFTNet(3, scale=4, num_feat=64, num_block=nb, num_grow_ch=32, num_cond=1)
    net.load_state_dict(sd)
    net.train()
    g = torch.Generator().manual_seed(31)
    target = torch.rand([pr * pc, 3], generator=g)
    target_4x = torch.rand([16 * pr * pc, 3], generator=g)
    cfg = dict(weight_main=1.0, weight_entropy_last=0.001, weight_distortion=0.01, weight_rgbper=0.01)
    rk = dict(ck['render_kwargs'], render_depth=True, rand_bkgd=True)
    torch.manual_seed(3)
    rr = model(*rays, global_step=5, is_train=True, **rk)
    rgb_cache = rr['rgb_feature'].reshape(1, pr, pc, -1).movedim(-1, 1)
    cond = rr['depth'].reshape(1, pr, pc, 1).movedim(-1, 1)
    rgb_sr = net(rgb_cache, cond)
    total, terms = oto.joint_losses(rr, rgb_sr, target, target_4x, pr, pc, cfg)
    total.backward()
    arrs = {'model_class': np.array(ck['model_class']), 'model_kwargs_json': np.array(_kwargs_json(ck['model_kwargs'])),
            'render_kwargs_json': np.array(json.dumps(rk)), 'cfg_json': np.array(json.dumps(cfg)),
            'sr_seed': np.array(400), 'num_block': np.array(nb), 'patch': np.array([pr, pc]), 'global_step': np.array(5),
            'target': _np(target), 'target_4x': _np(target_4x), 'loss/total': _np(total.detach()),
            'out/rgb_sr': _np(rgb_sr.detach()), 'out/rgb_feature': _np(rr['rgb_feature'].detach()), 'out/depth': _np(rr['depth'])}
    for k, v in terms.items():
        arrs['loss/' + k] = _np(v.detach())
    for k, v in ck['model_state_dict'].items():
        arrs['sd/' + k] = _np(v)
    for k, v in zip(('rays_o', 'rays_d', 'viewdirs'), rays):
        arrs['in/' + k] = _np(v)
    ng = 0
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            arrs['grad/' + k] = _np(prm.grad)
            ng += 1
    names, stats = [], []
    for k, prm in net.named_parameters():
        names.append(k)
        stats.append([float(prm.grad.double().sum()), float(prm.grad.double().norm())])
    arrs['sr_names'], arrs['sr_stats'] = np.array(names), np.array(stats, dtype=np.float64)
    for k in ('conv_first.weight', 'CondNet.0.weight', 'conv_last.bias'):
        arrs['sr_grad/' + k] = _np(dict(net.named_parameters())[k].grad)
    path = os.path.join(GOLDEN, 'grad_joint.npz')
    np.savez_compressed(path, **arrs)
    print(f'grad_joint: {os.path.getsize(path) / 1024:.1f} KiB, total {float(total):.6f}', {k: float(v) for k, v in terms.items()},
          f'{ng} marcher gradients, {len(names)} decoder gradients, samples {rr["weights"].numel()}')


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = ref_import.load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'grad':
        gen_grad(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'occ':
        gen_occ(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'grad_sr':
        gen_grad_sr(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'joint':
        gen_patch(ref)
        gen_grad_joint(ref)
        return
    gen_rays(ref)
    gen_march(ref)
    gen_sr(ref)
    gen_grad(ref)
    gen_occ(ref)
    gen_grad_sr(ref)
    gen_patch(ref)
    gen_grad_joint(ref)


if __name__ == '__main__':
    main()
