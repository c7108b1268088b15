"""CPU oracle of the voxel-grid ray marcher.  TEST INFRASTRUCTURE ONLY.

Restates, step for step and in fp32, what the reference computes in
  * ``DirectMPIGO.forward``   -- /root/reference/lib/dmpigo.py:292-427  (LLFF / NDC scenes)
  * ``DirectVoxGO.forward``   -- /root/reference/lib/dvgo.py:327-448    (bounded scenes)
  * ``DenseGrid.forward``     -- /root/reference/lib/grid.py:117-128
  * ``MaskGrid.forward``      -- /root/reference/lib/grid.py:295-304
  * ``get_rays`` / ``ndc_rays`` / ``get_rays_of_a_view`` -- lib/dvgo.py:516-582
on top of ``oracle/native_cpu.py`` (the 13 native kernels) with
``torch.nn.functional.grid_sample`` (PyTorch CPU, the reference's own provider of the
trilinear gather, lib/grid.py:124) and ``index_add_`` for ``torch_scatter.segment_coo``
(lib/dmpigo.py:382-386, lib/dvgo.py:415-419; torch_scatter is unpinned and not vendored).

The model is described by the reference checkpoint contents: ``model_kwargs`` and
``model_state_dict`` (lib/utils.py:62-66), nothing else.

Pin: ``tests/golden/march_*.npz`` were produced by the reference's own Python
classes run in the build container (``oracle/gen_golden.py`` via ``oracle/ref_import.py``);
``tests/test_oracle_golden.py`` checks this file against them.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import native_cpu as nat


# ---------------------------------------------------------------------------
# rays                                                  lib/dvgo.py:516-582
# ---------------------------------------------------------------------------
def get_rays(H, W, K, c2w, inverse_y=False, flip_x=False, flip_y=False, mode='center'):
    """Pixel (row h, col w) -> camera ray; pixel centres at +0.5 (lib/dvgo.py:516-544)."""
    K = torch.as_tensor(np.asarray(K), dtype=torch.float32)
    c2w = torch.as_tensor(np.asarray(c2w), dtype=torch.float32)
    i = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    j = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    if mode == 'center':
        i, j = i + 0.5, j + 0.5
    elif mode != 'lefttop':
        raise NotImplementedError(mode)
    if flip_x:
        i = i.flip((1,))
    if flip_y:
        j = j.flip((0,))
    if inverse_y:
        dirs = torch.stack([(i - K[0][2]) / K[0][0], (j - K[1][2]) / K[1][1], torch.ones_like(i)], -1)
    else:
        dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """lib/dvgo.py:557-574"""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y=False, flip_x=False, flip_y=False, mode='center'):
    """viewdirs are normalised BEFORE the NDC warp (lib/dvgo.py:577-582)."""
    rays_o, rays_d = get_rays(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y, mode=mode)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    if ndc:
        K = np.asarray(K)
        rays_o, rays_d = ndc_rays(H, W, float(K[0][0]), 1., rays_o, rays_d)
    return rays_o, rays_d, viewdirs


# ---------------------------------------------------------------------------
# grids                                                  lib/grid.py
# ---------------------------------------------------------------------------
def dense_grid(grid, xyz, xyz_min, xyz_max):
    """DenseGrid.forward (lib/grid.py:117-128): normalise to [-1,1], flip xyz->zyx,
    trilinear ``grid_sample`` with align_corners=True and zero padding."""
    C = grid.shape[1]
    shape = xyz.shape[:-1]
    ind_norm = ((xyz.reshape(1, 1, 1, -1, 3) - xyz_min) / (xyz_max - xyz_min)).flip((-1,)) * 2 - 1
    out = F.grid_sample(grid, ind_norm, mode='bilinear', align_corners=True)
    out = out.reshape(C, -1).T.reshape(*shape, C)
    if C == 1:
        out = out.squeeze(-1)
    return out


def mask_grid(mask, xyz, xyz2ijk_scale, xyz2ijk_shift):
    """MaskGrid.forward (lib/grid.py:295-304)"""
    shape = xyz.shape[:-1]
    return nat.maskcache_lookup(mask, xyz.reshape(-1, 3), xyz2ijk_scale, xyz2ijk_shift).reshape(shape)


def mask_scale_shift(mask_shape, xyz_min, xyz_max):
    """lib/grid.py:291-293"""
    scale = (torch.tensor(list(mask_shape), dtype=torch.float32) - 1) / (xyz_max - xyz_min)
    shift = -xyz_min * scale
    return scale, shift


# ---------------------------------------------------------------------------
# colour MLP                                              lib/dmpigo.py:112-120, lib/dvgo.py:116-124
# ---------------------------------------------------------------------------
def _rgbnet_layers(sd):
    """Collect (weight, bias) of ``Sequential(Linear, act, *[Sequential(Linear, act)], Linear)``
    in evaluation order from the reference state_dict key names
    (rgbnet.0.*, rgbnet.2.0.*, rgbnet.3.0.* ..., rgbnet.<last>.*)."""
    keys = [k for k in sd if k.startswith('rgbnet.') and k.endswith('.weight')]
    if not keys:
        return None

    def order(k):
        return [int(p) for p in k.split('.')[1:-1]]
    keys.sort(key=order)
    return [(sd[k].float(), sd[k[:-len('weight')] + 'bias'].float()) for k in keys]


def _mlp(layers, x):
    for li, (w, b) in enumerate(layers):
        x = F.linear(x, w, b)
        if li + 1 < len(layers):
            x = F.relu(x)
    return x


def _pe(v, freq):
    """[v, sin(v (x) freq), cos(...)], component-major / frequency-minor flattening
    (lib/dmpigo.py:347-351, lib/dvgo.py:387-388)."""
    e = (v.unsqueeze(-1) * freq).flatten(-2)
    return torch.cat([v, e.sin(), e.cos()], -1)


def _segment_sum(src, index, n):
    """torch_scatter.segment_coo(src, index, out=zeros, reduce='sum') for sorted index."""
    out = torch.zeros([n] + list(src.shape[1:]), dtype=src.dtype, device=src.device)
    if src.numel():
        out.index_add_(0, index, src)
    return out


# ---------------------------------------------------------------------------
# DirectMPIGO.forward                                     lib/dmpigo.py:292-427
# ---------------------------------------------------------------------------
def mpi_forward(model_kwargs, sd, rays_o, rays_d, viewdirs, near=0, far=1, stepsize=1.0, bg=0,
                render_depth=False, counters=None, **_ignored):
    assert near == 0 and far == 1                                   # lib/dmpigo.py:275
    rays_o = rays_o.float().contiguous()
    rays_d = rays_d.float().contiguous()
    viewdirs = viewdirs.float()
    N = rays_o.shape[0]
    xyz_min = sd['xyz_min'].float()
    xyz_max = sd['xyz_max'].float()
    mpi_depth = int(model_kwargs['mpi_depth'])
    voxel_size_ratio = 256. / mpi_depth                              # lib/dmpigo.py:164
    thres = float(model_kwargs.get('fast_color_thres', 0))
    N_samples = int((mpi_depth - 1) / stepsize) + 1                  # lib/dmpigo.py:278
    interval = stepsize * voxel_size_ratio                           # lib/dmpigo.py:306

    # sample_ray (lib/dmpigo.py:263-290)
    pts, mask_outbbox = nat.sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples)
    mask_inbbox = ~mask_outbbox
    ray_pts = pts.view(-1, 3)[mask_inbbox.view(-1)]
    dev = rays_o.device          # CPU everywhere except bench.py's `reference_pipeline_rocm` leg (reference-compiled kernels on the GPU)
    ray_id = torch.arange(N, device=dev).view(-1, 1).expand_as(mask_inbbox)[mask_inbbox]
    step_id = torch.arange(N_samples, device=dev).view(1, -1).expand_as(mask_inbbox)[mask_inbbox]
    n_inbbox = ray_pts.shape[0]

    # skip known free space (lib/dmpigo.py:309-313)
    mask1 = mask_grid(sd['mask_cache.mask'], ray_pts, sd['mask_cache.xyz2ijk_scale'].float(),
                      sd['mask_cache.xyz2ijk_shift'].float())
    ray_pts, ray_id, step_id = ray_pts[mask1], ray_id[mask1], step_id[mask1]
    n_mask = ray_pts.shape[0]

    # density + per-plane act_shift GRID, Raw2Alpha shift 0 (lib/dmpigo.py:316-317, 258-261)
    density = dense_grid(sd['density.grid'].float(), ray_pts, xyz_min, xyz_max) \
        + dense_grid(sd['act_shift.grid'].float(), ray_pts, xyz_min, xyz_max)
    _, alpha = nat.raw2alpha(density.flatten(), 0, interval)
    if thres > 0:
        mask2 = alpha > thres                                        # strict > (lib/dmpigo.py:319)
        ray_pts, ray_id, step_id, alpha = ray_pts[mask2], ray_id[mask2], step_id[mask2], alpha[mask2]
    n_alpha = ray_pts.shape[0]

    weights, _, alphainv_last, _, _ = nat.alpha2weight(alpha, ray_id, N)
    if thres > 0:
        mask3 = weights > thres                                      # lib/dmpigo.py:328
        ray_pts, ray_id, step_id = ray_pts[mask3], ray_id[mask3], step_id[mask3]
        alpha, weights = alpha[mask3], weights[mask3]
    n_shade = ray_pts.shape[0]

    # colour (lib/dmpigo.py:336-379)
    vox_emb = dense_grid(sd['k0.grid'].float(), ray_pts, xyz_min, xyz_max)
    if vox_emb.dim() == 1:
        vox_emb = vox_emb.unsqueeze(-1)
    layers = _rgbnet_layers(sd)
    if layers is None:
        rgb_raw = torch.sigmoid(vox_emb)
    else:
        pe_spa = ((ray_pts - xyz_min) / (xyz_max - xyz_min)).flip((-1,)) * 2 - 1
        viewdirs_emb = _pe(viewdirs, sd['viewfreq'].float())[ray_id]
        pe_emb = _pe(pe_spa, sd['posfreq'].float())
        rgb_feat = torch.cat([vox_emb, pe_emb, viewdirs_emb], -1)
        rgb_raw = torch.sigmoid(_mlp(layers, rgb_feat))

    # compositing; rgb_marched ALIASES rgb_feature in eval (lib/dmpigo.py:382-397)
    rgb_feature = _segment_sum(weights.unsqueeze(-1) * rgb_raw, ray_id, N)
    rgb_marched = rgb_feature
    rgb_marched += alphainv_last.unsqueeze(-1) * bg
    s = (step_id + 0.5) / N_samples                                  # original step index (lib/dmpigo.py:398)
    ret = {
        'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
        'rgb_feature': rgb_feature, 'raw_alpha': alpha, 'raw_rgb': rgb_raw, 'ray_id': ray_id,
        'n_max': N_samples, 's': s,
    }
    if render_depth:
        ret['depth'] = _segment_sum(weights * s, ray_id, N)           # lib/dmpigo.py:418-425
    if counters is not None:
        counters.update(n_rays=N, n_total=N * N_samples, n_inbbox=n_inbbox, n_mask=n_mask,
                        n_alpha=n_alpha, n_shade=n_shade)
    return ret


# ---------------------------------------------------------------------------
# DirectVoxGO.forward                                     lib/dvgo.py:327-448
# ---------------------------------------------------------------------------
def dvgo_geometry(model_kwargs, sd):
    """Restates DirectVoxGO.__init__/_set_grid_resolution (lib/dvgo.py:41-46, 152-158) in the
    same fp32 tensor arithmetic so stepdist / interval round identically."""
    xyz_min = sd['xyz_min'].float()
    xyz_max = sd['xyz_max'].float()
    num_voxels = model_kwargs['num_voxels']
    num_voxels_base = model_kwargs['num_voxels_base']
    voxel_size_base = ((xyz_max - xyz_min).prod() / num_voxels_base).pow(1 / 3)
    voxel_size = ((xyz_max - xyz_min).prod() / num_voxels).pow(1 / 3)
    world_size = ((xyz_max - xyz_min) / voxel_size).long()
    return dict(voxel_size=voxel_size, voxel_size_base=voxel_size_base, world_size=world_size,
                max_world_size=world_size.max(), voxel_size_ratio=voxel_size / voxel_size_base)


def dvgo_forward(model_kwargs, sd, rays_o, rays_d, viewdirs, near, far, stepsize, bg=0,
                 render_depth=False, counters=None, **_ignored):
    rays_o = rays_o.float().contiguous()
    rays_d = rays_d.float().contiguous()
    viewdirs = viewdirs.float()
    N = rays_o.shape[0]
    xyz_min = sd['xyz_min'].float()
    xyz_max = sd['xyz_max'].float()
    geo = dvgo_geometry(model_kwargs, sd)
    thres = float(model_kwargs.get('fast_color_thres', 0))
    interval = stepsize * geo['voxel_size_ratio']                    # lib/dvgo.py:341
    far = 1e9                                                        # lib/dvgo.py:307
    stepdist = stepsize * geo['voxel_size']                          # lib/dvgo.py:310
    N_samples = int((geo['max_world_size'] - 1) / stepsize) + 1      # lib/dvgo.py:311

    pts, mask_outbbox, ray_id, step_id, _, _, _ = nat.sample_pts_on_rays(
        rays_o, rays_d, xyz_min, xyz_max, near, far, float(stepdist))
    n_total = pts.shape[0]
    mask_inbbox = ~mask_outbbox
    ray_pts, ray_id, step_id = pts[mask_inbbox], ray_id[mask_inbbox], step_id[mask_inbbox]
    n_inbbox = ray_pts.shape[0]

    mask1 = mask_grid(sd['mask_cache.mask'], ray_pts, sd['mask_cache.xyz2ijk_scale'].float(),
                      sd['mask_cache.xyz2ijk_shift'].float())
    ray_pts, ray_id, step_id = ray_pts[mask1], ray_id[mask1], step_id[mask1]
    n_mask = ray_pts.shape[0]

    # scalar act_shift BUFFER goes into Raw2Alpha (lib/dvgo.py:46, 276-279, 351-352)
    density = dense_grid(sd['density.grid'].float(), ray_pts, xyz_min, xyz_max)
    _, alpha = nat.raw2alpha(density.flatten(), float(sd['act_shift']), float(interval))
    if thres > 0:
        mask2 = alpha > thres
        ray_pts, ray_id, step_id, alpha = ray_pts[mask2], ray_id[mask2], step_id[mask2], alpha[mask2]
    n_alpha = ray_pts.shape[0]

    weights, _, alphainv_last, _, _ = nat.alpha2weight(alpha, ray_id, N)
    if thres > 0:
        mask3 = weights > thres
        ray_pts, ray_id, step_id = ray_pts[mask3], ray_id[mask3], step_id[mask3]
        alpha, weights = alpha[mask3], weights[mask3]
    n_shade = ray_pts.shape[0]

    # colour (lib/dvgo.py:372-412)
    k0 = dense_grid(sd['k0.grid'].float(), ray_pts, xyz_min, xyz_max)
    if k0.dim() == 1:
        k0 = k0.unsqueeze(-1)
    layers = _rgbnet_layers(sd)
    if layers is None:
        rgb_raw = torch.sigmoid(k0)
    else:
        direct = bool(model_kwargs.get('rgbnet_direct', False))
        k0_view = k0 if direct else k0[:, 3:]
        viewdirs_emb = _pe(viewdirs, sd['viewfreq'].float()).flatten(0, -2)[ray_id]
        rgb_logit = _mlp(layers, torch.cat([k0_view, viewdirs_emb], -1))
        rgb_raw = torch.sigmoid(rgb_logit if direct else rgb_logit + k0[:, :3])

    rgb_feature = _segment_sum(weights.unsqueeze(-1) * rgb_raw, ray_id, N)
    rgb_marched = rgb_feature
    rgb_marched += alphainv_last.unsqueeze(-1) * bg                   # lib/dvgo.py:427 (aliases)
    s = (step_id + 0.5) / N_samples
    ret = {
        'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
        'rgb_feature': rgb_feature, 'raw_alpha': alpha, 'raw_rgb': rgb_raw, 'ray_id': ray_id,
    }
    if render_depth:
        ret['depth'] = _segment_sum(weights * s, ray_id, N)
    if counters is not None:
        counters.update(n_rays=N, n_total=n_total, n_inbbox=n_inbbox, n_mask=n_mask,
                        n_alpha=n_alpha, n_shade=n_shade)
    return ret


def forward(model_class, model_kwargs, sd, rays_o, rays_d, viewdirs, chunk=8192, **render_kwargs):
    """Chunked evaluation as run_sr.py:121-124 does it (8192-ray chunks); returns the four keys
    the render loop consumes (run_sr.py:107), concatenated."""
    fn = {'DirectMPIGO': mpi_forward, 'DirectVoxGO': dvgo_forward}[model_class]
    keys = ['rgb_marched', 'depth', 'alphainv_last', 'rgb_feature']
    outs = []
    cnt_total = {}
    for ro, rd, vd in zip(rays_o.split(chunk, 0), rays_d.split(chunk, 0), viewdirs.split(chunk, 0)):
        cnt = {}
        r = fn(model_kwargs, sd, ro, rd, vd, counters=cnt, **render_kwargs)
        outs.append({k: r[k] for k in keys if k in r})
        for k, v in cnt.items():
            cnt_total[k] = cnt_total.get(k, 0) + v
    ret = {k: torch.cat([o[k] for o in outs]) for k in outs[0]}
    ret['counters'] = cnt_total
    return ret


def algorithmic_bytes(counters, k0_ch):
    """B_alg = N_rays*56 + S_inbbox*1 + S_mask*32 + S_shade*(8*C_k0*4)."""
    return (counters['n_rays'] * 56 + counters['n_inbbox'] * 1 + counters['n_mask'] * 32
            + counters['n_shade'] * 8 * k0_ch * 4)
