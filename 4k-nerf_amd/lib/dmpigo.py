"""DirectMPIGO (the LLFF / NDC model) with the reference's interface (/root/reference/lib/dmpigo.py).

Constructor kwargs, ``get_kwargs()``, ``state_dict`` key names (``density.grid``, ``k0.grid``,
``act_shift.grid`` [1,1,1,1,D], ``rgbnet.*``, ``viewfreq``, ``posfreq``, ``mask_cache.*``) and
``forward(rays_o, rays_d, viewdirs, global_step=None, **render_kwargs) -> dict`` are the reference's
(lib/dmpigo.py:18-187,292-427).  Inside, inference is ONE fused HIP launch (``k4_march_mpi_fwd``);
``k4_staged=True`` / autograd / unsupported MLP shapes take the staged gfx950 kernels and return every
key of the reference dict.  No CPU path.

Reference quirks honoured: ``act_type`` / ``mode_type`` are required kwargs when
``rgbnet_dim>0`` (lib/dmpigo.py:89,124); ``rgb_feature`` aliases ``rgb_marched`` in eval
(lib/dmpigo.py:392-397); ``mode_type`` 'TRANS'/'adain' reference undefined modules upstream and are
rejected here.
"""

import numpy as np
import torch
import torch.nn as nn

from .. import _native as N
from . import grid
from . import dvgo as _dvgo
from . import train_ops

_TRAIN_PRESEL = True
_FUSED_MLP_INPUT = True       # False: the colour MLP's input of the training forward built op by op as the reference does (A/B, tests)
DEPTH_SPLIT = True              # fused marcher: depth-ordered geometry stage where the scene's density says it pays (_k4_depth_split); False: always one launch
DEPTH_SPLIT_MIN_GAIN = 0.05     # smallest share of alpha-passing voxels behind the split in stopped columns for which the geometry stage is cut in two launches
                                # (measured, profiles/r06_depth_split.md: the cut itself costs 0-2 % of the call at 128 / 192 of 256 samples, +4 % at 64; the opaque scene gains 10 %)
DEPTH_SPLIT_MIN_OPAQUE = 0.5    # ... and smallest share of occupied z columns along which a ray reaches the T < 1e-3 stop (bench scene: 0.16 -> one launch; opaque wall: 1.0)      # training forward: the three sample filters decided by one launch (same values)
from .dvgo import Raw2Alpha, Alphas2Weights, render_utils_cuda, _FusedMarcher, segment_sum, coarse_mask_on_grid, _take


'''Model'''
def _mlp(dim_in, width, depth, dim_out):
    """Linear-ReLU stack with the reference's module nesting (state-dict keys '0', '2.0', ..., lib/dmpigo.py:112-120); the last
    bias starts at 0."""
    act = nn.ReLU(inplace=True)
    hidden = [nn.Sequential(nn.Linear(width, width), act) for _ in range(depth - 2)]
    net = nn.Sequential(nn.Linear(dim_in, width), act, *hidden, nn.Linear(width, dim_out))
    nn.init.constant_(net[-1].bias, 0)
    return net


class DirectMPIGO(torch.nn.Module, _FusedMarcher):
    def __init__(self, xyz_min, xyz_max,
                 num_voxels=0, mpi_depth=0,
                 mask_cache_path=None, mask_cache_thres=1e-3, mask_cache_world_size=None,
                 fast_color_thres=0,
                 density_type='DenseGrid', k0_type='DenseGrid',
                 density_config={}, k0_config={},
                 rgbnet_dim=0,
                 rgbnet_depth=3, rgbnet_width=128,
                 viewbase_pe=0,
                 spatial_pe=0,
                 **kwargs):
        super().__init__()
        for name, val in (('xyz_min', xyz_min), ('xyz_max', xyz_max)):
            self.register_buffer(name, torch.Tensor(val))
        self.fast_color_thres = fast_color_thres
        self._set_grid_resolution(num_voxels, mpi_depth)
        self.density_type, self.density_config = density_type, density_config
        self.k0_type, self.k0_config = k0_type, k0_config
        self.rgbnet_kwargs = dict(rgbnet_dim=rgbnet_dim, rgbnet_depth=rgbnet_depth, rgbnet_width=rgbnet_width,
                                  viewbase_pe=viewbase_pe, spatial_pe=spatial_pe)
        self.density = self._new_grid(density_type, 1, density_config)
        self._init_act_shift(xyz_min, xyz_max)

        # colour: feature grid + MLP (lib/dmpigo.py:65-131).  The coarse "colour voxel" branch (rgbnet_dim <= 0) is kept for
        # state-dict compatibility although upstream cannot run it (forward reads self.dim_rend, set only when rgbnet_dim > 0).
        self.dim_rend = 3
        self.act_type, self.mode_type = kwargs.get('act_type', 'relu'), kwargs.get('mode_type', 'mlp')
        self.k0_dim = rgbnet_dim if rgbnet_dim > 0 else 3
        self.k0 = self._new_grid(k0_type, self.k0_dim, k0_config)
        self.rgbnet = None
        if rgbnet_dim > 0:
            self.act_type, self.mode_type = kwargs['act_type'], kwargs['mode_type']      # required upstream (lib/dmpigo.py:89)
            if self.act_type != 'relu':
                raise NotImplementedError(f"act_type={self.act_type!r}: every BASELINE config uses 'relu' "
                                          "(configs/llff/fern_lg_joint_l1.py)")
            if self.mode_type in ('TRANS', 'adain'):
                raise NotImplementedError(f'mode_type={self.mode_type!r} needs modules the reference never defines '
                                          '(lib/dmpigo.py:124-130)')
            for name, n in (('viewfreq', viewbase_pe), ('posfreq', spatial_pe)):
                self.register_buffer(name, torch.FloatTensor([2 ** i for i in range(n)]))
            self.pe_dim = 3 * (1 + 2 * viewbase_pe) + 3 * (1 + 2 * spatial_pe)           # viewdirs + position embeddings
            self.dim0 = self.pe_dim + self.k0_dim
            self.rgbnet = _mlp(self.dim0, rgbnet_width, rgbnet_depth, self.dim_rend)

        # occupancy: from a coarse checkpoint when given, else everything occupied (lib/dmpigo.py:133-152)
        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, mask_cache_thres
        mask_ws = self.world_size if mask_cache_world_size is None else mask_cache_world_size
        if mask_cache_path:
            mask = coarse_mask_on_grid(mask_cache_path, mask_cache_thres, self.xyz_min, self.xyz_max, mask_ws)
        else:
            mask = torch.ones(list(mask_ws), dtype=torch.bool)
        self.mask_cache = grid.MaskGrid(path=None, mask=mask, xyz_min=self.xyz_min, xyz_max=self.xyz_max)

    def _new_grid(self, kind, channels, config):
        return grid.create_grid(kind, channels=channels, world_size=self.world_size, xyz_min=self.xyz_min,
                                xyz_max=self.xyz_max, config=config)

    def _init_act_shift(self, xyz_min, xyz_max):
        """Per-plane density bias chosen so that a ray through an empty volume sees equal alphas on every plane
        (lib/dmpigo.py:48-58): alpha_i = g = 1/D - 1e-6 of the remaining transmittance."""
        D = self.mpi_depth
        self.act_shift = grid.DenseGrid(channels=1, world_size=[1, 1, D], xyz_min=xyz_min, xyz_max=xyz_max)
        self.act_shift.grid.requires_grad = False
        g = np.full([D], 1. / D - 1e-6)
        keep = [1 - g[0]] + [(1 - g[:i + 1].sum()) / (1 - g[:i].sum()) for i in range(1, D)]
        with torch.no_grad():
            for i, p_i in enumerate(keep):
                self.act_shift.grid[..., i].fill_(np.log(p_i ** (-1 / self.voxel_size_ratio) - 1))

    def _set_grid_resolution(self, num_voxels, mpi_depth):
        # lib/dmpigo.py:156-164
        self.num_voxels = num_voxels
        self.mpi_depth = mpi_depth
        r = (num_voxels / self.mpi_depth / (self.xyz_max - self.xyz_min)[:2].prod()).sqrt()
        self.world_size = torch.zeros(3, dtype=torch.long)
        self.world_size[:2] = (self.xyz_max - self.xyz_min)[:2] * r
        self.world_size[2] = self.mpi_depth
        self.voxel_size_ratio = 256. / mpi_depth

    def get_kwargs(self):
        return {
            'xyz_min': self.xyz_min.cpu().numpy(),
            'xyz_max': self.xyz_max.cpu().numpy(),
            'num_voxels': self.num_voxels,
            'mpi_depth': self.mpi_depth,
            'voxel_size_ratio': self.voxel_size_ratio,
            'mask_cache_path': self.mask_cache_path,
            'mask_cache_thres': self.mask_cache_thres,
            'mask_cache_world_size': list(self.mask_cache.mask.shape),
            'fast_color_thres': self.fast_color_thres,
            'density_type': self.density_type,
            'k0_type': self.k0_type,
            'density_config': self.density_config,
            'k0_config': self.k0_config,
            'mode_type': self.mode_type,
            'act_type': self.act_type,
            'dim_rend': self.dim_rend,
            **self.rgbnet_kwargs,
        }

    # ------------------------------------------------------------------ resolution / occupancy maintenance (training loop)
    @torch.no_grad()
    def scale_volume_grid(self, num_voxels, mpi_depth):
        """Progressive growing (lib/dmpigo.py:189-212): resample density / k0 to the new resolution (k4_resample_trilinear) and, while
        the grid is small enough (<= 256^3), refresh the occupancy: old mask looked up at the new nodes AND max-pooled alpha > thres."""
        self._set_grid_resolution(num_voxels, mpi_depth)
        self.density.scale_volume_grid(self.world_size)
        self.k0.scale_volume_grid(self.world_size)
        if int(np.prod(self.world_size.tolist())) <= 256 ** 3:
            nodes = grid.grid_nodes(self.xyz_min, self.xyz_max, self.world_size.tolist())
            dens = self.density.get_dense_grid() + self.act_shift.grid        # [1,1,X,Y,Z] + [1,1,1,1,D] (D == Z here)
            occupied = grid.occupancy_from_alpha(self.activate_density(dens)[0, 0], self.fast_color_thres)
            self.mask_cache = grid.MaskGrid(path=None, mask=self.mask_cache(nodes) & occupied,
                                            xyz_min=self.xyz_min, xyz_max=self.xyz_max).to(nodes.device)

    @torch.no_grad()
    def update_occupancy_cache(self):
        """lib/dmpigo.py:214-226: mask &= maxpool3(alpha(density at the mask's nodes)) > fast_color_thres.  The density lookup
        (k4_grid_sample_3d), the activation (k4_raw2alpha) and the pooled threshold (k4_alpha_maxpool3_gt) are HIP kernels; the mask
        tensor is updated in place, its version bump re-keys the fused marcher's occupancy summary."""
        nodes = grid.grid_nodes(self.xyz_min, self.xyz_max, list(self.mask_cache.mask.shape))
        alpha = self.activate_density(self.density(nodes))
        self.mask_cache.mask &= grid.occupancy_from_alpha(alpha, self.fast_color_thres)

    def update_occupancy_cache_lt_nviews(self, rays_o_tr, rays_d_tr, imsz, render_kwargs, maskout_lt_nviews):
        """lib/dmpigo.py:228-246: drop voxels seen by fewer than `maskout_lt_nviews` training views.  A view "sees" a voxel when the
        gradient of sum(ones(ray_pts)) w.r.t. an all-ones grid exceeds 1 there (k4_grid_sample_3d_backward scatter)."""
        count = torch.zeros_like(self.density.get_dense_grid()).long()
        dev = count.device
        for rays_o_, rays_d_ in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
            ones = grid.DenseGrid(1, self.world_size, self.xyz_min, self.xyz_max).to(dev)
            for rays_o, rays_d in zip(rays_o_.split(8192), rays_d_.split(8192)):
                ray_pts = self.sample_ray(rays_o=rays_o.to(dev), rays_d=rays_d.to(dev), **render_kwargs)[0]
                with torch.enable_grad():
                    ones(ray_pts).sum().backward()
            count += (ones.grid.grad > 1)
        with torch.no_grad():
            self.mask_cache.mask &= (count >= maskout_lt_nviews)[0, 0]

    def density_total_variation_add_grad(self, weight, dense_mode):
        '''lib/dmpigo.py:248-251: separate in-plane / depth weights (the reference passes them as wx=wy=wxy, wz).'''
        wxy = weight * self.world_size[:2].max() / 128
        wz = weight * self.mpi_depth / 128
        self.density.total_variation_add_grad(wxy, wxy, wz, dense_mode)

    def k0_total_variation_add_grad(self, weight, dense_mode):
        '''lib/dmpigo.py:253-256.'''
        wxy = weight * self.world_size[:2].max() / 128
        wz = weight * self.mpi_depth / 128
        self.k0.total_variation_add_grad(wxy, wxy, wz, dense_mode)

    def activate_density(self, density, interval=None):
        interval = interval if interval is not None else self.voxel_size_ratio
        shape = density.shape
        return Raw2Alpha.apply(density.flatten(), 0, interval).reshape(shape)

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        '''Sample query points on rays (lib/dmpigo.py:263-290).'''
        assert near == 0 and far == 1
        rays_o = rays_o.contiguous()
        rays_d = rays_d.contiguous()
        N_samples = int((self.mpi_depth - 1) / stepsize) + 1
        ray_pts, mask_outbbox = render_utils_cuda.sample_ndc_pts_on_rays(
            rays_o, rays_d, self.xyz_min, self.xyz_max, N_samples)
        mask_inbbox = ~mask_outbbox
        # ONE compaction index for the three gathers (each boolean-mask indexing is a nonzero + a host synchronisation of its own)
        idx = mask_inbbox.view(-1).nonzero().squeeze(1)
        ray_pts = ray_pts.view(-1, 3).index_select(0, idx)
        ray_id = torch.div(idx, mask_inbbox.shape[1], rounding_mode='floor')        # = arange(N).expand_as(mask)[mask]
        step_id = idx - ray_id * mask_inbbox.shape[1]                               # = arange(N_samples).expand_as(mask)[mask]
        return ray_pts, ray_id, step_id, N_samples, mask_inbbox

    # ------------------------------------------------------------------ forward
    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        '''Volume rendering
        @rays_o:   [N, 3] the starting point of the N shooting rays.
        @rays_d:   [N, 3] the shooting direction of the N rays.
        @viewdirs: [N, 3] viewing direction to compute positional embedding for MLP.
        '''
        rays_o, rays_d, viewdirs = self._k4_check_rays(rays_o, rays_d, viewdirs)
        staged = render_kwargs.get('k4_staged', False) or torch.is_grad_enabled() or not self._k4_fusable()
        if staged:
            return self._forward_staged(rays_o, rays_d, viewdirs, global_step=global_step, **render_kwargs)
        self._k4_params_ready()
        return self._forward_fused(rays_o, rays_d, viewdirs, **render_kwargs)

    def _forward_fused(self, rays_o, rays_d, viewdirs, near, far, stepsize, bg, render_depth=False,
                       k4_img_w=0, k4_counters=None, k4_out=None, k4_ws_slot=0, k4_live_mask=True, **_ignored):
        assert near == 0 and far == 1                                     # lib/dmpigo.py:275
        Nr = rays_o.shape[0]
        dev = rays_o.device
        if k4_out is not None:                  # caller-provided outputs (e.g. slices of an all-gather send buffer)
            rgb, depth, ainv = k4_out
            assert rgb.shape == (Nr, 3) and depth.shape == (Nr,) and ainv.shape == (Nr,)
        else:
            rgb = torch.empty([Nr, 3], dtype=torch.float32, device=dev)
            depth = torch.empty([Nr], dtype=torch.float32, device=dev)
            ainv = torch.empty([Nr], dtype=torch.float32, device=dev)
        use_live = bool(k4_live_mask and (k4_counters is None or k4_live_mask == 'force'))      # ('force': profiling builds that stamp the render instantiation)

        def build():
            md, keep = self._k4_mlp(k0_skip=0, spatial_pe=len(self.posfreq) if self.rgbnet is not None else 0)
            n_samples = int((self.mpi_depth - 1) / stepsize) + 1          # lib/dmpigo.py:278
            itv = float(stepsize * self.voxel_size_ratio)                 # lib/dmpigo.py:306
            # sample counters are the ALGORITHM's counts: the counting pass looks samples up in mask_cache itself
            gd = self._k4_grid(act_shift_grid=self.act_shift.grid, live=(0.0, itv) if use_live else None)
            gd.depth_split = self._k4_depth_split(gd, n_samples, itv) if (use_live and DEPTH_SPLIT) else 0
            return (md, gd, n_samples, itv), keep
        md, gd, N_samples, interval = self._k4_plan('mpi', (float(stepsize), use_live, float(self.fast_color_thres), DEPTH_SPLIT, DEPTH_SPLIT_MIN_GAIN, DEPTH_SPLIT_MIN_OPAQUE), build)
        if Nr > 0:
            ws, ws_bytes = self._k4_workspace(Nr, k4_img_w, N_samples, dev, k4_ws_slot)
            N.check(N.lib().k4_march_mpi_fwd(
                N.f32(rays_o), N.f32(rays_d), N.f32(viewdirs), Nr, int(k4_img_w), N.C.byref(gd), N.C.byref(md),
                N_samples, interval, float(self.fast_color_thres), float(bg), N.ptr(ws), ws_bytes,
                N.f32(rgb), N.f32(depth), N.f32(ainv),
                None if k4_counters is None else N.ptr(k4_counters), N.stream()), 'k4_march_mpi_fwd')
        ret = {'alphainv_last': ainv, 'rgb_marched': rgb, 'rgb_feature': rgb, 'n_max': N_samples}
        if render_depth:
            ret['depth'] = depth
        return ret

    def _k4_depth_split(self, gd, n_samples, interval):
        """k4_grid_desc.depth_split of this scene (include/k4nerf.h): the sample index at which the fused marcher's geometry stage is cut in a
        front and a back launch, the back one skipping every ray the front one's transmittance scan stopped -- Alphas2Weights' early stop
        (render_utils_kernel.cu:597-600) exploited in the density stage instead of after it, as the reference does (lib/dmpigo.py:316-333
        evaluates density for every mask-passing sample and drops the ones behind the stop afterwards).  Chosen from the density grid at load
        time (k4_mpi_depth_split_stats: per z column the plane where a ray along the column stops; ONE 16-float read-back per density
        version): the multiple of 64 samples behind which the largest share of alpha-passing voxels sits in columns already stopped, if that
        share is at least DEPTH_SPLIT_MIN_GAIN and at least DEPTH_SPLIT_MIN_OPAQUE of the occupied columns stop a ray at all (a scene of opaque
        surfaces, as a trained LLFF scene is) -- translucent scenes keep the single launch (0).  Identical outputs either way (tests)."""
        dens, act = self.density.grid, self.act_shift.grid
        key = ('dsplit', dens.data_ptr(), dens._version, act.data_ptr(), act._version, float(interval), float(self.fast_color_thres), int(n_samples))
        c = self._k4_cache()
        if c.get('dsplit_key') != key:
            out = torch.empty([16], dtype=torch.float32, device=dens.device)
            N.check(N.lib().k4_mpi_depth_split_stats(N.C.byref(gd), float(interval), float(self.fast_color_thres), N.f32(out), N.stream()),
                    'k4_mpi_depth_split_stats')
            st = out.cpu().tolist()
            best, gain = 0, 0.0
            for k in range(64, int(n_samples), 64):
                b = min(7, (k * 8) // int(n_samples))
                if b >= 1 and st[b] > gain:
                    best, gain = k, st[b]
            use = gain >= DEPTH_SPLIT_MIN_GAIN and st[0] >= DEPTH_SPLIT_MIN_OPAQUE
            c['dsplit_key'], c['dsplit'], c['dsplit_stats'] = key, (best if use else 0), st[:8]
        return int(c['dsplit'])

    def _select_samples(self, rays_o, rays_d, N_samples, interval):
        """The three sample filters of lib/dmpigo.py:300-333 (bounding box + mask cache, alpha > thres, weight > thres) decided for the whole
        batch by ONE launch (k4_train_select_mpi: the staged ops' arithmetic, so its decisions are theirs) + a compaction; ONE read-back (the
        two list lengths) instead of one per boolean-mask indexing.  -> (ray_pts, ray_id, step_id) of the alpha-passing samples and idx3, the
        positions of the shaded ones in that list."""
        Nr, dev = rays_o.shape[0], rays_o.device
        L = N.lib()
        mc, dg, ag = self.mask_cache, self.density.grid, self.act_shift.grid
        self.density.params_ready()
        steps2 = torch.empty([Nr, N_samples], dtype=torch.int16, device=dev)
        keep3 = torch.empty([Nr, N_samples], dtype=torch.uint8, device=dev)
        cnt = torch.empty([2, Nr], dtype=torch.int64, device=dev)
        st = N.stream()
        N.check(L.k4_train_select_mpi(N.f32(rays_o), N.f32(rays_d), N.f32(self.xyz_min), N.f32(self.xyz_max), Nr, int(N_samples),
                                      N.ptr(mc.mask), N.f32(mc.xyz2ijk_scale), N.f32(mc.xyz2ijk_shift), *[int(v) for v in mc.mask.shape],
                                      N.f32(dg), *[int(v) for v in dg.shape[2:]], N.f32(ag), int(ag.numel()),
                                      float(interval), float(self.fast_color_thres),
                                      N.ptr(steps2), N.ptr(keep3), N.ptr(cnt[0]), N.ptr(cnt[1]), st), 'k4_train_select_mpi')
        cum = cnt.cumsum(1)
        n2, n3 = (int(v) for v in cum[:, -1].tolist()) if Nr > 0 else (0, 0)            # the iteration's one device-to-host read in the marcher forward
        # (+1: empty lists still hand the kernels a valid pointer; a zero-element view reports a NULL data pointer)
        ray_id_b = torch.empty([n2 + 1], dtype=torch.int64, device=dev)
        step_id_b = torch.empty([n2 + 1], dtype=torch.int64, device=dev)
        idx3_b = torch.empty([n3 + 1], dtype=torch.int64, device=dev)
        N.check(L.k4_train_compact(N.ptr(steps2), N.ptr(keep3), N.ptr(cnt[0]), N.ptr(cum[0]), N.ptr(cum[1]), Nr, int(N_samples),
                                   N.ptr(ray_id_b), N.ptr(step_id_b), N.ptr(idx3_b), st), 'k4_train_compact')
        ray_pts_b = torch.empty([n2 + 1, 3], dtype=torch.float32, device=dev)
        N.check(L.k4_ndc_points_of(N.f32(rays_o), N.f32(rays_d), N.ptr(ray_id_b), N.ptr(step_id_b), n2, int(N_samples), N.f32(ray_pts_b), st), 'k4_ndc_points_of')
        ray_pts, ray_id, step_id, idx3 = ray_pts_b[:n2], ray_id_b[:n2], step_id_b[:n2], idx3_b[:n3]
        return ray_pts, ray_id, step_id, idx3

    def _forward_staged(self, rays_o, rays_d, viewdirs, near, far, stepsize, bg, render_depth=False,
                        global_step=None, rand_bkgd=False, k4_presel=None, **_ignored):
        """The reference's op sequence (lib/dmpigo.py:300-427) on the staged gfx950 kernels.  With a mask cache and fast_color_thres > 0 (every
        BASELINE configuration) the three sample filters are decided up front by ``_select_samples``: the differentiable ops then run on the
        same lists the op-for-op sequence ends with (same values, same gradients), with one host synchronisation instead of four.
        ``k4_presel=False`` keeps the filter-by-filter form (A/B, tests)."""
        ret_dict = {}
        Nr = len(rays_o)
        interval = stepsize * self.voxel_size_ratio
        presel = (_TRAIN_PRESEL if k4_presel is None else bool(k4_presel)) and self.mask_cache is not None and self.fast_color_thres > 0
        if presel:
            assert near == 0 and far == 1
            N_samples = int((self.mpi_depth - 1) / stepsize) + 1
            presel = N_samples <= 32767
        if presel:
            ray_pts, ray_id, step_id, idx3 = self._select_samples(rays_o.contiguous(), rays_d.contiguous(), N_samples, interval)
            density = self.density(ray_pts) + self.act_shift(ray_pts)
            alpha = self.activate_density(density, interval)
            weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, Nr)
            ray_pts, ray_id, step_id, alpha, weights = [t.index_select(0, idx3) for t in (ray_pts, ray_id, step_id, alpha, weights)]
        else:
            ray_pts, ray_id, step_id, N_samples, mask_inbbox = self.sample_ray(
                rays_o=rays_o, rays_d=rays_d, near=near, far=far, stepsize=stepsize)
            if self.mask_cache is not None:
                mask1 = self.mask_cache(ray_pts)
                ray_pts, ray_id, step_id = _take(mask1, ray_pts, ray_id, step_id)
            density = self.density(ray_pts) + self.act_shift(ray_pts)
            alpha = self.activate_density(density, interval)
            if self.fast_color_thres > 0:
                mask2 = (alpha > self.fast_color_thres)
                ray_pts, ray_id, step_id, alpha = _take(mask2, ray_pts, ray_id, step_id, alpha)
            weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, Nr)
            if self.fast_color_thres > 0:
                mask3 = (weights > self.fast_color_thres)
                ray_pts, ray_id, step_id, alpha, weights = _take(mask3, ray_pts, ray_id, step_id, alpha, weights)
        vox_emb = self.k0(ray_pts)
        if vox_emb.dim() == 1:
            vox_emb = vox_emb.unsqueeze(-1)
        fused_in = None
        if self.rgbnet is not None and _FUSED_MLP_INPUT:                  # the 16 ops below in one launch (lib/train_ops.RgbnetInputMPI: same values)
            fused_in = train_ops.rgbnet_input_mpi(vox_emb, ray_pts, viewdirs, ray_id, self.xyz_min, self.xyz_max, self.posfreq, self.viewfreq)
        if fused_in is not None:
            rgb_raw = self._k4_rgbnet_sigmoid(fused_in)
        elif self.rgbnet is None:
            rgb_raw = torch.sigmoid(vox_emb)
        else:
            pe_spa = ((ray_pts - self.xyz_min) / (self.xyz_max - self.xyz_min)).flip((-1,)) * 2 - 1
            viewdirs_emb = (viewdirs.unsqueeze(-1) * self.viewfreq).flatten(-2)
            viewdirs_emb = torch.cat([viewdirs, viewdirs_emb.sin(), viewdirs_emb.cos()], -1)
            viewdirs_emb = viewdirs_emb[ray_id]
            pe_emb = (pe_spa.unsqueeze(-1) * self.posfreq).flatten(-2)
            pe_emb = torch.cat([pe_spa, pe_emb.sin(), pe_emb.cos()], -1)
            rgb_raw = self._k4_rgbnet_sigmoid(torch.cat([vox_emb, pe_emb, viewdirs_emb], -1))
        rgb_feature = segment_sum(weights.unsqueeze(-1) * rgb_raw, ray_id, Nr)
        rgb_marched = rgb_feature
        if rand_bkgd and global_step is not None:
            rgb_marched = rgb_marched + (alphainv_last.unsqueeze(-1) * torch.rand_like(rgb_marched))
        else:
            rgb_marched += (alphainv_last.unsqueeze(-1) * bg)             # aliases rgb_feature (lib/dmpigo.py:397)
        s = (step_id + 0.5) / N_samples
        ret_dict.update({
            'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
            'rgb_feature': rgb_feature, 'raw_alpha': alpha, 'raw_rgb': rgb_raw, 'ray_id': ray_id,
            'n_max': N_samples, 's': s,
        })
        if render_depth:
            with torch.no_grad():
                ret_dict['depth'] = segment_sum(weights * s, ray_id, Nr)
        return ret_dict
