"""DirectVoxGO with the reference's interface (/root/reference/lib/dvgo.py), MI355X-native inside.

Kept verbatim from the reference contract: constructor kwargs, ``get_kwargs()``
round trip, ``state_dict`` key names (``density.grid``, ``k0.grid``, ``act_shift``, ``rgbnet.*``,
``mask_cache.*``, ``viewfreq``), ``forward(rays_o, rays_d, viewdirs, global_step=None,
**render_kwargs) -> dict`` and the module-level names other reference code imports from here
(``render_utils_cuda``, ``Raw2Alpha``, ``Alphas2Weights``, ``get_rays*``, ``ndc_rays``).

What is new is everything inside ``forward``:
  * inference (``torch.no_grad``): ONE fused HIP launch (``k4_march_dvgo_fwd``, csrc/k4_march.hip)
    returning the keys the render loop consumes (run_sr.py:107): ``rgb_marched`` (is ``rgb_feature``,
    they alias in the reference too, lib/dvgo.py:425-427), ``depth``, ``alphainv_last``;
  * ``render_kwargs['k4_staged']=True`` or autograd: the staged path = the reference's op sequence
    on the gfx950 staged kernels, returning every key of the reference dict (``weights``,
    ``raw_alpha``, ``raw_rgb``, ``ray_id``).
There is no CPU path: CPU tensors raise.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as N
from . import grid
from . import render_utils_cuda
from . import train_ops

_FUSED_WIDTHS = (32, 64, 128)
_LIVE_MASK = os.environ.get('K4_LIVE_MASK', '1') != '0'      # A/B switch of the density-derived live mask (identical results)


class _FusedMarcher:
    """Mixin: device-side descriptors for the fused kernels, cached per parameter version."""

    def _k4_cache(self):
        if not hasattr(self, '_k4c'):
            object.__setattr__(self, '_k4c', {})
        return self._k4c

    def _k4_fusable(self):
        c = self._k4_cache()
        hit = c.get('fusable')
        if hit is None or hit[0] is not self.rgbnet:
            hit = c['fusable'] = (self.rgbnet, self._k4_fusable_uncached())
        return hit[1]

    def _k4_fusable_uncached(self):
        if self.rgbnet is None:
            return self.k0_dim == 3
        lins = [m for m in self.rgbnet.modules() if isinstance(m, nn.Linear)]
        acts_ok = all(isinstance(m, (nn.Linear, nn.ReLU, nn.Sequential)) for m in self.rgbnet.modules())
        return (acts_ok and len(lins) in (2, 3) and lins[0].out_features in _FUSED_WIDTHS
                and lins[-1].out_features == 3 and getattr(self, 'mode_type', 'mlp') not in ('TRANS', 'adain')
                and not getattr(self, 'rgbnet_full_implicit', False))

    def _k4_k0_channel_last(self):
        """Load-time repack of k0.grid [1,C,X,Y,Z] -> [X,Y,Z,CP] (CP = C rounded up to 4): the 8 corners of a
        shaded sample become 4 runs of 2*CP contiguous floats instead of 8*C scattered dwords.
        Returns (repacked grid, CP, k4_grid_desc.k0_layout)."""
        g = self.k0.grid
        key = ('k0', g.data_ptr(), g._version, str(g.device))
        c = self._k4_cache()
        if c.get('k0_key') != key:
            C = g.shape[1]
            CP = (C + 3) // 4 * 4
            X, Y, Z = (int(v) for v in g.shape[2:])
            nvox = X * Y * Z
            src = g.detach().contiguous()
            out = torch.empty([nvox * CP], dtype=torch.float32, device=g.device)
            N.check(N.lib().k4_repack_k0(N.f32(src), C, CP, nvox, N.f32(out), N.stream()), 'repack_k0')
            c['k0_key'], c['k0_cl'], c['k0_cpad'] = key, out, CP
        return c['k0_cl'], c['k0_cpad'], N.K0_CHANNEL_LAST

    def _k4_occ_summary(self, m=None, slot='occ'):
        """Load-time coarse occupancy summary of a mask (k4_build_occupancy_summary), cached per mask version: lets the
        geometry kernel skip 16-sample groups of a ray that cannot touch an occupied voxel (identical results)."""
        m = self.mask_cache.mask if m is None else m
        key = (slot, m.data_ptr(), m._version, str(m.device))
        c = self._k4_cache()
        if c.get(slot + '_key') != key:
            mx, my, mz = (int(v) for v in m.shape)
            nbytes = int(N.lib().k4_occupancy_summary_bytes(mx, my, mz))
            out = torch.empty([nbytes // 4], dtype=torch.int32, device=m.device)
            N.check(N.lib().k4_build_occupancy_summary(N.ptr(m), mx, my, mz, N.ptr(out), N.stream()), 'k4_build_occupancy_summary')
            c[slot + '_key'], c[slot] = key, out
        return c[slot]

    def _k4_live_mask(self, gd, act_shift_grid, shift, interval):
        """mask_cache.mask AND "some density cell a sample of this voxel can lie in reaches alpha > fast_color_thres"
        (k4_build_live_mask): the geometry kernel looks samples up in THIS mask, so the density stage no longer runs on samples that
        the reference drops one step later at lib/dmpigo.py:319-323 / lib/dvgo.py:356-360.  Bit-identical outputs; cached per
        density / act_shift / mask version and per (shift, interval, threshold).  Returns (live mask, its coarse summary)."""
        dens, m = self.density.grid, self.mask_cache.mask
        key = ('live', dens.data_ptr(), dens._version, m.data_ptr(), m._version, str(m.device), float(shift), float(interval),
               float(self.fast_color_thres)) + ((act_shift_grid.data_ptr(), act_shift_grid._version) if act_shift_grid is not None else ())
        c = self._k4_cache()
        if c.get('live_key') != key:
            X, Y, Z = (int(v) for v in dens.shape[2:])
            ws = torch.empty([int(N.lib().k4_live_mask_workspace_bytes(X, Y, Z))], dtype=torch.uint8, device=m.device)
            out = torch.empty(list(m.shape), dtype=torch.uint8, device=m.device)
            N.check(N.lib().k4_build_live_mask(N.C.byref(gd), float(shift), float(interval), float(self.fast_color_thres),
                                               N.ptr(ws), N.ptr(out), N.stream()), 'k4_build_live_mask')
            occ = self._k4_occ_summary(out, slot='occ_live')
            ev = torch.cuda.Event()
            ev.record()
            c['live_key'], c['live'], c['live_ev'], c['live_seen'] = key, (out, occ), ev, {N.stream().value}
        elif N.stream().value not in c['live_seen']:
            # first use on another HIP stream: order it behind the build (streams forked before the build would race with it)
            torch.cuda.current_stream().wait_event(c['live_ev'])
            c['live_seen'].add(N.stream().value)
        return c['live']

    def _k4_mlp(self, k0_skip, spatial_pe):
        md = N.MlpDesc()
        md.viewbase_pe = int(len(self.viewfreq)) if self.rgbnet is not None else 0
        md.spatial_pe = int(spatial_pe)
        md.k0_skip = int(k0_skip)
        md.arith = {'fp32': 1, 'b3': 2}.get(os.environ.get('K4_MLP', ''), 0)      # K4_MLP_ARITH_*: default 'b2' (2-term hidden activations); 'b3' exact 3-term; 'fp32' fp32-input MFMA (tests, A/B runs)
        if self.rgbnet is None:
            md.packed, md.dim0, md.width, md.n_hidden = None, 0, 0, 0
            return md, None
        lins = [m for m in self.rgbnet.modules() if isinstance(m, nn.Linear)]
        key = ('mlp',) + tuple((l.weight.data_ptr(), l.weight._version, l.bias._version) for l in lins)
        c = self._k4_cache()
        if c.get('mlp_key') != key:
            c['mlp_key'], c['mlp_packed'] = key, pack_mlp_mfma(lins)
        md.packed = c['mlp_packed'].data_ptr()
        md.dim0 = lins[0].in_features
        md.width = lins[0].out_features
        md.n_hidden = len(lins) - 2
        return md, c['mlp_packed']

    def _k4_rgbnet_sigmoid(self, feat, add=None):
        """``torch.sigmoid(self.rgbnet(feat) [+ add])`` of the staged / training forward (lib/dmpigo.py:375-379, lib/dvgo.py:407-412)
        on k4_rgbnet_fwd / k4_rgbnet_bwd (lib/train_ops.py); Linear-ReLU stacks outside those kernels' shapes (deeper, other widths) run
        layer by layer on the exact-fp32 MFMA 1x1 convolution (inference only).  There is no PyTorch path: anything else raises."""
        c = self._k4_cache()
        if 'rgbnet_native' not in c:
            c['rgbnet_native'] = train_ops.rgbnet_supported(self.rgbnet)
        if c['rgbnet_native']:
            return train_ops.rgbnet_sigmoid(self.rgbnet, feat, add)
        return train_ops.rgbnet_sigmoid_layers(self.rgbnet, feat, add)

    def _k4_workspace(self, n_rays, img_w, max_steps, device, slot=0):
        """Scratch between the geometry and the shading kernel: worst-case sized (every sample of every ray
        shaded), sparsely touched, cached and grown on demand.  One per ``slot``: calls that may be in flight
        concurrently (different HIP streams) must use different slots (render_kwargs['k4_ws_slot'])."""
        c = self._k4_cache()
        nkey = ('ws_need', int(n_rays), int(img_w), int(max_steps))
        need = c.get(nkey)
        if need is None:
            need = int(N.lib().k4_march_workspace_bytes(int(n_rays), int(img_w), int(max_steps)))
            if need < 0:
                raise N.K4Error('k4_march_workspace_bytes: bad arguments')
            c[nkey] = need
        ws = c.get(('workspace', slot))
        if ws is None or ws.numel() < need or ws.device != device:
            ws = None
            c[('workspace', slot)] = None
            ws = torch.empty([max(need, 256)], dtype=torch.uint8, device=device)
            c[('workspace', slot)] = ws
        return ws, need

    def _k4_params_ready(self):
        """A grid whose optimizer step runs on a second stream (MaskedAdam.update_on_side_stream): the current stream waits for it."""
        for g in (self.density, self.k0):
            ready = getattr(g, 'params_ready', None)
            if ready is not None:
                ready()

    def _k4_grid(self, act_shift_grid=None, live=None):
        """k4_grid_desc of this model.  live = (act_shift scalar, interval): look samples up in the live mask (_k4_live_mask)
        instead of mask_cache.mask -- the render path; None: the MaskGrid itself (sample counters, fast_color_thres == 0)."""
        gd = N.GridDesc()
        self._k4_params_ready()
        dens = self.density.grid
        k0cl, cpad, k0_layout = self._k4_k0_channel_last()
        gd.density = dens.data_ptr()
        gd.k0 = k0cl.data_ptr()
        gd.k0_layout = k0_layout
        gd.k0_cpad = cpad
        gd.k0_ch = self.k0.grid.shape[1]
        gd.dims = (N.C.c_int32 * 3)(*[int(v) for v in dens.shape[2:]])
        if act_shift_grid is not None:
            gd.act_shift = act_shift_grid.data_ptr()
            gd.act_depth = int(act_shift_grid.numel())
        mc = self.mask_cache
        gd.mask = mc.mask.data_ptr()
        gd.mask_dims = (N.C.c_int32 * 3)(*[int(v) for v in mc.mask.shape])
        c = self._k4_cache()
        hkey = ('host3', str(dens.device)) + tuple((t.data_ptr(), t._version) for t in
                                                   (self.xyz_min, self.xyz_max, mc.xyz2ijk_scale, mc.xyz2ijk_shift))
        if c.get('host_key') != hkey:      # tiny D2H copies, once
            c['host_key'] = hkey
            c['host'] = (N.vec3(self.xyz_min), N.vec3(self.xyz_max), N.vec3(mc.xyz2ijk_scale), N.vec3(mc.xyz2ijk_shift))
        gd.xyz_min, gd.xyz_max, gd.xyz2ijk_scale, gd.xyz2ijk_shift = c['host']
        if live is not None and self.fast_color_thres > 0 and _LIVE_MASK:
            lm, occ = self._k4_live_mask(gd, act_shift_grid, live[0], live[1])
            gd.mask = lm.data_ptr()
            gd.occ_summary = occ.data_ptr()
        else:
            gd.occ_summary = self._k4_occ_summary().data_ptr()
        return gd

    def _k4_plan(self, tag, extra_key, build):
        """Per-call descriptors of the fused path (k4_grid_desc, k4_mlp_desc and the tensors they point into) cached on the VERSIONS of every
        tensor they are derived from: a render loop re-derives nothing per frame but this key (~30 attribute reads) -- the per-part caches
        below (k0 repack, live mask, packed rgbnet, host copies) each rebuilt their own key and ctypes structs on every call, ~0.2 ms of
        Python per frame, which paces bench.py's pipelined loop on a slow host (round 5: 0.86 -> 1.3 ms per frame on some boxes).
        ``build()`` -> (value, tensors to keep alive).  A plan is valid per HIP stream it was built or re-validated on: the first call on
        another stream goes through ``build`` again, which orders that stream behind the load-time kernels (see _k4_live_mask)."""
        mc = self.mask_cache
        ts = [self.density.grid, self.k0.grid, mc.mask, self.xyz_min, self.xyz_max, mc.xyz2ijk_scale, mc.xyz2ijk_shift]
        act = getattr(self, 'act_shift', None)
        ts.append(act.grid if isinstance(act, nn.Module) else act)
        if self.rgbnet is not None:
            c = self._k4_cache()
            lins = c.get('plan_lins')
            if lins is None or lins[0] is not self.rgbnet:
                lins = c['plan_lins'] = (self.rgbnet, [m for m in self.rgbnet.modules() if isinstance(m, nn.Linear)])
            for l in lins[1]:
                ts.append(l.weight)
                ts.append(l.bias)
        key = (tag, extra_key, _LIVE_MASK, os.environ.get('K4_MLP')) + tuple((t.data_ptr(), t._version) for t in ts if t is not None)
        return self._k4_plan_for(tag, key, build)

    def _k4_versions_key(self):
        """(storage, version) of every tensor the fused path's load-time state is derived from (see _k4_plan)."""
        mc = self.mask_cache
        ts = [self.density.grid, self.k0.grid, mc.mask, self.xyz_min, self.xyz_max, mc.xyz2ijk_scale, mc.xyz2ijk_shift]
        act = getattr(self, 'act_shift', None)
        ts.append(act.grid if isinstance(act, nn.Module) else act)
        if self.rgbnet is not None:
            for l in self.rgbnet.modules():
                if isinstance(l, nn.Linear):
                    ts.append(l.weight)
                    ts.append(l.bias)
        return tuple((t.data_ptr(), t._version) for t in ts if t is not None)

    def _k4_plan_for(self, tag, key, build):
        c = self._k4_cache()
        plan = c.get(('plan', tag))
        st = N.stream().value
        if plan is not None and plan[0] == key and st in plan[2]:
            return plan[1]
        value, keep = build()
        seen = plan[2] if (plan is not None and plan[0] == key) else set()
        seen.add(st)
        c[('plan', tag)] = (key, value, seen, keep)
        return value

    def k4_warm(self, stepsize=None):
        """Build every load-time cache of the fused path (k0 channel-last repack, packed rgbnet, host copies of the bbox and, when the
        caller names the render stepsize, the live mask) on the current stream.  Callers that fan work out over several HIP streams
        call this before forking them."""
        if self._k4_fusable():
            # (a render loop warms before every frame: nothing to do while no tensor behind the caches changed -- the per-part caches below rebuild their keys
            #  and ctypes structs on every call, ~0.2 ms of host time in front of a frame's first kernel)
            c = self._k4_cache()
            wkey = None
            if getattr(self, 'mask_cache', None) is not None:
                wkey = (None if stepsize is None else float(stepsize), _LIVE_MASK, os.environ.get('K4_MLP'), N.stream().value) + self._k4_versions_key()
                if c.get('warm_key') == wkey:
                    return
            act = getattr(self, 'act_shift', None)
            mpi = isinstance(act, nn.Module)
            live = None
            if stepsize is not None:
                live = (0.0 if mpi else self._k4_host_scalar('act_shift', act), float(stepsize * self.voxel_size_ratio))
            self._k4_grid(act_shift_grid=act.grid if mpi else None, live=live)
            if self.rgbnet is not None:
                self._k4_mlp(k0_skip=0, spatial_pe=0)
            c['warm_key'] = wkey

    def _k4_host_scalar(self, name, t):
        """float(t) for a 1-element device buffer without a D2H sync per call (cached per version)."""
        c = self._k4_cache()
        key = (t.data_ptr(), t._version)
        if c.get(name + '_key') != key:
            c[name + '_key'], c[name] = key, float(t)
        return c[name]

    @staticmethod
    def _k4_check_rays(rays_o, rays_d, viewdirs):
        assert len(rays_o.shape) == 2 and rays_o.shape[-1] == 3, 'Only suuport point queries in [N, 3] format'
        if not rays_o.is_cuda:
            raise N.K4Error('rays must be on the GPU: the MI355X-native marcher has no CPU path '
                            '(the CPU oracle lives in oracle/ and is test infrastructure only)')
        return rays_o.float().contiguous(), rays_d.float().contiguous(), viewdirs.float().contiguous()


'''Model'''
def _take(mask, *ts):
    """``t[mask]`` for every t (first axis) with ONE nonzero -- one host synchronisation per mask instead of one per tensor: the staged
    (training) forward of the reference's op sequence filters its sample list three times (lib/dvgo.py:360-378)."""
    i = mask.nonzero().squeeze(1)
    return [t.index_select(0, i) for t in ts]


class DirectVoxGO(torch.nn.Module, _FusedMarcher):
    def __init__(self, xyz_min, xyz_max,
                 num_voxels=0, num_voxels_base=0,
                 alpha_init=None,
                 mask_cache_path=None, mask_cache_thres=1e-3, mask_cache_world_size=None,
                 fast_color_thres=0,
                 density_type='DenseGrid', k0_type='DenseGrid',
                 density_config={}, k0_config={},
                 rgbnet_dim=0, rgbnet_direct=False, rgbnet_full_implicit=False,
                 rgbnet_depth=3, rgbnet_width=128,
                 viewbase_pe=4,
                 **kwargs):
        super(DirectVoxGO, self).__init__()
        self.register_buffer('xyz_min', torch.Tensor(xyz_min))
        self.register_buffer('xyz_max', torch.Tensor(xyz_max))
        self.fast_color_thres = fast_color_thres

        # base grid resolution / density bias (lib/dvgo.py:41-50)
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self.alpha_init = alpha_init
        self.register_buffer('act_shift', torch.FloatTensor([np.log(1 / (1 - alpha_init) - 1)]))
        self._set_grid_resolution(num_voxels)

        self.density_type = density_type
        self.density_config = density_config
        self.density = grid.create_grid(
            density_type, channels=1, world_size=self.world_size,
            xyz_min=self.xyz_min, xyz_max=self.xyz_max, config=self.density_config)

        self.rgbnet_kwargs = {
            'rgbnet_dim': rgbnet_dim, 'rgbnet_direct': rgbnet_direct,
            'rgbnet_full_implicit': rgbnet_full_implicit,
            'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width,
            'viewbase_pe': viewbase_pe,
        }
        self.k0_type = k0_type
        self.k0_config = k0_config
        self.rgbnet_full_implicit = rgbnet_full_implicit
        self.dim_rend = 3
        self.act_type = 'mlp'
        self.mode_type = 'mlp'
        if rgbnet_full_implicit:
            raise NotImplementedError('rgbnet_full_implicit is not used by any BASELINE configuration')
        if rgbnet_dim <= 0:
            # colour voxel grid (coarse stage, lib/dvgo.py:74-81)
            self.k0_dim = 3
            self.k0 = grid.create_grid(
                k0_type, channels=self.k0_dim, world_size=self.world_size,
                xyz_min=self.xyz_min, xyz_max=self.xyz_max, config=self.k0_config)
            self.rgbnet = None
            self.rgbnet_direct = True
        else:
            # feature voxel grid + shallow MLP (fine stage, lib/dvgo.py:82-124)
            self.k0_dim = rgbnet_dim
            self.k0 = grid.create_grid(
                k0_type, channels=self.k0_dim, world_size=self.world_size,
                xyz_min=self.xyz_min, xyz_max=self.xyz_max, config=self.k0_config)
            self.rgbnet_direct = rgbnet_direct
            self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
            dim0 = (3 + 3 * viewbase_pe * 2)
            dim0 += self.k0_dim if rgbnet_direct else self.k0_dim - 3
            self.dim0 = dim0
            self.rgbnet = nn.Sequential(
                nn.Linear(dim0, rgbnet_width), nn.ReLU(inplace=True),
                *[
                    nn.Sequential(nn.Linear(rgbnet_width, rgbnet_width), nn.ReLU(inplace=True))
                    for _ in range(rgbnet_depth - 2)
                ],
                nn.Linear(rgbnet_width, 3),
            )
            nn.init.constant_(self.rgbnet[-1].bias, 0)

        # occupancy grid (lib/dvgo.py:130-150)
        self.mask_cache_path = mask_cache_path
        self.mask_cache_thres = mask_cache_thres
        if mask_cache_world_size is None:
            mask_cache_world_size = self.world_size
        if mask_cache_path is not None and mask_cache_path:
            mask = coarse_mask_on_grid(mask_cache_path, mask_cache_thres, self.xyz_min, self.xyz_max,
                                       mask_cache_world_size)
        else:
            mask = torch.ones(list(mask_cache_world_size), dtype=torch.bool)
        self.mask_cache = grid.MaskGrid(path=None, mask=mask, xyz_min=self.xyz_min, xyz_max=self.xyz_max)

    def _set_grid_resolution(self, num_voxels):
        self.num_voxels = num_voxels
        self.voxel_size = ((self.xyz_max - self.xyz_min).prod() / num_voxels).pow(1 / 3)
        self.world_size = ((self.xyz_max - self.xyz_min) / self.voxel_size).long()
        self.max_world_size = self.world_size.max()
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base

    def get_kwargs(self):
        return {
            'xyz_min': self.xyz_min.cpu().numpy(),
            'xyz_max': self.xyz_max.cpu().numpy(),
            'num_voxels': self.num_voxels,
            'num_voxels_base': self.num_voxels_base,
            'alpha_init': self.alpha_init,
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
    def maskout_near_cam_vox(self, cam_o, near_clip):
        """lib/dvgo.py:186-198: density = -100 at grid nodes closer than `near_clip` to any camera centre."""
        nodes = grid.grid_nodes(self.xyz_min, self.xyz_max, self.world_size.tolist())
        cam_o = cam_o.to(nodes.device)
        nearest = torch.stack([(nodes.unsqueeze(-2) - co).pow(2).sum(-1).sqrt().amin(-1) for co in cam_o.split(100)]).amin(0)
        self.density.grid.data[nearest[None, None] <= near_clip] = -100
        torch.autograd.graph.increment_version(self.density.grid)

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        """Progressive growing (lib/dvgo.py:200-221): resample density / k0 (k4_resample_trilinear); refresh the occupancy while the
        grid is <= 256^3: old mask at the new nodes AND max-pooled alpha > fast_color_thres."""
        self._set_grid_resolution(num_voxels)
        self.density.scale_volume_grid(self.world_size)
        self.k0.scale_volume_grid(self.world_size)
        if int(np.prod(self.world_size.tolist())) <= 256 ** 3:
            nodes = grid.grid_nodes(self.xyz_min, self.xyz_max, self.world_size.tolist())
            occupied = grid.occupancy_from_alpha(self.activate_density(self.density.get_dense_grid())[0, 0], self.fast_color_thres)
            self.mask_cache = grid.MaskGrid(path=None, mask=self.mask_cache(nodes) & occupied,
                                            xyz_min=self.xyz_min, xyz_max=self.xyz_max).to(nodes.device)

    @torch.no_grad()
    def update_occupancy_cache(self):
        """lib/dvgo.py:223-233: mask &= maxpool3(alpha(density at the mask's nodes)) > fast_color_thres, on HIP kernels."""
        nodes = grid.grid_nodes(self.xyz_min, self.xyz_max, list(self.mask_cache.mask.shape))
        alpha = self.activate_density(self.density(nodes))
        self.mask_cache.mask &= grid.occupancy_from_alpha(alpha, self.fast_color_thres)

    def voxel_count_views(self, rays_o_tr, rays_d_tr, imsz, near, far, stepsize, downrate=1, irregular_shape=False):
        """lib/dvgo.py:235-268: per voxel, the number of training views whose rays pass it (gradient of an all-ones grid > 1)."""
        far = 1e9
        n_samples = int(np.linalg.norm(np.array(self.world_size.cpu()) + 1) / stepsize) + 1
        count = torch.zeros_like(self.density.get_dense_grid())
        dev = count.device
        rng = torch.arange(n_samples, device=dev)[None].float()
        for rays_o_, rays_d_ in zip(rays_o_tr.split(imsz), rays_d_tr.split(imsz)):
            ones = grid.DenseGrid(1, self.world_size, self.xyz_min, self.xyz_max).to(dev)
            if irregular_shape:
                chunks_o, chunks_d = rays_o_.split(10000), rays_d_.split(10000)
            else:
                chunks_o = rays_o_[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
                chunks_d = rays_d_[::downrate, ::downrate].to(dev).flatten(0, -2).split(10000)
            for rays_o, rays_d in zip(chunks_o, chunks_d):
                rays_o, rays_d = rays_o.to(dev), rays_d.to(dev)
                vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
                rate_a, rate_b = (self.xyz_max - rays_o) / vec, (self.xyz_min - rays_o) / vec
                t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
                step = stepsize * self.voxel_size * rng
                interpx = t_min[..., None] + step / rays_d.norm(dim=-1, keepdim=True)
                rays_pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
                with torch.enable_grad():
                    ones(rays_pts).sum().backward()
            with torch.no_grad():
                count += (ones.grid.grad > 1)
        return count

    def density_total_variation_add_grad(self, weight, dense_mode):
        '''lib/dvgo.py:268-270: isotropic TV weight scaled by max(world_size)/128.'''
        w = weight * self.world_size.max() / 128
        self.density.total_variation_add_grad(w, w, w, dense_mode)

    def k0_total_variation_add_grad(self, weight, dense_mode):
        '''lib/dvgo.py:272-274.'''
        w = weight * self.world_size.max() / 128
        self.k0.total_variation_add_grad(w, w, w, dense_mode)

    def activate_density(self, density, interval=None):
        interval = interval if interval is not None else self.voxel_size_ratio
        shape = density.shape
        return Raw2Alpha.apply(density.flatten(), self.act_shift, interval).reshape(shape)

    @torch.no_grad()
    def hit_coarse_geo(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        '''Which rays meet an occupied cell of the mask cache (lib/dvgo.py:281-293)?  Rays of any leading shape -> bool of that shape.'''
        shape = rays_o.shape[:-1]
        ro = rays_o.reshape(-1, 3).contiguous()
        rd = rays_d.reshape(-1, 3).contiguous()
        # far = 1e9: the caller's far can be too small, rays end at the scene box anyway
        ray_pts, mask_outbbox, ray_id = render_utils_cuda.sample_pts_on_rays(
            ro, rd, self.xyz_min, self.xyz_max, near, 1e9, stepsize * self.voxel_size)[:3]
        inside = ~mask_outbbox
        occupied = self.mask_cache(ray_pts[inside])
        hit = torch.zeros([ro.shape[0]], dtype=torch.bool, device=ro.device)
        hit[ray_id[inside][occupied]] = True
        return hit.reshape(shape)

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        '''Sample query points on rays (lib/dvgo.py:295-325): points sorted near to far.'''
        far = 1e9  # the given far can be too small while rays stop when hitting scene bbox
        rays_o = rays_o.contiguous()
        rays_d = rays_d.contiguous()
        stepdist = stepsize * self.voxel_size
        N_samples = int((self.max_world_size - 1) / stepsize) + 1
        ray_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max = render_utils_cuda.sample_pts_on_rays(
            rays_o, rays_d, self.xyz_min, self.xyz_max, near, far, stepdist)
        ray_pts, ray_id, step_id = _take(~mask_outbbox, ray_pts, ray_id, step_id)
        return ray_pts, ray_id, step_id, None, N_samples

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
            return self._forward_staged(rays_o, rays_d, viewdirs, **render_kwargs)
        self._k4_params_ready()
        return self._forward_fused(rays_o, rays_d, viewdirs, **render_kwargs)

    def _forward_fused(self, rays_o, rays_d, viewdirs, near, far, stepsize, bg, render_depth=False,
                       k4_img_w=0, k4_counters=None, k4_out=None, k4_ws_slot=0, k4_live_mask=True, **_ignored):
        Nr = rays_o.shape[0]
        dev = rays_o.device
        if k4_out is not None:                  # caller-provided outputs (e.g. slices of an all-gather send buffer)
            rgb, depth, ainv = k4_out
            assert rgb.shape == (Nr, 3) and depth.shape == (Nr,) and ainv.shape == (Nr,)
        else:
            rgb = torch.empty([Nr, 3], dtype=torch.float32, device=dev)
            depth = torch.empty([Nr], dtype=torch.float32, device=dev)
            ainv = torch.empty([Nr], dtype=torch.float32, device=dev)
        use_live = bool(k4_live_mask and k4_counters is None)

        def build():
            md, keep = self._k4_mlp(k0_skip=0 if (self.rgbnet is None or self.rgbnet_direct) else 3, spatial_pe=0)
            sd = float(stepsize * self.voxel_size)                         # lib/dvgo.py:310
            itv = float(stepsize * self.voxel_size_ratio)                  # lib/dvgo.py:341
            ash = self._k4_host_scalar('act_shift', self.act_shift)
            # sample counters are the ALGORITHM's counts: the counting pass looks samples up in mask_cache itself
            gd = self._k4_grid(live=(ash, itv) if use_live else None)
            dn = int((self.max_world_size - 1) / stepsize) + 1             # lib/dvgo.py:311
            diag = float((self.xyz_max - self.xyz_min).norm())             # (one read-back per plan, not per call)
            return (md, gd, sd, itv, ash, dn, int(np.ceil(diag / sd)) + 2), keep
        md, gd, stepdist, interval, act_shift, depth_n, max_steps = self._k4_plan(
            'dvgo', (float(stepsize), use_live, float(self.fast_color_thres), float(self.voxel_size), float(self.voxel_size_ratio)), build)
        if Nr == 0:
            ret = {'alphainv_last': ainv, 'rgb_marched': rgb, 'rgb_feature': rgb}
            if render_depth:
                ret['depth'] = depth
            return ret
        ws, ws_bytes = self._k4_workspace(Nr, k4_img_w, max_steps, dev, k4_ws_slot)
        N.check(N.lib().k4_march_dvgo_fwd(
            N.f32(rays_o), N.f32(rays_d), N.f32(viewdirs), Nr, int(k4_img_w), N.C.byref(gd), N.C.byref(md),
            float(near), 1e9, stepdist, max_steps, depth_n, act_shift, interval,
            float(self.fast_color_thres), float(bg), N.ptr(ws), ws_bytes, N.f32(rgb), N.f32(depth), N.f32(ainv),
            None if k4_counters is None else N.ptr(k4_counters), N.stream()), 'k4_march_dvgo_fwd')
        ret = {'alphainv_last': ainv, 'rgb_marched': rgb, 'rgb_feature': rgb}
        if render_depth:
            ret['depth'] = depth
        return ret

    def _forward_staged(self, rays_o, rays_d, viewdirs, near, far, stepsize, bg, render_depth=False, **_ignored):
        """The reference's op sequence (lib/dvgo.py:338-446) on the staged gfx950 kernels."""
        ret_dict = {}
        Nr = len(rays_o)
        ray_pts, ray_id, step_id, _, N_samples = self.sample_ray(
            rays_o=rays_o, rays_d=rays_d, near=near, far=far, stepsize=stepsize)
        interval = stepsize * self.voxel_size_ratio
        if self.mask_cache is not None:
            mask1 = self.mask_cache(ray_pts)
            ray_pts, ray_id, step_id = _take(mask1, ray_pts, ray_id, step_id)
        density = self.density(ray_pts)
        alpha = self.activate_density(density, interval)
        if self.fast_color_thres > 0:
            mask2 = (alpha > self.fast_color_thres)
            ray_pts, ray_id, step_id, alpha = _take(mask2, ray_pts, ray_id, step_id, alpha)
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, Nr)
        if self.fast_color_thres > 0:
            mask3 = (weights > self.fast_color_thres)
            ray_pts, ray_id, step_id, alpha, weights = _take(mask3, ray_pts, ray_id, step_id, alpha, weights)
        k0 = self.k0(ray_pts)
        if k0.dim() == 1:
            k0 = k0.unsqueeze(-1)
        if self.rgbnet is None:
            rgb_raw = torch.sigmoid(k0)
        else:
            if self.rgbnet_direct:
                k0_view = k0
            else:
                k0_view = k0[:, 3:]
                k0_diffuse = k0[:, :3]
            viewdirs_emb = (viewdirs.unsqueeze(-1) * self.viewfreq).flatten(-2)
            viewdirs_emb = torch.cat([viewdirs, viewdirs_emb.sin(), viewdirs_emb.cos()], -1)
            viewdirs_emb = viewdirs_emb.flatten(0, -2)[ray_id]
            rgb_raw = self._k4_rgbnet_sigmoid(torch.cat([k0_view, viewdirs_emb], -1), None if self.rgbnet_direct else k0_diffuse)
        rgb_feature = segment_sum(weights.unsqueeze(-1) * rgb_raw, ray_id, Nr)
        rgb_marched = rgb_feature
        rgb_marched += (alphainv_last.unsqueeze(-1) * bg)                 # aliases rgb_feature (lib/dvgo.py:425-427)
        s = (step_id + 0.5) / N_samples
        ret_dict.update({
            'alphainv_last': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched,
            'rgb_feature': rgb_feature, 'raw_alpha': alpha, 'raw_rgb': rgb_raw, 'ray_id': ray_id,
        })
        if render_depth:
            with torch.no_grad():
                ret_dict['depth'] = segment_sum(weights * s, ray_id, Nr)
        return ret_dict


def pack_mlp_mfma(lins):
    """rgbnet weights in v_mfma_f32_32x32x2_f32 operand order (layout: include/k4nerf.h, k4_mlp_desc)."""
    dev = lins[0].weight.device
    W, dim0 = lins[0].out_features, lins[0].in_features
    NB, k1p = W // 32, (dim0 + 2) & ~1
    lane = torch.arange(64, device=dev)
    row = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
    w1 = torch.zeros([W, k1p], dtype=torch.float32, device=dev)
    w1[:, :dim0] = lins[0].weight.detach().float()
    w1[:, dim0] = lins[0].bias.detach().float()
    mb = torch.arange(NB, device=dev)
    kk = torch.arange(k1p // 2, device=dev)
    r = torch.arange(16, device=dev)
    parts = [w1[(mb[:, None, None] * 32 + (lane & 31)[None, None, :]),
                (2 * kk[None, :, None] + (lane >> 5)[None, None, :])].reshape(-1)]
    if len(lins) == 3:
        w2 = lins[1].weight.detach().float()
        j2 = mb[:, None, None, None] * 32 + (lane & 31)[None, None, None, :]
        k = mb[None, :, None, None] * 32 + row(r[None, None, :, None], (lane >> 5)[None, None, None, :])
        parts.append(w2[j2.expand(NB, NB, 16, 64), k.expand(NB, NB, 16, 64)].reshape(-1))
        b2 = lins[1].bias.detach().float()
        b2a = torch.zeros([NB, 64], dtype=torch.float32, device=dev)
        b2a[:, :32] = b2.reshape(NB, 32)
        parts.append(b2a.reshape(-1))
    wo = lins[-1].weight.detach().float()                               # [3, W]
    wot = torch.zeros([NB, 16, 2, 4], dtype=torch.float32, device=dev)
    h = torch.arange(2, device=dev)
    idx = mb[:, None, None] * 32 + row(r[None, :, None], h[None, None, :])      # [NB,16,2]
    wot[..., :3] = wo.t()[idx]
    parts.append(wot.reshape(-1))
    bo = torch.zeros([4], dtype=torch.float32, device=dev)
    bo[:3] = lins[-1].bias.detach().float()
    parts.append(bo)
    parts += _pack_mlp_split_bf16(w1, lins, wot, bo, 3, 3)                                    # K4_MLP_ARITH_B3 section: exact
    parts += _pack_mlp_split_bf16(w1, lins, wot, bo, int(N.lib().k4_mlp_b2_layer1_terms()), 2)   # default (b2) section: the two leading terms
    out = torch.cat(parts).contiguous()
    want = N.lib().k4_mlp_packed_floats(dim0, W, len(lins) - 2)
    assert out.numel() == want, (out.numel(), want)
    return out


def _split3_bf16(x):
    """Exact 3-term bf16 split of an fp32 tensor: x == t0 + t1 + t2 (each RNE to bf16 of the running remainder)."""
    t0 = x.to(torch.bfloat16)
    r1 = x - t0.float()
    t1 = r1.to(torch.bfloat16)
    t2 = (r1 - t1.float()).to(torch.bfloat16)
    return t0, t1, t2


def _pack_mlp_split_bf16(w1ext, lins, wot, bo, nt1=3, nt2=3):
    """Split-bf16 section of the packed rgbnet buffer (layout: csrc/k4_march.hip, MlpLayoutB3), returned as fp32-typed
    views of the raw bytes.  v_mfma_f32_32x32x16_bf16 operand order: lane l holds 8 bf16 = row (l&31), k = 8*(l>>5)+e."""
    dev = w1ext.device
    W, k1p = w1ext.shape
    NB, KB1, KB2 = W // 32, (k1p + 15) // 16, W // 16
    lane = torch.arange(64, device=dev)
    e = torch.arange(8, device=dev)
    mb = torch.arange(NB, device=dev)

    def as_f32(terms, index_fn):
        # [..., terms, 64 lanes, 8] bf16 -> flat fp32 view
        g = torch.stack([index_fn(t) for t in terms], dim=-3)
        return g.contiguous().view(torch.int16).reshape(-1).view(torch.float32)

    w1p = torch.zeros([W, KB1 * 16], dtype=torch.float32, device=dev)
    w1p[:, :k1p] = w1ext
    kb = torch.arange(KB1, device=dev)
    j = (mb[:, None, None, None] * 32 + (lane & 31)[None, None, :, None]).expand(NB, KB1, 64, 8)
    k = (kb[None, :, None, None] * 16 + 8 * (lane >> 5)[None, None, :, None] + e[None, None, None, :]).expand(NB, KB1, 64, 8)
    parts = [as_f32(_split3_bf16(w1p)[:nt1], lambda t: t[j, k])]
    if len(lins) == 3:
        w2 = lins[1].weight.detach().float()
        kb = torch.arange(KB2, device=dev)
        h = (lane >> 5)[None, None, :, None]
        n = ((kb >> 1)[None, :, None, None] * 32 + (e & 3)[None, None, None, :]
             + 8 * (2 * (kb & 1)[None, :, None, None] + (e >> 2)[None, None, None, :]) + 4 * h).expand(NB, KB2, 64, 8)
        j2 = (mb[:, None, None, None] * 32 + (lane & 31)[None, None, :, None]).expand(NB, KB2, 64, 8)
        parts.append(as_f32(_split3_bf16(w2)[:nt2], lambda t: t[j2, n]))
        b2 = lins[1].bias.detach().float()
        r = torch.arange(16, device=dev)
        hh = torch.arange(2, device=dev)
        row = (r & 3)[None, None, :] + 8 * (r >> 2)[None, None, :] + 4 * hh[None, :, None]
        parts.append(b2[mb[:, None, None] * 32 + row].reshape(-1))
    parts += [wot.reshape(-1), bo]
    return parts


def coarse_mask_on_grid(path, thres, xyz_min, xyz_max, world_size):
    """Evaluate the coarse-stage occupancy (MaskGrid(path=...)) at the nodes of a finer grid
    (lib/dvgo.py:136-145).  The lookup itself runs on the GPU kernel."""
    if not torch.cuda.is_available():
        raise N.K4Error('mask_cache_path needs the GPU maskcache_lookup kernel (no CPU path)')
    mc = grid.MaskGrid(path=path, mask_cache_thres=thres).cuda()
    xyz = torch.stack(torch.meshgrid(
        torch.linspace(float(xyz_min[0]), float(xyz_max[0]), int(world_size[0])),
        torch.linspace(float(xyz_min[1]), float(xyz_max[1]), int(world_size[1])),
        torch.linspace(float(xyz_min[2]), float(xyz_max[2]), int(world_size[2])), indexing='ij'), -1)
    return mc(xyz.cuda()).cpu()


def segment_sum(src, index, n):
    """torch_scatter.segment_coo(src, index, out=zeros([n,...]), reduce='sum') for a sorted index
    (lib/dvgo.py:415-419).  Inference: k4_segment_sum; under autograd: differentiable index_add."""
    if torch.is_grad_enabled() and src.requires_grad:
        return SegmentSum.apply(src, index, n)
    return _segment_sum_fwd(src, index, n)


def _segment_sum_fwd(src, index, n):
    C_ = 1 if src.dim() == 1 else src.shape[1]
    out = torch.empty([n] + list(src.shape[1:]), dtype=torch.float32, device=src.device)
    srcc = src.detach().float().contiguous()
    idx = index.contiguous()
    N.check(N.lib().k4_segment_sum(N.f32(srcc), N.ptr(idx), srcc.shape[0], C_, n, N.f32(out), N.stream()),
            'segment_sum')
    return out


class SegmentSum(torch.autograd.Function):
    """segment_coo(sum) with its HIP backward: grad_src[i] = grad_out[index[i]] (k4_segment_sum_backward)."""

    @staticmethod
    def forward(ctx, src, index, n):
        ctx.save_for_backward(index)
        ctx.src_shape = tuple(src.shape)
        return _segment_sum_fwd(src.detach(), index, n)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        index, = ctx.saved_tensors
        C_ = 1 if len(ctx.src_shape) == 1 else ctx.src_shape[1]
        gs = torch.empty(ctx.src_shape, dtype=torch.float32, device=grad_out.device)
        go = grad_out.float().contiguous()
        idx = index.contiguous()
        N.check(N.lib().k4_segment_sum_backward(N.f32(go), N.ptr(idx), ctx.src_shape[0], C_, N.f32(gs), N.stream()),
                'segment_sum_backward')
        return gs, None, None


''' Misc
'''
class Raw2Alpha(torch.autograd.Function):
    """alpha = 1 - (1 + exp(density + shift)) ^ (-interval)   (lib/dvgo.py:453-477)"""
    @staticmethod
    def forward(ctx, density, shift, interval):
        exp, alpha = render_utils_cuda.raw2alpha(density.contiguous(), float(shift), float(interval))
        if density.requires_grad:
            ctx.save_for_backward(exp)
            ctx.interval = float(interval)
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        exp = ctx.saved_tensors[0]
        return render_utils_cuda.raw2alpha_backward(exp, grad_back.contiguous(), ctx.interval), None, None


class Raw2Alpha_nonuni(torch.autograd.Function):
    @staticmethod
    def forward(ctx, density, shift, interval):
        exp, alpha = render_utils_cuda.raw2alpha_nonuni(density.contiguous(), float(shift), interval.contiguous())
        if density.requires_grad:
            ctx.save_for_backward(exp, interval)
        return alpha

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_back):
        exp, interval = ctx.saved_tensors
        return render_utils_cuda.raw2alpha_nonuni_backward(exp, grad_back.contiguous(), interval), None, None


class Alphas2Weights(torch.autograd.Function):
    """(lib/dvgo.py:495-511)"""
    @staticmethod
    def forward(ctx, alpha, ray_id, N_):
        weights, T, alphainv_last, i_start, i_end = render_utils_cuda.alpha2weight(alpha.contiguous(), ray_id.contiguous(), N_)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, weights, T, alphainv_last, i_start, i_end)
            ctx.n_rays = N_
        return weights, alphainv_last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, weights, T, alphainv_last, i_start, i_end = ctx.saved_tensors
        grad = render_utils_cuda.alpha2weight_backward(
            alpha, weights, T, alphainv_last, i_start, i_end, ctx.n_rays, grad_weights, grad_last)
        return grad, None, None


''' Ray and batch
'''
def get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode='center'):
    """Pixel -> camera ray, pixel centres at +0.5 (lib/dvgo.py:516-544); runs on c2w's device."""
    dev = c2w.device
    K = torch.as_tensor(np.asarray(K), dtype=torch.float32, device=dev) if not torch.is_tensor(K) else K.to(dev).float()
    i = torch.arange(W, dtype=torch.float32, device=dev)[None, :].expand(H, W)
    j = torch.arange(H, dtype=torch.float32, device=dev)[:, None].expand(H, W)
    if mode == 'lefttop':
        pass
    elif mode == 'center':
        i, j = i + 0.5, j + 0.5
    elif mode == 'random':
        i = i + torch.rand_like(i)
        j = j + torch.rand_like(j)
    else:
        raise NotImplementedError
    if flip_x:
        i = i.flip((1,))
    if flip_y:
        j = j.flip((0,))
    if inverse_y:
        dirs = torch.stack([(i - K[0][2]) / K[0][0], (j - K[1][2]) / K[1][1], torch.ones_like(i)], -1)
    else:
        dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
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


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center'):
    """viewdirs normalised BEFORE the NDC warp (lib/dvgo.py:577-582).  On a GPU pose: one HIP launch
    (k4_get_rays_of_a_view) instead of ~25 elementwise kernels and their [H,W,3] temporaries."""
    if torch.is_tensor(c2w) and c2w.is_cuda and mode in ('center', 'lefttop'):
        dev = c2w.device
        # focal from the caller's K BEFORE any fp32 cast: the reference evaluates -1./(W/(2.*focal)) in Python floats from K[0][0]
        # (typically float64, lib/dvgo.py:559-563), exactly as the host fallback below does
        focal = float(K[0][0])
        Kt = (torch.as_tensor(np.asarray(K), dtype=torch.float32) if not torch.is_tensor(K) else K.detach().float().cpu())
        Kd = Kt.contiguous().to(dev)
        M = c2w.detach().float().contiguous()
        if M.shape[-1] != 4 or M.shape[0] < 3:
            raise ValueError('c2w must be [3,4] or [4,4]')
        ro, rd, vd = (torch.empty([H, W, 3], dtype=torch.float32, device=dev) for _ in range(3))
        N.check(N.lib().k4_get_rays_of_a_view(int(H), int(W), N.f32(Kd), N.f32(M), int(bool(ndc)), int(bool(inverse_y)),
                                              int(bool(flip_x)), int(bool(flip_y)), 1 if mode == 'center' else 0, focal,
                                              N.f32(ro), N.f32(rd), N.f32(vd), N.stream()), 'k4_get_rays_of_a_view')
        return ro, rd, vd
    rays_o, rays_d = get_rays(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y, mode=mode)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, float(K[0][0]), 1., rays_o, rays_d)
    return rays_o, rays_d, viewdirs


# ---------------------------------------------------------------------------------------------------------------------
# training-ray tables and batch generators (lib/dvgo.py:584-680,761-768): what run.py / run_sr.py call before the optimisation
# loop.  Rays come from get_rays_of_a_view (one HIP launch per view on a GPU pose); same return tuples as the reference.
# ---------------------------------------------------------------------------------------------------------------------
def _rays_of(view_hw, K, c2w, ndc, inverse_y, flip_x, flip_y, device):
    H, W = int(view_hw[0]), int(view_hw[1])
    ro, rd, vd = get_rays_of_a_view(H=H, W=W, K=K, c2w=c2w, ndc=ndc, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
    return ro.to(device), rd.to(device), vd.to(device)


@torch.no_grad()
def get_training_rays(rgb_tr, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """All views share one size and one intrinsic matrix: per-view [H,W,3] tables (lib/dvgo.py:584-607)."""
    n = len(rgb_tr)
    assert n == len(train_poses) == len(Ks) == len(HW)
    assert len(np.unique(np.asarray(HW), axis=0)) == 1, 'get_training_rays: views of one size only'
    assert len(np.unique(np.asarray(Ks).reshape(n, -1), axis=0)) == 1, 'get_training_rays: one intrinsic matrix only'
    H, W = (int(v) for v in HW[0])
    dev = rgb_tr.device
    rays_o_tr, rays_d_tr, viewdirs_tr = (torch.zeros([n, H, W, 3], device=dev) for _ in range(3))
    for i, c2w in enumerate(train_poses):
        rays_o_tr[i], rays_d_tr[i], viewdirs_tr[i] = _rays_of((H, W), Ks[0], c2w, ndc, inverse_y, flip_x, flip_y, dev)
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, [1] * n


@torch.no_grad()
def get_training_rays_flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """Views of different sizes: one flat [N,3] table, imsz = pixels per view (lib/dvgo.py:610-640)."""
    assert len(rgb_tr_ori) == len(train_poses) == len(Ks) == len(HW)
    dev = rgb_tr_ori[0].device
    cols = ([], [], [], [])
    imsz = []
    for c2w, img, hw, K in zip(train_poses, rgb_tr_ori, HW, Ks):
        assert tuple(img.shape[:2]) == (int(hw[0]), int(hw[1]))
        ro, rd, vd = _rays_of(hw, K, c2w, ndc, inverse_y, flip_x, flip_y, dev)
        for lst, t in zip(cols, (img, ro, rd, vd)):
            lst.append(t.reshape(-1, 3).float())
        imsz.append(int(hw[0]) * int(hw[1]))
    rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr = (torch.cat(c, 0) for c in cols)
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz


@torch.no_grad()
def get_training_rays_in_maskcache_sampling(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, model, render_kwargs):
    """Only the rays that meet the coarse geometry (model.hit_coarse_geo), flat tables, imsz = kept rays per view
    (lib/dvgo.py:643-680; 64 image rows per hit test as there)."""
    assert len(rgb_tr_ori) == len(train_poses) == len(Ks) == len(HW)
    dev = rgb_tr_ori[0].device
    rows = 64
    cols = ([], [], [], [])
    imsz = []
    total = 0
    for c2w, img, hw, K in zip(train_poses, rgb_tr_ori, HW, Ks):
        assert tuple(img.shape[:2]) == (int(hw[0]), int(hw[1]))
        ro, rd, vd = _rays_of(hw, K, c2w, ndc, inverse_y, flip_x, flip_y, dev)
        keep = torch.cat([model.hit_coarse_geo(rays_o=ro[i:i + rows], rays_d=rd[i:i + rows], **render_kwargs).to(dev)
                          for i in range(0, img.shape[0], rows)], 0)
        for lst, t in zip(cols, (img, ro, rd, vd)):
            lst.append(t[keep].float())
        imsz.append(keep.sum())
        total += img.shape[0] * img.shape[1]
    rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr = (torch.cat(c, 0) for c in cols)
    print('get_training_rays_in_maskcache_sampling: ratio', rgb_tr.shape[0] / max(total, 1))
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz


def batch_indices_generator(N, BS):
    """Endless batches of BS indices out of N, a fresh host-side permutation whenever fewer than BS are left (lib/dvgo.py:761-768)."""
    perm, pos = torch.from_numpy(np.random.permutation(N)).long(), 0
    while True:
        if pos + BS > N:
            perm, pos = torch.from_numpy(np.random.permutation(N)).long(), 0
        yield perm[pos:pos + BS]
        pos += BS


def patch_gen(imsz, num_im, BS, sz_patch):
    """The patch table of the 'patch_mimg' ray sampler (lib/dvgo.py:822-850): the image is cut in bs x bs tiles, bs = BS // sz_patch
    (4096 // 64 = 64 for configs/llff/fern_lg_joint_l1.py).  Returns a list of [rows, cols, 2] int64 arrays of (row, col) pixel indices
    in the reference's order: the full tiles column band by column band, then the right-edge remainders (one per row band), then the
    bottom-edge remainders (one per column band, then the corner).  As upstream, an exactly divisible side yields EMPTY remainder
    entries (the list length, which seeds the permutation, is kept)."""
    bs = BS // sz_patch
    H, W = int(imsz[0]), int(imsz[1])
    nr, nc = H // bs, W // bs

    def patch(r0, r1, c0, c1):
        rr, cc = np.meshgrid(np.arange(r0, r1, dtype=np.int64), np.arange(c0, c1, dtype=np.int64), indexing='ij')
        return np.stack((rr, cc), axis=-1)

    full = [patch(rb * bs, (rb + 1) * bs, cb * bs, (cb + 1) * bs) for cb in range(nc) for rb in range(nr)]
    right = [patch(rb * bs, (rb + 1) * bs, nc * bs, W) for rb in range(nr)]
    bottom = [patch(nr * bs, H, cb * bs, (cb + 1) * bs) for cb in range(nc)] + [patch(nr * bs, H, nc * bs, W)]
    return full + right + bottom


def mimg_patch_indices_generator(imsz, num_im, BS, sz_patch, sr_ratio):
    """Endless (image, LR patch, matching HR patch) choices for the joint training loop (lib/dvgo.py:852-880, used at
    run_sr.py:753-754,828-835): every (image, patch) pair once per epoch in a host-side random order.
    Yields ``image, rows, cols, rows_4x, cols_4x, [pr, pc]`` (flattened pixel index lists, patch height / width)."""
    imsz = np.asarray(imsz)
    lr, hr = patch_gen(imsz, num_im, BS, sz_patch), patch_gen(imsz * sr_ratio, num_im, BS * sr_ratio, sz_patch)
    num_p = len(lr)
    pairs = np.stack((np.repeat(np.arange(num_im, dtype=np.float64), num_p), np.tile(np.arange(num_p, dtype=np.float64), num_im)), axis=1)
    order, pos = torch.LongTensor(np.random.permutation(pairs)), 0
    while True:
        if pos >= len(pairs):
            order, pos = torch.LongTensor(np.random.permutation(pairs)), 0
        image, p = order[pos][0], int(order[pos][1])
        pos += 1
        pr, pc = lr[p].shape[0], lr[p].shape[1]
        a, b = lr[p].reshape(-1, 2), hr[p].reshape(-1, 2)
        yield image, list(a[:, 0]), list(a[:, 1]), list(b[:, 0]), list(b[:, 1]), [pr, pc]
