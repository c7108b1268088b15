"""Seeded synthetic scenes and camera paths in the reference's checkpoint format.

No datasets or trained checkpoints are available offline, so parity tests and
``bench.py`` run on scenes generated here.  The output is exactly what the reference stores in
a ``fine_last.tar`` (run_sr.py:1173-1178): ``{'model_kwargs': ..., 'model_state_dict': ...}``
with the reference's key names, so ``utils.load_model`` semantics (lib/utils.py:62-66) are
exercised by both the product modules and the oracle.

  * ``make_llff_checkpoint``  -- DirectMPIGO (NDC) scene, BASELINE configs 2-4
  * ``make_lego_checkpoint``  -- DirectVoxGO bounded scene, BASELINE config 1
  * ``llff_spiral_poses`` / ``lego_pose`` / ``LLFF_K`` -- synthetic cameras
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LLFF_HW = (756, 1008)                                  # configs/llff/llff_default_lg.py:8-9
LLFF_K = np.array([[815., 0, 504.], [0, 815., 378.], [0, 0, 1.]], dtype=np.float32)
LLFF_BBOX = ([-1.3, -1.1, -1.0], [1.3, 1.1, 1.0])


def _gen(seed):
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    return g


def _axis(lo, hi, n):
    return torch.linspace(float(lo), float(hi), int(n))


def _blob_field(xyz_min, xyz_max, world_size, g, n_blobs, amp, sigma_range, sheet=True, sheet_z=0.72, sheet_amp=None, sheet_vox=1.5):
    """Sum of anisotropic Gaussian blobs (+ one slanted sheet: thin and translucent by default; ``sheet_amp`` / ``sheet_vox`` / ``sheet_z``
    make it an opaque wall somewhere else in depth) evaluated on the voxel grid.
    Separable per blob, so the full-size LLFF grid (37.7 M voxels) takes a few seconds on CPU."""
    X, Y, Z = [int(v) for v in world_size]
    ax = [_axis(xyz_min[i], xyz_max[i], world_size[i]) for i in range(3)]
    ext = [float(xyz_max[i] - xyz_min[i]) for i in range(3)]
    field = torch.zeros([X, Y, Z])
    for _ in range(n_blobs):
        c = [float(xyz_min[i] + ext[i] * (0.12 + 0.76 * torch.rand([], generator=g))) for i in range(3)]
        s = [float(ext[i] * (sigma_range[0] + (sigma_range[1] - sigma_range[0]) * torch.rand([], generator=g)))
             for i in range(3)]
        a = float(amp * (0.75 + 0.5 * torch.rand([], generator=g)))
        gx = torch.exp(-0.5 * ((ax[0] - c[0]) / s[0]) ** 2)
        gy = torch.exp(-0.5 * ((ax[1] - c[1]) / s[1]) ** 2)
        gz = torch.exp(-0.5 * ((ax[2] - c[2]) / s[2]) ** 2)
        field += a * gx[:, None, None] * gy[None, :, None] * gz[None, None, :]
    if sheet:
        # slanted plane  z = z0 + sx*x + sy*y, thickness ~1.5 voxels in z
        z0 = float(xyz_min[2] + ext[2] * sheet_z)
        sx, sy = 0.11, -0.07
        th = sheet_vox * ext[2] / Z
        zc = z0 + sx * ax[0][:, None] + sy * ax[1][None, :]
        field += (amp if sheet_amp is None else sheet_amp) * torch.exp(-0.5 * ((ax[2][None, None, :] - zc[:, :, None]) / th) ** 2)
    return field


def _smooth3(x):
    """3-tap box filter along the three spatial axes (keeps shape) so trilinear weights matter."""
    return F.avg_pool3d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)


def _linear_init(g, fan_out, fan_in):
    """nn.Linear default init (kaiming_uniform a=sqrt(5) -> U(+-1/sqrt(fan_in))) from our own generator."""
    bound = 1.0 / math.sqrt(fan_in)
    w = (torch.rand([fan_out, fan_in], generator=g) * 2 - 1) * bound
    b = (torch.rand([fan_out], generator=g) * 2 - 1) * bound
    return w, b


def _rgbnet_state(g, dim0, width, depth, gain=1.0):
    """Keys of ``Sequential(Linear, act, *[Sequential(Linear, act)]*(depth-2), Linear)``
    (lib/dmpigo.py:112-120, lib/dvgo.py:116-124); last bias 0 (lib/dmpigo.py:120)."""
    sd = {}
    w, b = _linear_init(g, width, dim0)
    sd['rgbnet.0.weight'], sd['rgbnet.0.bias'] = w * gain, b
    for i in range(depth - 2):
        w, b = _linear_init(g, width, width)
        sd[f'rgbnet.{2 + i}.0.weight'], sd[f'rgbnet.{2 + i}.0.bias'] = w * gain, b
    w, b = _linear_init(g, 3, width)
    last = depth  # index of the final Linear inside the Sequential
    sd[f'rgbnet.{last}.weight'], sd[f'rgbnet.{last}.bias'] = w * gain, torch.zeros(3)
    return sd


def mpi_act_shift(mpi_depth, voxel_size_ratio):
    """Per-plane density bias of DirectMPIGO (lib/dmpigo.py:53-58)."""
    g = np.full([mpi_depth], 1. / mpi_depth - 1e-6)
    p = [1 - g[0]]
    for i in range(1, len(g)):
        p.append((1 - g[:i + 1].sum()) / (1 - g[:i].sum()))
    out = torch.zeros([1, 1, 1, 1, mpi_depth])
    for i in range(len(p)):
        out[..., i].fill_(np.log(p[i] ** (-1 / voxel_size_ratio) - 1))
    return out


def _raw2alpha(density, shift, interval):
    return 1 - torch.pow(1 + torch.exp(density + shift), -interval)


def make_llff_checkpoint(seed=777, num_voxels=384 * 384 * 256, mpi_depth=256, rgbnet_dim=9,
                         rgbnet_width=64, rgbnet_depth=3, viewbase_pe=0, spatial_pe=0,
                         stepsize=1.0, bbox=LLFF_BBOX, n_blobs=24, mask_margin=5.5,
                         mask_cache_world_size=None, opaque=False):
    """DirectMPIGO checkpoint with the LLFF configuration of configs/llff/llff_default_lg.py:33-44
    (defaults) or a scaled-down version of it (smaller ``num_voxels`` / ``mpi_depth``).
    ``opaque=True``: the statistics of a TRAINED forward-facing scene instead of translucent blobs -- an opaque slanted wall around 0.4
    of the depth range (density +22 over three planes: alpha = 1 to fp32) behind the front blobs, so that every ray reaches the
    T < 1e-3 stop of Alphas2Weights (render_utils_kernel.cu:597-600) by mid-depth and the blobs behind the wall are never seen."""
    g = _gen(seed)
    xyz_min = torch.tensor(bbox[0], dtype=torch.float32)
    xyz_max = torch.tensor(bbox[1], dtype=torch.float32)
    # DirectMPIGO._set_grid_resolution (lib/dmpigo.py:156-164)
    r = (num_voxels / mpi_depth / (xyz_max - xyz_min)[:2].prod()).sqrt()
    world_size = torch.zeros(3, dtype=torch.long)
    world_size[:2] = (xyz_max - xyz_min)[:2] * r
    world_size[2] = mpi_depth
    voxel_size_ratio = 256. / mpi_depth
    fast_color_thres = stepsize / mpi_depth / 5                       # llff_default_lg.py:43
    ws = world_size.tolist()

    field = _blob_field(xyz_min, xyz_max, ws, g, n_blobs, amp=16.0, sigma_range=(0.028, 0.075),
                        **(dict(sheet_z=0.4, sheet_amp=32.0, sheet_vox=3.0) if opaque else {}))
    density = (field - 10.0)[None, None].contiguous()                 # background -10
    act_shift = mpi_act_shift(mpi_depth, voxel_size_ratio)

    # occupancy mask as the reference derives it (lib/dmpigo.py:221-224): maxpool3(alpha) > thres,
    # taken on a looser (earlier-in-training) density so that it over-covers like a real mask_cache
    alpha_loose = _raw2alpha(density + mask_margin + act_shift, 0, voxel_size_ratio)
    mask = (F.max_pool3d(alpha_loose, kernel_size=3, padding=1, stride=1)[0, 0] > fast_color_thres)
    del alpha_loose
    if mask_cache_world_size is not None:
        idx = [torch.linspace(0, ws[i] - 1, int(mask_cache_world_size[i])).round().long() for i in range(3)]
        mask = mask[idx[0]][:, idx[1]][:, :, idx[2]].contiguous()

    k0 = torch.randn([1, rgbnet_dim if rgbnet_dim > 0 else 3] + ws, generator=g) * 0.5
    k0 = _smooth3(k0).contiguous()

    sd = {
        'xyz_min': xyz_min.clone(), 'xyz_max': xyz_max.clone(),
        'density.grid': density, 'density.xyz_min': xyz_min.clone(), 'density.xyz_max': xyz_max.clone(),
        'act_shift.grid': act_shift, 'act_shift.xyz_min': xyz_min.clone(), 'act_shift.xyz_max': xyz_max.clone(),
        'k0.grid': k0, 'k0.xyz_min': xyz_min.clone(), 'k0.xyz_max': xyz_max.clone(),
    }
    if rgbnet_dim > 0:
        sd['viewfreq'] = torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)])
        sd['posfreq'] = torch.FloatTensor([(2 ** i) for i in range(spatial_pe)])
        dim0 = (3 + 3 * viewbase_pe * 2 + 3 + 3 * spatial_pe * 2) + rgbnet_dim     # lib/dmpigo.py:85
        sd.update(_rgbnet_state(g, dim0, rgbnet_width, rgbnet_depth, gain=2.0))
    scale = (torch.tensor(list(mask.shape), dtype=torch.float32) - 1) / (xyz_max - xyz_min)
    sd['mask_cache.mask'] = mask
    sd['mask_cache.xyz2ijk_scale'] = scale                            # lib/grid.py:291-293
    sd['mask_cache.xyz2ijk_shift'] = -xyz_min * scale

    kwargs = {
        'xyz_min': xyz_min.numpy().copy(), 'xyz_max': xyz_max.numpy().copy(),
        'num_voxels': num_voxels, 'mpi_depth': mpi_depth, 'voxel_size_ratio': voxel_size_ratio,
        'mask_cache_path': None, 'mask_cache_thres': 1e-3,
        'mask_cache_world_size': list(mask.shape),
        'fast_color_thres': fast_color_thres,
        'density_type': 'DenseGrid', 'k0_type': 'DenseGrid', 'density_config': {}, 'k0_config': {},
        'mode_type': 'mlp', 'act_type': 'relu', 'dim_rend': 3,
        'rgbnet_dim': rgbnet_dim, 'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width,
        'viewbase_pe': viewbase_pe, 'spatial_pe': spatial_pe,
    }
    render_kwargs = {'near': 0, 'far': 1, 'bg': 0, 'stepsize': stepsize, 'inverse_y': False,
                     'flip_x': False, 'flip_y': False, 'render_depth': True}
    return {'global_step': 0, 'model_kwargs': kwargs, 'model_state_dict': sd,
            'model_class': 'DirectMPIGO', 'render_kwargs': render_kwargs}


def make_lego_checkpoint(seed=777, num_voxels=160 ** 3, rgbnet_dim=12, rgbnet_width=128, rgbnet_depth=3,
                         viewbase_pe=4, rgbnet_direct=True, alpha_init=1e-2, fast_color_thres=1e-4,
                         stepsize=0.5, n_blobs=10, world_bound_scale=1.05):
    """DirectVoxGO fine-stage checkpoint in the nerf_synthetic configuration
    (configs/default.py:107-119, configs/syn/syn_default.py, lib/load_data.py:58)."""
    g = _gen(seed)
    half = 1.5 * world_bound_scale
    xyz_min = torch.tensor([-half] * 3, dtype=torch.float32)
    xyz_max = torch.tensor([half] * 3, dtype=torch.float32)
    # DirectVoxGO._set_grid_resolution (lib/dvgo.py:152-158)
    voxel_size = ((xyz_max - xyz_min).prod() / num_voxels).pow(1 / 3)
    world_size = ((xyz_max - xyz_min) / voxel_size).long()
    ws = world_size.tolist()
    act_shift = torch.FloatTensor([np.log(1 / (1 - alpha_init) - 1)])  # lib/dvgo.py:46

    field = _blob_field(xyz_min, xyz_max, ws, g, n_blobs, amp=22.0, sigma_range=(0.05, 0.12), sheet=False)
    density = (field - 8.0)[None, None].contiguous()
    alpha_loose = _raw2alpha(density + 2.0, act_shift, 1.0)
    mask = F.max_pool3d(alpha_loose, kernel_size=3, padding=1, stride=1)[0, 0] > fast_color_thres
    del alpha_loose

    k0_dim = rgbnet_dim if rgbnet_dim > 0 else 3
    k0 = _smooth3(torch.randn([1, k0_dim] + ws, generator=g) * 0.5).contiguous()
    sd = {
        'xyz_min': xyz_min.clone(), 'xyz_max': xyz_max.clone(), 'act_shift': act_shift,
        'density.grid': density, 'density.xyz_min': xyz_min.clone(), 'density.xyz_max': xyz_max.clone(),
        'k0.grid': k0, 'k0.xyz_min': xyz_min.clone(), 'k0.xyz_max': xyz_max.clone(),
    }
    if rgbnet_dim > 0:
        sd['viewfreq'] = torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)])
        dim0 = 3 + 3 * viewbase_pe * 2 + (k0_dim if rgbnet_direct else k0_dim - 3)   # lib/dvgo.py:94-101
        sd.update(_rgbnet_state(g, dim0, rgbnet_width, rgbnet_depth, gain=1.5))
    scale = (torch.tensor(list(mask.shape), dtype=torch.float32) - 1) / (xyz_max - xyz_min)
    sd['mask_cache.mask'] = mask
    sd['mask_cache.xyz2ijk_scale'] = scale
    sd['mask_cache.xyz2ijk_shift'] = -xyz_min * scale
    kwargs = {
        'xyz_min': xyz_min.numpy().copy(), 'xyz_max': xyz_max.numpy().copy(),
        'num_voxels': num_voxels, 'num_voxels_base': num_voxels, 'alpha_init': alpha_init,
        'voxel_size_ratio': 1.0, 'mask_cache_path': None, 'mask_cache_thres': 1e-3,
        'mask_cache_world_size': list(mask.shape), 'fast_color_thres': fast_color_thres,
        'density_type': 'DenseGrid', 'k0_type': 'DenseGrid', 'density_config': {}, 'k0_config': {},
        'mode_type': 'mlp', 'act_type': 'mlp', 'dim_rend': 3,
        'rgbnet_dim': rgbnet_dim, 'rgbnet_direct': rgbnet_direct, 'rgbnet_full_implicit': False,
        'rgbnet_depth': rgbnet_depth, 'rgbnet_width': rgbnet_width, 'viewbase_pe': viewbase_pe,
    }
    render_kwargs = {'near': 2., 'far': 6., 'bg': 1, 'stepsize': stepsize, 'inverse_y': False,
                     'flip_x': False, 'flip_y': False, 'render_depth': True}
    return {'global_step': 0, 'model_kwargs': kwargs, 'model_state_dict': sd,
            'model_class': 'DirectVoxGO', 'render_kwargs': render_kwargs}


# ---------------------------------------------------------------------------
# cameras
# ---------------------------------------------------------------------------
def _normalize(v):
    return v / np.linalg.norm(v)


def _viewmatrix(z, up, pos):
    vec2 = _normalize(z)
    vec0 = _normalize(np.cross(up, vec2))
    vec1 = _normalize(np.cross(vec2, vec0))
    return np.stack([vec0, vec1, vec2, pos], 1)


def llff_spiral_poses(n_frames=20, rads=(0.3, 0.2, 0.1), focal=3.0, zdelta=0.2, zrate=0.5, rots=2):
    """LLFF-style spiral around the recentred average pose (identity), same construction as the
    reference's render path (lib/load_llff.py:238-249).  -> [n,3,4] float32 c2w."""
    c2w = np.concatenate([np.eye(3), np.zeros([3, 1])], 1)
    up = np.array([0., 1., 0.])
    r = np.array(list(rads) + [1.])
    poses = []
    for theta in np.linspace(0., 2 * np.pi * rots, n_frames + 1)[:-1]:
        c = c2w @ (np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate) * zdelta, 1.]) * r)
        z = _normalize(c - c2w @ np.array([0, 0, -focal, 1.]))
        poses.append(_viewmatrix(z, up, c))
    return np.stack(poses, 0).astype(np.float32)


def lego_pose(theta_deg=30., phi_deg=-30., radius=4.0):
    """``pose_spherical`` of the blender loader (lib/load_blender.py:29-34). -> [4,4] float32."""
    def trans_t(t):
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], dtype=np.float64)

    def rot_phi(phi):
        return np.array([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0],
                         [0, np.sin(phi), np.cos(phi), 0], [0, 0, 0, 1]], dtype=np.float64)

    def rot_theta(th):
        return np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0],
                         [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], dtype=np.float64)
    c2w = trans_t(radius)
    c2w = rot_phi(phi_deg / 180. * np.pi) @ c2w
    c2w = rot_theta(theta_deg / 180. * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64) @ c2w
    return c2w.astype(np.float32)


def lego_K(H, W, camera_angle_x=0.6911112070083618):
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
