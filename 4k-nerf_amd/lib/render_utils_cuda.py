"""Drop-in for the reference's JIT-built ``render_utils_cuda`` extension module.

Same 13 function names, argument order and returned tensors as the pybind11 module defined in
/root/reference/lib/cuda/render_utils.cpp:170-184, implemented on the gfx950 staged kernels of
``csrc/k4_staged.hip`` through the C ABI (``include/k4nerf.h``).  Differences, all deliberate:
  * kernels run on torch's CURRENT stream (the reference uses the legacy default stream);
  * launch errors are checked; inputs must be GPU + contiguous (same guard as render_utils.cpp:46-48);
  * ``sample_pts_on_rays`` still needs one host sync for the data-dependent total length (as the
    reference does at render_utils_kernel.cu:212) -- the fused marcher has none.
"""
import torch

from .. import _native as N

__all__ = ['infer_t_minmax', 'infer_n_samples', 'infer_ray_start_dir', 'sample_pts_on_rays',
           'sample_ndc_pts_on_rays', 'sample_bg_pts_on_rays', 'maskcache_lookup', 'raw2alpha',
           'raw2alpha_backward', 'raw2alpha_nonuni', 'raw2alpha_nonuni_backward', 'alpha2weight',
           'alpha2weight_backward']


def _f(x):
    return float(x)


def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    n = rays_o.shape[0]
    t_min = torch.empty([n], dtype=torch.float32, device=rays_o.device)
    t_max = torch.empty_like(t_min)
    N.check(N.lib().k4_infer_t_minmax(N.f32(rays_o), N.f32(rays_d), N.f32(xyz_min), N.f32(xyz_max),
                                      _f(near), _f(far), n, N.f32(t_min), N.f32(t_max), N.stream()), 'infer_t_minmax')
    return [t_min, t_max]


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    n = t_min.shape[0]
    out = torch.empty([n], dtype=torch.int64, device=rays_d.device)
    N.check(N.lib().k4_infer_n_samples(N.f32(rays_d), N.f32(t_min), N.f32(t_max), _f(stepdist), n,
                                       N.ptr(out), N.stream()), 'infer_n_samples')
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    start = torch.empty_like(rays_o)
    rdir = torch.empty_like(rays_o)
    N.check(N.lib().k4_infer_ray_start_dir(N.f32(rays_o), N.f32(rays_d), N.f32(t_min), rays_o.shape[0],
                                           N.f32(start), N.f32(rdir), N.stream()), 'infer_ray_start_dir')
    return [start, rdir]


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> [rays_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max]  (render_utils_kernel.cu:241)"""
    dev = rays_o.device
    n = rays_o.shape[0]
    N_steps = torch.empty([n], dtype=torch.int64, device=dev)
    t_min = torch.empty([n], dtype=torch.float32, device=dev)
    t_max = torch.empty_like(t_min)
    L = N.lib()
    N.check(L.k4_sample_pts_on_rays_count(N.f32(rays_o), N.f32(rays_d), N.f32(xyz_min), N.f32(xyz_max),
                                          _f(near), _f(far), _f(stepdist), n, N.ptr(N_steps), N.f32(t_min),
                                          N.f32(t_max), N.stream()), 'sample_pts_on_rays(count)')
    cum = N_steps.cumsum(0)
    total = int(cum[-1].item()) if n > 0 else 0
    pts = torch.empty([total, 3], dtype=torch.float32, device=dev)
    mask = torch.empty([total], dtype=torch.bool, device=dev)
    ray_id = torch.empty([total], dtype=torch.int64, device=dev)
    step_id = torch.empty([total], dtype=torch.int64, device=dev)
    N.check(L.k4_sample_pts_on_rays_fill(N.f32(rays_o), N.f32(rays_d), N.f32(xyz_min), N.f32(xyz_max),
                                         N.f32(t_min), N.ptr(cum), _f(stepdist), n, total, N.f32(pts),
                                         N.ptr(mask), N.ptr(ray_id), N.ptr(step_id), N.stream()),
            'sample_pts_on_rays(fill)')
    return [pts, mask, ray_id, step_id, N_steps, t_min, t_max]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    n = rays_o.shape[0]
    pts = torch.empty([n, N_samples, 3], dtype=torch.float32, device=rays_o.device)
    mask = torch.empty([n, N_samples], dtype=torch.bool, device=rays_o.device)
    N.check(N.lib().k4_sample_ndc_pts_on_rays(N.f32(rays_o), N.f32(rays_d), N.f32(xyz_min), N.f32(xyz_max),
                                              n, int(N_samples), N.f32(pts), N.ptr(mask), N.stream()),
            'sample_ndc_pts_on_rays')
    return [pts, mask]


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    raise NotImplementedError(
        'sample_bg_pts_on_rays serves the unbounded-scene model lib/dbvgo.py only, which no BASELINE '
        'configuration selects (SURVEY.md 2.2: OUT OF SCOPE)')


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    n = xyz.shape[0]
    out = torch.zeros([n], dtype=torch.bool, device=xyz.device)
    if n == 0:
        return out
    N.check(N.lib().k4_maskcache_lookup(N.ptr(world), N.f32(xyz), N.f32(xyz2ijk_scale), N.f32(xyz2ijk_shift),
                                        world.shape[0], world.shape[1], world.shape[2], n, N.ptr(out), N.stream()),
            'maskcache_lookup')
    return out


def _raw2alpha(density, shift, interval, ipp):
    exp_d = torch.empty_like(density)
    alpha = torch.empty_like(density)
    N.check(N.lib().k4_raw2alpha(N.f32(density), _f(shift), _f(interval), None if ipp is None else N.f32(ipp),
                                 density.shape[0], N.f32(exp_d), N.f32(alpha), N.stream()), 'raw2alpha')
    return [exp_d, alpha]


def raw2alpha(density, shift, interval):
    return _raw2alpha(density, shift, interval, None)


def raw2alpha_nonuni(density, shift, interval):
    return _raw2alpha(density, shift, 0.0, interval)


def _raw2alpha_bwd(exp_d, grad_back, interval, ipp):
    grad = torch.empty_like(exp_d)
    N.check(N.lib().k4_raw2alpha_backward(N.f32(exp_d), N.f32(grad_back), _f(interval),
                                          None if ipp is None else N.f32(ipp), exp_d.shape[0], N.f32(grad),
                                          N.stream()), 'raw2alpha_backward')
    return grad


def raw2alpha_backward(exp_d, grad_back, interval):
    return _raw2alpha_bwd(exp_d, grad_back, interval, None)


def raw2alpha_nonuni_backward(exp_d, grad_back, interval):
    return _raw2alpha_bwd(exp_d, grad_back, 0.0, interval)


def alpha2weight(alpha, ray_id, n_rays):
    """-> [weight, T, alphainv_last, i_start, i_end]"""
    dev = alpha.device
    n_rays = int(n_rays)
    weight = torch.empty_like(alpha)
    T = torch.empty_like(alpha)
    ainv = torch.empty([n_rays], dtype=alpha.dtype, device=dev)
    i_start = torch.empty([n_rays], dtype=torch.int64, device=dev)
    i_end = torch.empty([n_rays], dtype=torch.int64, device=dev)
    N.check(N.lib().k4_alpha2weight(N.f32(alpha), N.ptr(ray_id), alpha.shape[0], n_rays, N.f32(weight), N.f32(T),
                                    N.f32(ainv), N.ptr(i_start), N.ptr(i_end), N.stream()), 'alpha2weight')
    return [weight, T, ainv, i_start, i_end]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    grad = torch.empty_like(alpha)
    gw, gl = grad_weights.contiguous(), grad_last.contiguous()       # named: the copies must outlive the launch
    N.check(N.lib().k4_alpha2weight_backward(N.f32(alpha), N.f32(weight), N.f32(T), N.f32(alphainv_last),
                                             N.ptr(i_start), N.ptr(i_end), int(n_rays), alpha.shape[0],
                                             N.f32(gw), N.f32(gl), N.f32(grad), N.stream()), 'alpha2weight_backward')
    return grad
