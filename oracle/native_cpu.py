"""CPU restatement of the reference's ``render_utils_cuda`` extension.  TEST INFRASTRUCTURE ONLY.

Every function restates one pybind entry point of
``/root/reference/lib/cuda/render_utils.cpp:170-184`` (kernels in
``lib/cuda/render_utils_kernel.cu``), same names, same argument order, same
returned tuple, on CPU tensors.  All arithmetic is fp32 as in the kernels (they
read ``scalar_t`` but compute in ``float``).

nvcc contracts ``a*b+c`` into one FMA by default (``-fmad=true``); the kernels'
``o + d*t`` and ``x*scale + shift`` are therefore single-rounding.  ``_fma`` emulates
that through fp64 (the 24x24-bit product is exact in fp64; the residual
double-rounding case has probability ~2^-29 per op).

Pin: the reference ships no CPU path, tests or golden vectors for these kernels (SURVEY.md
section 4 / 8c), so the pin is made from the reference ITSELF: its ``lib/cuda/render_utils*.cu|.cpp``
are compiled for gfx950 by ``oracle/build_ref.py`` (-> ``oracle/_ref/``, the same
``torch.utils.cpp_extension.load`` call the reference makes), run on an MI355X by
``oracle/gen_native_golden.py`` and the outputs committed as ``tests/golden/native_*.npz``;
``tests/test_oracle_golden.py::test_native_cpu_matches_reference_compiled_kernels`` holds this file to
them (integer / boolean outputs and the transmittance scan bit-exact, a few ulp elsewhere).
"""
import torch

__all__ = [
    'infer_t_minmax', 'infer_n_samples', 'infer_ray_start_dir',
    'sample_pts_on_rays', 'sample_ndc_pts_on_rays', 'sample_bg_pts_on_rays',
    'maskcache_lookup', 'raw2alpha', 'raw2alpha_backward',
    'raw2alpha_nonuni', 'raw2alpha_nonuni_backward',
    'alpha2weight', 'alpha2weight_backward',
]


def _fma(a, b, c):
    return (a.double() * b.double() + c.double()).float()


def _round_half_away(x):
    """C ``round()``: halves away from zero (render_utils_kernel.cu:385-387);
    ``torch.round`` is half-to-even, so it cannot be used here."""
    return torch.sign(x) * torch.floor(torch.abs(x) + 0.5)


def _f(x):
    return float(x)


# ---------------------------------------------------------------------------
# ray-AABB sampler (DirectVoxGO)           render_utils_kernel.cu:12-242
# ---------------------------------------------------------------------------
def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    """render_utils_kernel.cu:12-35.  Zero direction components become 1e-6."""
    near = torch.tensor(_f(near), dtype=torch.float32)
    far = torch.tensor(_f(far), dtype=torch.float32)
    v = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    a = (xyz_max - rays_o) / v
    b = (xyz_min - rays_o) / v
    lo = torch.minimum(a, b)
    hi = torch.maximum(a, b)
    t_min = torch.maximum(torch.maximum(lo[:, 0], lo[:, 1]), lo[:, 2])
    t_max = torch.minimum(torch.minimum(hi[:, 0], hi[:, 1]), hi[:, 2])
    t_min = torch.maximum(torch.minimum(t_min, far), near)
    t_max = torch.maximum(torch.minimum(t_max, far), near)
    return t_min, t_max


def _rnorm(rays_d):
    # sqrt(dx*dx + dy*dy + dz*dz), contracted as fma(dz,dz,fma(dy,dy,dx*dx))
    s = rays_d[:, 0] * rays_d[:, 0]
    s = _fma(rays_d[:, 1], rays_d[:, 1], s)
    s = _fma(rays_d[:, 2], rays_d[:, 2], s)
    return torch.sqrt(s)


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    """render_utils_kernel.cu:38-55.  At least one sample per ray (:53)."""
    rnorm = _rnorm(rays_d)
    sd = torch.tensor(_f(stepdist), dtype=torch.float32)
    n = torch.ceil((t_max - t_min) * rnorm / sd)
    return torch.clamp(n, min=1.).to(torch.int64)


def infer_ray_start_dir(rays_o, rays_d, t_min):
    """render_utils_kernel.cu:58-79."""
    rnorm = _rnorm(rays_d)
    rays_start = _fma(rays_d, t_min[:, None], rays_o)
    rays_dir = rays_d / rnorm[:, None]
    return rays_start, rays_dir


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """render_utils_kernel.cu:196-242.
    -> rays_pts [M,3], mask_outbbox [M], ray_id [M] i64, step_id [M] i64,
       N_steps [N] i64, t_min [N], t_max [N]"""
    n_rays = rays_o.shape[0]
    t_min, t_max = infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far)
    N_steps = infer_n_samples(rays_d, t_min, t_max, stepdist)
    ray_id = torch.repeat_interleave(torch.arange(n_rays, dtype=torch.int64), N_steps)
    cum = N_steps.cumsum(0)
    seg_start = cum - N_steps
    step_id = torch.arange(ray_id.numel(), dtype=torch.int64) - seg_start[ray_id]
    rays_start, rays_dir = infer_ray_start_dir(rays_o, rays_d, t_min)
    sd = torch.tensor(_f(stepdist), dtype=torch.float32)
    dist = sd * step_id.float()                                   # :184
    pts = _fma(rays_dir[ray_id], dist[:, None], rays_start[ray_id])  # :185-187
    mask_outbbox = ((xyz_min > pts) | (xyz_max < pts)).any(-1)    # :191-192
    return pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max


# ---------------------------------------------------------------------------
# NDC sampler (DirectMPIGO)                render_utils_kernel.cu:245-293
# ---------------------------------------------------------------------------
def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    """-> rays_pts [N,Ns,3], mask_outbbox [N,Ns]"""
    N_samples = int(N_samples)
    step = torch.arange(N_samples, dtype=torch.float32)
    dist = step / torch.tensor(float(N_samples - 1), dtype=torch.float32)   # :260
    pts = _fma(rays_d[:, None, :], dist[None, :, None], rays_o[:, None, :])  # :261-263
    mask_outbbox = ((xyz_min > pts) | (xyz_max < pts)).any(-1)            # :267-268
    return pts, mask_outbbox


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    """render_utils_kernel.cu:301-340 (unbounded scenes; out of the hot-path scope,
    restated so the shim exports all 13 names)."""
    N_samples = int(N_samples)
    step = torch.arange(N_samples, dtype=torch.float32)
    t_outer = t_max[:, None] - 1. + 1. / (1. - step[None] / N_samples)
    p = rays_o[:, None] + rays_d[:, None] * t_outer[..., None]
    t = p.norm(dim=-1)
    m = p.abs().amax(-1)
    R = t / m
    o2i = R * R / (t * t) * (1. - bg_preserve) + R / t * bg_preserve
    return p * o2i[..., None]


# ---------------------------------------------------------------------------
# occupancy lookup                         render_utils_kernel.cu:374-424
# ---------------------------------------------------------------------------
def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    """Nearest-voxel bool lookup; out-of-range -> False (out zero-initialised :405)."""
    n = xyz.shape[0]
    out = torch.zeros([n], dtype=torch.bool)
    if n == 0:
        return out
    ijk = _round_half_away(_fma(xyz, xyz2ijk_scale, xyz2ijk_shift))
    sz = torch.tensor(world.shape, dtype=torch.float32)
    ok = ((ijk >= 0) & (ijk < sz)).all(-1)
    ijk = ijk[ok].long()
    out[ok] = world[ijk[:, 0], ijk[:, 1], ijk[:, 2]]
    return out


# ---------------------------------------------------------------------------
# density activation                        render_utils_kernel.cu:431-574
# ---------------------------------------------------------------------------
def raw2alpha(density, shift, interval):
    """e = exp(d+shift) (may be inf); alpha = 1-(1+e)^(-interval)   (:439-441)"""
    e = torch.exp(density + _f(shift))
    alpha = 1 - torch.pow(1 + e, -_f(interval))
    return e, alpha


def raw2alpha_nonuni(density, shift, interval):
    e = torch.exp(density + _f(shift))
    alpha = 1 - torch.pow(1 + e, -interval)
    return e, alpha


def raw2alpha_backward(exp_d, grad_back, interval):
    """render_utils_kernel.cu:507-517"""
    interval = _f(interval)
    return exp_d.clamp(max=1e10) * torch.pow(1 + exp_d, -interval - 1) * interval * grad_back


def raw2alpha_nonuni_backward(exp_d, grad_back, interval):
    return exp_d.clamp(max=1e10) * torch.pow(1 + exp_d, -interval - 1) * interval * grad_back


# ---------------------------------------------------------------------------
# transmittance scan                        render_utils_kernel.cu:577-707
# ---------------------------------------------------------------------------
def _segments(ray_id, n_rays):
    """i_start/i_end exactly as __set_i_for_segment_start_end (:607-617) plus the
    host-side ``i_end[ray_id[n-1]] = n`` (:635) leave them: rays without points keep 0/0."""
    n = ray_id.numel()
    i_start = torch.zeros([n_rays], dtype=torch.int64)
    i_end = torch.zeros([n_rays], dtype=torch.int64)
    if n == 0:
        return i_start, i_end
    chg = torch.nonzero(ray_id[1:] != ray_id[:-1]).flatten() + 1
    i_start[ray_id[chg]] = chg
    i_end[ray_id[chg - 1]] = chg
    i_end[ray_id[n - 1]] = n
    return i_start, i_end


def alpha2weight(alpha, ray_id, n_rays):
    """Per-ray SEQUENTIAL scan (:591-603):
         T[i]=Tc; w[i]=Tc*alpha[i]; Tc=(float)((double)Tc*(1.-alpha[i])); if Tc<1e-3: stop after i.
    Points after the stop keep weight 0 / T 1 (zeros_like / ones_like :624-625) and
    i_end is shrunk to the stop (:602).  Vectorised over rays, sequential over steps."""
    n_rays = int(n_rays)
    n = alpha.numel()
    weight = torch.zeros_like(alpha)
    T = torch.ones_like(alpha)
    alphainv_last = torch.ones([n_rays], dtype=alpha.dtype)
    i_start, i_end = _segments(ray_id, n_rays)
    if n == 0:
        return weight, T, alphainv_last, i_start, i_end
    seg_len = i_end - i_start
    max_len = int(seg_len.max())
    Tc = torch.ones([n_rays], dtype=torch.float32)
    alive = seg_len > 0
    new_end = i_end.clone()
    for s in range(max_len):
        act = alive & (s < seg_len)
        idx = (i_start + s)[act]
        a = alpha[idx]
        T[idx] = Tc[act]
        weight[idx] = Tc[act] * a
        Tn = (Tc[act].double() * (1. - a.double())).float()       # `T_cum *= (1. - alpha[i])`
        Tc[act] = Tn
        stop = Tn.double() < 1e-3
        rid = torch.nonzero(act).flatten()
        new_end[rid[stop]] = idx[stop] + 1
        alive[rid[stop]] = False
    return weight, T, Tc.clone(), i_start, new_end


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays,
                          grad_weights, grad_last):
    """render_utils_kernel.cu:654-677 (reverse per-ray scan)."""
    grad = torch.zeros_like(alpha)
    for r in range(int(n_rays)):
        s, e = int(i_start[r]), int(i_end[r])
        back = float(grad_last[r]) * float(alphainv_last[r])
        for i in range(e - 1, s - 1, -1):
            grad[i] = float(grad_weights[i]) * float(T[i]) - back / (1 - float(alpha[i]) + 1e-10)
            back += float(grad_weights[i]) * float(weight[i])
    return grad
