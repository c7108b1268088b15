"""CPU oracle for the training-step streaming kernels.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy, the reference's CUDA kernels (which cannot run here: no CUDA device, and the reference has no
CPU implementation or test vectors for them):

* ``adam_upd`` / ``masked_adam_upd`` / ``adam_upd_with_perlr`` -- lib/cuda/adam_upd_kernel.cu:8-58 (kernels) and
  :60-133 (host wrappers, `step_size`), driven by ``MaskedAdam.step`` (lib/masked_adam.py:39-71);
* ``total_variation_add_grad`` -- lib/cuda/total_variation_kernel.cu:13-66.

nvcc contracts ``a*b + c`` to one FMA; the products of two floats are exact in float64, so an FMA is emulated as
``float32(float64(a)*float64(b) + float64(c))`` (double rounding can differ from a true FMA in the last bit with
probability ~2^-29 per op; the tests use a 2-ulp tolerance).

Pinning: tests/golden/optim_ref.npz holds outputs of the reference's own adam_upd_cuda / total_variation_cuda extensions compiled
for gfx950 (oracle/build_ref.py) and run on an MI355X (oracle/gen_native_golden.py); tests/test_oracle_golden.py checks this file
against them.  tests/test_optim_oracle.py additionally pins the restatement against independent PyTorch formulations: torch.optim.Adam (identical up to where eps enters the denominator, so compared with
eps -> 0) and autograd of a Huber (smooth-L1, beta=1) neighbour loss whose gradient the TV kernel is.
"""
import numpy as np

f32 = np.float32
f64 = np.float64


def _fma(a, b, c):
    return (a.astype(f64) * f64(b) if np.isscalar(b) else a.astype(f64) * b.astype(f64)) + c.astype(f64)


def adam_step_size(lr, beta1, beta2, step):
    """adam_upd_kernel.cu:71 -- all float arithmetic."""
    lr, beta1, beta2 = f32(lr), f32(beta1), f32(beta2)
    p2 = f32(np.power(beta2, f32(step), dtype=f32))
    p1 = f32(np.power(beta1, f32(step), dtype=f32))
    return f32(f32(lr * f32(np.sqrt(f32(f32(1) - p2)))) / f32(f32(1) - p1))


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps, perlr=None, masked=False):
    """Returns the updated (param, exp_avg, exp_avg_sq); inputs are not modified.  Shapes arbitrary, fp32."""
    p = np.asarray(param, f32).copy()
    g = np.asarray(grad, f32)
    m = np.asarray(exp_avg, f32).copy()
    v = np.asarray(exp_avg_sq, f32).copy()
    b1, b2, eps = f32(beta1), f32(beta2), f32(eps)
    omb1, omb2 = f32(f32(1) - b1), f32(f32(1) - b2)
    ss = adam_step_size(lr, beta1, beta2, step)
    sel = (g != 0) if masked else np.ones(g.shape, bool)                       # .cu:34
    gs = g[sel]
    m_new = (m[sel].astype(f64) * f64(b1) + (omb1 * gs).astype(f64)).astype(f32)   # .cu:19
    v_new = (v[sel].astype(f64) * f64(b2) + ((omb2 * gs).astype(f32) * gs).astype(f64)).astype(f32)  # .cu:20
    num = (ss * m_new).astype(f32)
    if perlr is not None:                                                       # .cu:56: step_size*perlr*exp_avg
        num = ((ss * np.asarray(perlr, f32)[sel]).astype(f32) * m_new).astype(f32)
    den = (np.sqrt(v_new).astype(f32) + eps).astype(f32)
    p[sel] = (p[sel] - (num / den).astype(f32)).astype(f32)                    # .cu:21
    m[sel] = m_new
    v[sel] = v_new
    return p, m, v


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    """param, grad: [1, C, sz_i, sz_j, sz_k] fp32.  Returns the new grad (input not modified).
    total_variation_kernel.cu:13-35 with the host scaling w/=6 of :46-48; wx acts on the LAST axis (k)."""
    p = np.asarray(param, f32)
    g = np.asarray(grad, f32).copy()
    assert p.ndim == 5 and p.shape == g.shape
    wx, wy, wz = f32(f32(wx) / f32(6)), f32(f32(wy) / f32(6)), f32(f32(wz) / f32(6))
    add = np.zeros(p.shape, f64)        # holds an fp32 value after every step

    def term(axis, w, lo):
        nonlocal add
        n = p.shape[axis]
        if n < 2:
            return
        a = [slice(None)] * 5
        b = [slice(None)] * 5
        if lo:      # neighbour at index-1 exists for idx >= 1
            a[axis], b[axis] = slice(1, n), slice(0, n - 1)
        else:       # neighbour at index+1 exists for idx <= n-2
            a[axis], b[axis] = slice(0, n - 1), slice(1, n)
        a, b = tuple(a), tuple(b)
        d = np.clip((p[a] - p[b]).astype(f32), f32(-1), f32(1))
        add[a] = (f64(w) * d.astype(f64) + add[a]).astype(f32)              # fma(w, clamp, acc)

    term(4, wx, True); term(4, wx, False)      # k-1, k+1   (.cu:28-29)
    term(3, wy, True); term(3, wy, False)      # j-1, j+1   (.cu:30-31)
    term(2, wz, True); term(2, wz, False)      # i-1, i+1   (.cu:32-33)
    sel = np.ones(g.shape, bool) if dense_mode else (g != 0)
    g[sel] = (g[sel] + add[sel].astype(f32)).astype(f32)
    return g
