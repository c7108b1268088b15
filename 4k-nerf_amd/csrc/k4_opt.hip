// Training-step streaming kernels for the voxel grids: the three Adam variants of the
// reference's `adam_upd_cuda` (lib/cuda/adam_upd_kernel.cu:8-58, host wrappers :60-133) and
// `total_variation_add_grad` (lib/cuda/total_variation_kernel.cu:13-66).  All are HBM-bound element-wise passes
// over grids of up to 4.5e8 floats (k0 at 417x353x256x12), so the design is about bytes, not flops:
//   * 128-bit accesses (one thread = 4 consecutive floats), scalar tail / scalar kernel for unaligned sizes;
//   * the masked variants test the gradient FIRST and touch param / moments only where some lane has grad != 0:
//     sparse-touched grids (a training batch of 8192 rays touches <1 % of k0) then cost 4 B/voxel instead of 28;
//   * the TV stencil is launched with the XCD-aware block map so that the i-1 / i / i+1 planes a workgroup reads
//     live in one XCD's L2 (a plane of the LLFF grid is 361 KB; the stencil re-reads come from L2, not HBM).
// Arithmetic follows the reference op for op (same association; a*b+c written as fmaf where nvcc contracts it).
#include "k4_common.h"
#include <math.h>
#include <stdlib.h>

#define K4_OPT_THREADS 256

enum { K4_ADAM_PLAIN = 0, K4_ADAM_MASKED = 1, K4_ADAM_PERLR = 2 };

__device__ __forceinline__ void k4_adam_one(float& p, float g, float& m, float& v, float lrs, float step_size,
                                            float beta1, float beta2, float omb1, float omb2, float eps) {
    m = fmaf(beta1, m, omb1 * g);                          // .cu:19 / :37 / :54
    v = fmaf(beta2, v, (omb2 * g) * g);                    // .cu:20 / :38 / :55
    p -= (step_size * lrs * m) / (sqrtf(v) + eps);         // .cu:21 / :39 / :56 (lrs = perlr or exactly 1)
}

template <int MODE>
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_adam_vec_kernel(
    float4* __restrict__ param, const float4* __restrict__ grad, float4* __restrict__ exp_avg,
    float4* __restrict__ exp_avg_sq, const float4* __restrict__ perlr, int64_t n4, float step_size, float beta1,
    float beta2, float eps) {
    const int64_t i = (int64_t)blockIdx.x * K4_OPT_THREADS + threadIdx.x;
    if (i >= n4) return;
    const float4 g = grad[i];
    if (MODE == K4_ADAM_MASKED && g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;
    float4 p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    float4 l = make_float4(1.f, 1.f, 1.f, 1.f);
    if (MODE == K4_ADAM_PERLR) l = perlr[i];
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    if (MODE != K4_ADAM_MASKED || g.x != 0.f) k4_adam_one(p.x, g.x, m.x, v.x, l.x, step_size, beta1, beta2, omb1, omb2, eps);
    if (MODE != K4_ADAM_MASKED || g.y != 0.f) k4_adam_one(p.y, g.y, m.y, v.y, l.y, step_size, beta1, beta2, omb1, omb2, eps);
    if (MODE != K4_ADAM_MASKED || g.z != 0.f) k4_adam_one(p.z, g.z, m.z, v.z, l.z, step_size, beta1, beta2, omb1, omb2, eps);
    if (MODE != K4_ADAM_MASKED || g.w != 0.f) k4_adam_one(p.w, g.w, m.w, v.w, l.w, step_size, beta1, beta2, omb1, omb2, eps);
    param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
}

template <int MODE>
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_adam_scalar_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
    float* __restrict__ exp_avg_sq, const float* __restrict__ perlr, int64_t first, int64_t n, float step_size,
    float beta1, float beta2, float eps) {
    const int64_t i = first + (int64_t)blockIdx.x * K4_OPT_THREADS + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i];
    if (MODE == K4_ADAM_MASKED && g == 0.f) return;
    float p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    const float l = MODE == K4_ADAM_PERLR ? perlr[i] : 1.f;
    k4_adam_one(p, g, m, v, l, step_size, beta1, beta2, 1.f - beta1, 1.f - beta2, eps);
    param[i] = p; exp_avg[i] = m; exp_avg_sq[i] = v;
}

static inline bool k4_aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <int MODE>
static int k4_adam_launch(float* param, const float* grad, float* m, float* v, const float* perlr, int64_t n, int step,
                          float beta1, float beta2, float lr, float eps, hipStream_t st) {
    if (n < 0 || step < 1) return K4_ERR_BAD_ARG;
    if (n == 0) return 0;                                  // empty tensors carry NULL data pointers
    if (!param || !grad || !m || !v || (MODE == K4_ADAM_PERLR && !perlr)) return K4_ERR_BAD_ARG;
    // host scalar exactly as adam_upd_kernel.cu:71 (all float arithmetic)
    const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
    const bool vec = k4_aligned16(param) && k4_aligned16(grad) && k4_aligned16(m) && k4_aligned16(v) &&
                     (MODE != K4_ADAM_PERLR || k4_aligned16(perlr));
    const int64_t n4 = vec ? n / 4 : 0;
    if (n4 > 0) {
        const int64_t blocks = (n4 + K4_OPT_THREADS - 1) / K4_OPT_THREADS;
        if (blocks > 0x7fffffffLL) return K4_ERR_BAD_ARG;
        hipLaunchKernelGGL(k4_adam_vec_kernel<MODE>, dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, st, (float4*)param,
                           (const float4*)grad, (float4*)m, (float4*)v, (const float4*)perlr, n4, step_size, beta1, beta2, eps);
    }
    const int64_t first = n4 * 4;
    if (first < n) {
        const int64_t blocks = (n - first + K4_OPT_THREADS - 1) / K4_OPT_THREADS;
        if (blocks > 0x7fffffffLL) return K4_ERR_BAD_ARG;
        hipLaunchKernelGGL(k4_adam_scalar_kernel<MODE>, dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, st, param, grad, m, v,
                           perlr, first, n, step_size, beta1, beta2, eps);
    }
    return k4_check_launch();
}

extern "C" int k4_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                           float beta1, float beta2, float lr, float eps, void* stream) {
    return k4_adam_launch<K4_ADAM_PLAIN>(param, grad, exp_avg, exp_avg_sq, nullptr, n, step, beta1, beta2, lr, eps,
                                         (hipStream_t)stream);
}
extern "C" int k4_masked_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  int32_t step, float beta1, float beta2, float lr, float eps, void* stream) {
    return k4_adam_launch<K4_ADAM_MASKED>(param, grad, exp_avg, exp_avg_sq, nullptr, n, step, beta1, beta2, lr, eps,
                                          (hipStream_t)stream);
}
// MaskedAdam over the touched voxels of a multi-channel grid, straight from the channel-last scratch image of the lookup's backward
// (k4_grid_sample_3d_backward_cl_scatter, k4_staged.hip: scratch[voxel][C] sums + one flag byte per voxel).  After tv_before the reference's
// iteration (run_sr.py:1005-1014 with configs/llff/fern_lg_joint_l1.py: 290,000 of its 300,000) has no other contribution to the grid's
// gradient, and masked_adam_upd (lib/cuda/adam_upd_kernel.cu:27-42) skips every element whose gradient is zero: the dense gradient -- 1.36 GB
// cleared, swept into and read again per iteration for the LLFF k0 -- holds nothing this kernel does not find behind the flags.  Same
// arithmetic per element (k4_adam_one); scratch and flags are all-zero again behind it.
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_adam_sparse_cl_kernel(float* __restrict__ scratch, uint8_t* __restrict__ flags, int C, int64_t nvox,
                                                                            float* __restrict__ param, float* __restrict__ em, float* __restrict__ ev,
                                                                            float step_size, float beta1, float beta2, float eps) {
    const int64_t vx = (int64_t)blockIdx.x * K4_OPT_THREADS + threadIdx.x;
    if (vx >= nvox || flags[vx] == 0) return;
    flags[vx] = 0;
    float* const s = scratch + vx * C;
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    for (int ch = 0; ch < C; ++ch) {
        const float g = s[ch];
        if (g == 0.f) continue;
        s[ch] = 0.f;
        const int64_t i = (int64_t)ch * nvox + vx;
        float p = param[i], m = em[i], v = ev[i];
        k4_adam_one(p, g, m, v, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        param[i] = p; em[i] = m; ev[i] = v;
    }
}
extern "C" int k4_masked_adam_upd_sparse_cl(float* param, float* exp_avg, float* exp_avg_sq, void* workspace, int32_t C, int32_t X, int32_t Y, int32_t Z,
                                            int32_t step, float beta1, float beta2, float lr, float eps, void* stream) {
    if (!param || !exp_avg || !exp_avg_sq || !workspace || (((uintptr_t)workspace) & 15) || C <= 1 || C > 32 || X <= 0 || Y <= 0 || Z <= 0 || step < 1) return K4_ERR_BAD_ARG;
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t blocks = (nvox + K4_OPT_THREADS - 1) / K4_OPT_THREADS;
    if (blocks > 0x7fffffffLL) return K4_ERR_BAD_ARG;
    const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));      // adam_upd_kernel.cu:71
    hipLaunchKernelGGL(k4_adam_sparse_cl_kernel, dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, (hipStream_t)stream, (float*)workspace,
                       (uint8_t*)workspace + nvox * C * 4, C, nvox, param, exp_avg, exp_avg_sq, step_size, beta1, beta2, eps);
    return k4_check_launch();
}
// The masked step of a multi-channel grid [C][nvox] in TWO parts (exact: Adam is elementwise, every element is stepped once with its complete gradient).
//   k4_masked_adam_upd_unflagged          : every voxel whose flag byte is 0, gradient = `grad` (the dense TV term written ahead of the backward pass: for a voxel
//                                           the lookups' scatter cannot touch that IS the iteration's gradient) -- runs beside the rest of the iteration;
//   k4_masked_adam_upd_sparse_cl_seeded   : the flagged voxels after the backward pass, gradient = seed + the scatter's sum in the channel-last scratch image
//                                           (the sweep's `grad += sum`, same operand order); scratch, its flags and `flags` are all-zero again behind it.
// In the first 10,000 iterations of fern_lg_joint_l1 the dense k0 step (1.8 ms at 0.72-0.77 of the HBM roof) sat between the end of one iteration's backward pass
// and the next iteration's k0 lookup; of its 37.7 M voxels the scatter touches a few 10^5.
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_adam_unflagged_kernel(float4* __restrict__ param, const float4* __restrict__ grad, float4* __restrict__ em,
                                                                            float4* __restrict__ ev, const uint32_t* __restrict__ flags4, int64_t nvox4,
                                                                            float step_size, float beta1, float beta2, float eps) {
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    // grid-stride: the launch may be capped at a few workgroups per channel (max_workgroups) -- this pass runs BESIDE latency-bound kernels of other streams
    for (int64_t v4 = (int64_t)blockIdx.x * K4_OPT_THREADS + threadIdx.x; v4 < nvox4; v4 += (int64_t)gridDim.x * K4_OPT_THREADS) {
        const uint32_t f = flags4[v4];
        if ((f & 0xffu) && (f & 0xff00u) && (f & 0xff0000u) && (f & 0xff000000u)) continue;
        const int64_t i = (int64_t)blockIdx.y * nvox4 + v4;
        const float4 g = grad[i];
        if (g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) continue;
        float4 p = param[i], m = em[i], v = ev[i];
        if (!(f & 0xffu) && g.x != 0.f) k4_adam_one(p.x, g.x, m.x, v.x, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        if (!(f & 0xff00u) && g.y != 0.f) k4_adam_one(p.y, g.y, m.y, v.y, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        if (!(f & 0xff0000u) && g.z != 0.f) k4_adam_one(p.z, g.z, m.z, v.z, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        if (!(f & 0xff000000u) && g.w != 0.f) k4_adam_one(p.w, g.w, m.w, v.w, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        param[i] = p; em[i] = m; ev[i] = v;
    }
}
extern "C" int k4_masked_adam_upd_unflagged(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t C, int64_t nvox, const uint8_t* flags,
                                            int32_t step, float beta1, float beta2, float lr, float eps, int32_t max_workgroups, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !flags || C < 1 || C > 65535 || nvox <= 0 || step < 1 || max_workgroups < 0) return K4_ERR_BAD_ARG;
    if ((nvox & 3) || !k4_aligned16(param) || !k4_aligned16(grad) || !k4_aligned16(exp_avg) || !k4_aligned16(exp_avg_sq) || ((uintptr_t)flags & 3u)) return K4_ERR_UNSUPPORTED;
    const int64_t nvox4 = nvox / 4;
    int64_t blocks = (nvox4 + K4_OPT_THREADS - 1) / K4_OPT_THREADS;
    if (blocks > 0x7fffffffLL) return K4_ERR_BAD_ARG;
    if (max_workgroups > 0) {                              // `max_workgroups` in total: per channel (blockIdx.y) its share
        const int64_t per = (max_workgroups + C - 1) / C;
        if (per < blocks) blocks = per;
    }
    const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));      // adam_upd_kernel.cu:71
    hipLaunchKernelGGL(k4_adam_unflagged_kernel, dim3((unsigned)blocks, (unsigned)C), dim3(K4_OPT_THREADS), 0, (hipStream_t)stream, (float4*)param, (const float4*)grad,
                       (float4*)exp_avg, (float4*)exp_avg_sq, (const uint32_t*)flags, nvox4, step_size, beta1, beta2, eps);
    return k4_check_launch();
}
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_adam_sparse_cl_seeded_kernel(float* __restrict__ scratch, uint8_t* __restrict__ sflags, uint8_t* __restrict__ flags,
                                                                                   const float* __restrict__ seed, int C, int64_t nvox, float* __restrict__ param,
                                                                                   float* __restrict__ em, float* __restrict__ ev, float step_size, float beta1,
                                                                                   float beta2, float eps) {
    const int64_t vx = (int64_t)blockIdx.x * K4_OPT_THREADS + threadIdx.x;
    if (vx >= nvox || flags[vx] == 0) return;
    flags[vx] = 0;
    sflags[vx] = 0;
    float* const s = scratch + vx * C;
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    for (int ch = 0; ch < C; ++ch) {
        const int64_t i = (int64_t)ch * nvox + vx;
        const float q = s[ch];
        s[ch] = 0.f;
        const float g = seed[i] + q;                       // (k_gsb_cl_sweep: grad[i] += q)
        if (g == 0.f) continue;
        float p = param[i], m = em[i], v = ev[i];
        k4_adam_one(p, g, m, v, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        param[i] = p; em[i] = m; ev[i] = v;
    }
}
extern "C" int k4_masked_adam_upd_sparse_cl_seeded(float* param, float* exp_avg, float* exp_avg_sq, void* workspace, const float* seed, uint8_t* flags,
                                                   int32_t C, int32_t X, int32_t Y, int32_t Z, int32_t step, float beta1, float beta2, float lr, float eps, void* stream) {
    if (!param || !exp_avg || !exp_avg_sq || !workspace || !seed || !flags || (((uintptr_t)workspace) & 15) || C <= 1 || C > 32 || X <= 0 || Y <= 0 || Z <= 0 || step < 1)
        return K4_ERR_BAD_ARG;
    const int64_t nvox = (int64_t)X * Y * Z;
    const int64_t blocks = (nvox + K4_OPT_THREADS - 1) / K4_OPT_THREADS;
    if (blocks > 0x7fffffffLL) return K4_ERR_BAD_ARG;
    const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));      // adam_upd_kernel.cu:71
    hipLaunchKernelGGL(k4_adam_sparse_cl_seeded_kernel, dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, (hipStream_t)stream, (float*)workspace,
                       (uint8_t*)workspace + nvox * C * 4, flags, seed, C, nvox, param, exp_avg, exp_avg_sq, step_size, beta1, beta2, eps);
    return k4_check_launch();
}
extern "C" int k4_adam_upd_with_perlr(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                      const float* perlr, int64_t n, int32_t step, float beta1, float beta2, float lr,
                                      float eps, void* stream) {
    return k4_adam_launch<K4_ADAM_PERLR>(param, grad, exp_avg, exp_avg_sq, perlr, n, step, beta1, beta2, lr, eps,
                                         (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ many small tensors, one launch
// The decoder has 458 parameter tensors (3.96 M floats, 15.8 MB): one k4_adam_upd per tensor is 458 launches of a few microseconds each
// per optimizer step -- launch-bound on the host.  k4_adam_upd_multi walks up to K4_ADAM_MULTI_MAX tensors per launch: the job table
// travels as a kernel argument, a workgroup owns K4_ADAM_MULTI_CHUNK consecutive elements of one tensor (found by a scan of the block
// prefix table), same arithmetic per element as the single-tensor kernels.
struct K4AdamMultiArgs {
    float* param[K4_ADAM_MULTI_MAX];
    const float* grad[K4_ADAM_MULTI_MAX];
    float* exp_avg[K4_ADAM_MULTI_MAX];
    float* exp_avg_sq[K4_ADAM_MULTI_MAX];
    int64_t n[K4_ADAM_MULTI_MAX];
    int32_t blk_end[K4_ADAM_MULTI_MAX];       // exclusive prefix end of each job's workgroups
    int32_t n_jobs;
};
#define K4_ADAM_MULTI_CHUNK 1024

template <int MODE>
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_adam_multi_kernel(const K4AdamMultiArgs A, float step_size, float beta1, float beta2, float eps) {
    int j = 0;
    const int b = (int)blockIdx.x;
    while (j + 1 < A.n_jobs && b >= A.blk_end[j]) ++j;
    const int64_t base = (int64_t)(b - (j ? A.blk_end[j - 1] : 0)) * K4_ADAM_MULTI_CHUNK;
    float* const param = A.param[j];
    const float* const grad = A.grad[j];
    float* const em = A.exp_avg[j];
    float* const ev = A.exp_avg_sq[j];
    const int64_t n = A.n[j];
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
#pragma unroll
    for (int u = 0; u < K4_ADAM_MULTI_CHUNK / K4_OPT_THREADS; ++u) {
        const int64_t i = base + u * K4_OPT_THREADS + threadIdx.x;
        if (i >= n) break;
        const float g = grad[i];
        if (MODE == K4_ADAM_MASKED && g == 0.f) continue;
        float p = param[i], m = em[i], v = ev[i];
        k4_adam_one(p, g, m, v, 1.f, step_size, beta1, beta2, omb1, omb2, eps);
        param[i] = p; em[i] = m; ev[i] = v;
    }
}

extern "C" int k4_adam_upd_multi(const k4_adam_job* jobs, int32_t n_jobs, int32_t masked, int32_t step, float beta1, float beta2, float lr,
                                 float eps, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs) || step < 1) return K4_ERR_BAD_ARG;
    const float step_size = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));      // adam_upd_kernel.cu:71
    for (int first = 0; first < n_jobs;) {
        K4AdamMultiArgs A{};
        int k = 0, blocks = 0;
        for (; first < n_jobs && k < K4_ADAM_MULTI_MAX; ++first) {
            const k4_adam_job& J = jobs[first];
            if (J.n < 0) return K4_ERR_BAD_ARG;
            if (J.n == 0) continue;
            if (!J.param || !J.grad || !J.exp_avg || !J.exp_avg_sq) return K4_ERR_BAD_ARG;
            const int64_t nb = (J.n + K4_ADAM_MULTI_CHUNK - 1) / K4_ADAM_MULTI_CHUNK;
            if (nb + blocks > 0x3fffffffLL) return K4_ERR_BAD_ARG;
            A.param[k] = J.param; A.grad[k] = J.grad; A.exp_avg[k] = J.exp_avg; A.exp_avg_sq[k] = J.exp_avg_sq; A.n[k] = J.n;
            blocks += (int)nb;
            A.blk_end[k] = blocks;
            ++k;
        }
        if (k == 0) continue;
        A.n_jobs = k;
        if (masked) hipLaunchKernelGGL(k4_adam_multi_kernel<K4_ADAM_MASKED>, dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, (hipStream_t)stream, A, step_size, beta1, beta2, eps);
        else hipLaunchKernelGGL(k4_adam_multi_kernel<K4_ADAM_PLAIN>, dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, (hipStream_t)stream, A, step_size, beta1, beta2, eps);
        const int rc = k4_check_launch();
        if (rc) return rc;
    }
    return K4_OK;
}

// ------------------------------------------------------------------------------------------------ total variation
__device__ __forceinline__ float k4_clamp1(float x) { return fminf(fmaxf(x, -1.f), 1.f); }

// One thread = 4 consecutive k (needs sz_k % 4 == 0 and 16-byte aligned bases).  The +-1 neighbours along k are the
// adjacent lanes' values except at the two ends of the 4-group, fetched as scalars (same cache lines).
// The gradient is streamed (read once, written once).  K4_TV_NT=1: non-temporal accesses for it (so that it does not evict the parameter planes the stencil
// re-reads from L2) -- measured neutral on the dense sweeps (1.229 / 1.237 ms), -6 % on the sparse one; off.
#ifndef K4_TV_NT
#define K4_TV_NT 0
#endif
typedef float k4_tv_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 k4_tv_ld4(const float* p) {
#if K4_TV_NT
    const k4_tv_f4 v = __builtin_nontemporal_load(reinterpret_cast<const k4_tv_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void k4_tv_st4(float* p, const float4& g) {
#if K4_TV_NT
    const k4_tv_f4 v = {g.x, g.y, g.z, g.w};
    __builtin_nontemporal_store(v, reinterpret_cast<k4_tv_f4*>(p));
#else
    *reinterpret_cast<float4*>(p) = g;
#endif
}
// MODE 0: only where grad != 0 (sparse), 1: everywhere (dense), 2: dense and grad is OVERWRITTEN with the TV term (not read): the term
// computed before the backward pass into the buffer the lookups' backward then accumulates into (lib/grid.py total_variation_seed_grad) --
// 8 bytes per voxel instead of 4 (zero-fill) + 12, and off the tail of the training iteration.  0 + term == term: same values.
// IDX = unsigned for tensors below 2^31 elements: the voxel coordinates cost three 32-bit divisions per thread instead of five 64-bit ones -- which, not
// memory, bounded the first form (the LLFF k0 sweep: 1.23 ms at 12 B / voxel = 0.55 of HBM peak with ~600 integer instructions per thread in front of 8 memory
// instructions).
template <int MODE, typename IDX>
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_tv_vec_kernel(const float* __restrict__ param, float* __restrict__ grad,
                                                                   float wx, float wy, float wz, int64_t sz_i_, int64_t sz_j_,
                                                                   int64_t sz_k_, int64_t n4) {
    const int b = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const IDX t = (IDX)b * K4_OPT_THREADS + threadIdx.x;
    if ((int64_t)t >= n4) return;
    const IDX index = t * 4;
    constexpr bool DENSE = MODE != 0;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE != 2) g = k4_tv_ld4(grad + index);
    if (!DENSE && g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;
    const IDX sz_i = (IDX)sz_i_, sz_j = (IDX)sz_j_, sz_k = (IDX)sz_k_;
    const IDX row = index / sz_k, k = index - row * sz_k;      // row = (c * sz_i + i) * sz_j + j
    const IDX pl = row / sz_j, j = row - pl * sz_j;
    const IDX i = pl % sz_i;
    const IDX sj = sz_k, si = sz_k * sz_j;
    const float4 c = *(const float4*)(param + index);
    const float cv[4] = {c.x, c.y, c.z, c.w};
    float add[4] = {0.f, 0.f, 0.f, 0.f};
    // k axis (.cu:28-29): term order of the reference: k-1, k+1, j-1, j+1, i-1, i+1
    const float km = k == 0 ? 0.f : param[index - 1];
    const float kp = k + 4 >= sz_k ? 0.f : param[index + 4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool has_m = (k + e) != 0, has_p = (k + e) != sz_k - 1;
        const float vm = e == 0 ? km : cv[e - 1];
        const float vp = e == 3 ? kp : cv[e + 1];
        if (has_m) add[e] = fmaf(wx, k4_clamp1(cv[e] - vm), add[e]);
        if (has_p) add[e] = fmaf(wx, k4_clamp1(cv[e] - vp), add[e]);
    }
    if (j != 0) {
        const float4 q = *(const float4*)(param + index - sj);
        add[0] = fmaf(wy, k4_clamp1(cv[0] - q.x), add[0]); add[1] = fmaf(wy, k4_clamp1(cv[1] - q.y), add[1]);
        add[2] = fmaf(wy, k4_clamp1(cv[2] - q.z), add[2]); add[3] = fmaf(wy, k4_clamp1(cv[3] - q.w), add[3]);
    }
    if (j != sz_j - 1) {
        const float4 q = *(const float4*)(param + index + sj);
        add[0] = fmaf(wy, k4_clamp1(cv[0] - q.x), add[0]); add[1] = fmaf(wy, k4_clamp1(cv[1] - q.y), add[1]);
        add[2] = fmaf(wy, k4_clamp1(cv[2] - q.z), add[2]); add[3] = fmaf(wy, k4_clamp1(cv[3] - q.w), add[3]);
    }
    if (i != 0) {
        const float4 q = *(const float4*)(param + index - si);
        add[0] = fmaf(wz, k4_clamp1(cv[0] - q.x), add[0]); add[1] = fmaf(wz, k4_clamp1(cv[1] - q.y), add[1]);
        add[2] = fmaf(wz, k4_clamp1(cv[2] - q.z), add[2]); add[3] = fmaf(wz, k4_clamp1(cv[3] - q.w), add[3]);
    }
    if (i != sz_i - 1) {
        const float4 q = *(const float4*)(param + index + si);
        add[0] = fmaf(wz, k4_clamp1(cv[0] - q.x), add[0]); add[1] = fmaf(wz, k4_clamp1(cv[1] - q.y), add[1]);
        add[2] = fmaf(wz, k4_clamp1(cv[2] - q.z), add[2]); add[3] = fmaf(wz, k4_clamp1(cv[3] - q.w), add[3]);
    }
    if (DENSE || g.x != 0.f) g.x += add[0];
    if (DENSE || g.y != 0.f) g.y += add[1];
    if (DENSE || g.z != 0.f) g.z += add[2];
    if (DENSE || g.w != 0.f) g.w += add[3];
    k4_tv_st4(grad + index, g);
}

template <int MODE, typename IDX>
__global__ __launch_bounds__(K4_OPT_THREADS) void k4_tv_scalar_kernel(const float* __restrict__ param, float* __restrict__ grad,
                                                                      float wx, float wy, float wz, int64_t sz_i_, int64_t sz_j_,
                                                                      int64_t sz_k_, int64_t n) {
    const int b = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const IDX index = (IDX)b * K4_OPT_THREADS + threadIdx.x;
    if ((int64_t)index >= n) return;
    constexpr bool DENSE = MODE != 0;
    const float g = MODE == 2 ? 0.f : grad[index];
    if (!DENSE && g == 0.f) return;
    const IDX sz_i = (IDX)sz_i_, sz_j = (IDX)sz_j_, sz_k = (IDX)sz_k_;
    const IDX row = index / sz_k, k = index - row * sz_k;
    const IDX pl = row / sz_j, j = row - pl * sz_j;
    const IDX i = pl % sz_i;
    const IDX sj = sz_k, si = sz_k * sz_j;
    const float c = param[index];
    float add = 0.f;
    if (k != 0) add = fmaf(wx, k4_clamp1(c - param[index - 1]), add);
    if (k != sz_k - 1) add = fmaf(wx, k4_clamp1(c - param[index + 1]), add);
    if (j != 0) add = fmaf(wy, k4_clamp1(c - param[index - sj]), add);
    if (j != sz_j - 1) add = fmaf(wy, k4_clamp1(c - param[index + sj]), add);
    if (i != 0) add = fmaf(wz, k4_clamp1(c - param[index - si]), add);
    if (i != sz_i - 1) add = fmaf(wz, k4_clamp1(c - param[index + si]), add);
    grad[index] = g + add;
}

extern "C" int k4_total_variation_add_grad(const float* param, float* grad, float wx, float wy, float wz, int64_t sz_i,
                                           int64_t sz_j, int64_t sz_k, int64_t n, int32_t dense_mode, void* stream) {
    if (n < 0 || sz_i <= 0 || sz_j <= 0 || sz_k <= 0 || n % (sz_i * sz_j * sz_k) != 0 || dense_mode < 0 || dense_mode > 2) return K4_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!param || !grad) return K4_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    wx /= 6.f; wy /= 6.f; wz /= 6.f;                       // total_variation_kernel.cu:46-48
    const bool vec = (sz_k % 4 == 0) && k4_aligned16(param) && k4_aligned16(grad);
    const int64_t units = vec ? n / 4 : n;
    const int64_t blocks = (units + K4_OPT_THREADS - 1) / K4_OPT_THREADS;
    if (blocks > 0x7fffffffLL) return K4_ERR_BAD_ARG;
    const bool small = n + 4 * K4_OPT_THREADS < 0x7fffffffLL;            // every index the kernel forms (incl. the last block's overhang) fits 31 bits
#define K4_TV_LAUNCH_I(KERNEL, IDX) do { \
        if (dense_mode == 2) hipLaunchKernelGGL((KERNEL<2, IDX>), dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, st, param, grad, wx, wy, wz, sz_i, sz_j, sz_k, units); \
        else if (dense_mode) hipLaunchKernelGGL((KERNEL<1, IDX>), dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, st, param, grad, wx, wy, wz, sz_i, sz_j, sz_k, units); \
        else hipLaunchKernelGGL((KERNEL<0, IDX>), dim3((unsigned)blocks), dim3(K4_OPT_THREADS), 0, st, param, grad, wx, wy, wz, sz_i, sz_j, sz_k, units); } while (0)
#define K4_TV_LAUNCH(KERNEL) do { if (small) K4_TV_LAUNCH_I(KERNEL, unsigned); else K4_TV_LAUNCH_I(KERNEL, int64_t); } while (0)
    if (vec) K4_TV_LAUNCH(k4_tv_vec_kernel);
    else K4_TV_LAUNCH(k4_tv_scalar_kernel);
#undef K4_TV_LAUNCH
#undef K4_TV_LAUNCH_I
    return k4_check_launch();
}
