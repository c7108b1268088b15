// Training-step kernels of the marcher that the reference leaves to PyTorch library ops (SURVEY.md 8f rank 1 "MLP bwd", 3.4):
//   * the colour MLP `rgbnet` forward WITH saved activations and its backward (lib/dmpigo.py:112-120,375-379,
//     lib/dvgo.py:116-124,407-412: nn.Sequential of nn.Linear / ReLU + torch.sigmoid, differentiated by autograd over
//     cuBLAS GEMMs): two launches instead of ~20, exact fp32 FMA chains, deterministic (no atomics);
//   * the distortion loss of the joint training step (run_sr.py:976-988 calls torch_efficient_distloss.flatten_eff_distloss,
//     a third-party CUDA extension that is not vendored: the published prefix-sum form is restated here), forward value and
//     gradient in one launch;
//   * the touched-voxel list of a grid gradient for the data-parallel exchange (k4_touched_voxels; SURVEY.md 8e "Training (config 5)":
//     dense k0 grad 1.36 GB, a 64x64 patch touches < 1 % of it);
//   * the SFTLayer of the decoder's training graph, forward and backward fused (k4_sft_train_*), and LeakyReLU backward on channel slices.
// gfx950 only (wave64).  A training batch is ~10^5 points x (15..39 -> 64..128 -> 64..128 -> 3): ~1 GFLOP, far below the
// matrix cores' interest; what matters is launch count and that nothing but x / h / grad streams through HBM once.
#include "k4_common.h"
#include <atomic>
#include <mutex>

#define TR_LS 68            // LDS row stride in floats: rows 16-byte aligned (ds_read_b128 along the sample axis), lane = sample reads conflict-free
#define TR_MAX_DIM0 64

// --------------------------------------------------------------------------------------------------------------------
// rgbnet forward: tile = 64 samples, lane = sample, wave w owns a quarter of a layer's neurons (4 per pass: one LDS read
// of the input feeds 4 FMAs whose weights are wave-uniform -> scalar loads, SGPR operands)
// --------------------------------------------------------------------------------------------------------------------
template <int W, int NT>
__device__ __forceinline__ void tr_dense_relu(const float* in, int K, k4_cptr w, k4_cptr b, float* out, int wv, int lane) {
    constexpr int PER = W / (NT / 64);
    static_assert(PER % 4 == 0, "a wave owns whole groups of 4 neurons");
    for (int j0 = wv * PER; j0 < (wv + 1) * PER; j0 += 4) {
        float a0 = b[j0], a1 = b[j0 + 1], a2 = b[j0 + 2], a3 = b[j0 + 3];
        for (int k = 0; k < K; ++k) {
            const float v = in[k * TR_LS + lane];
            a0 = fmaf(v, w[(j0 + 0) * K + k], a0);
            a1 = fmaf(v, w[(j0 + 1) * K + k], a1);
            a2 = fmaf(v, w[(j0 + 2) * K + k], a2);
            a3 = fmaf(v, w[(j0 + 3) * K + k], a3);
        }
        out[(j0 + 0) * TR_LS + lane] = fmaxf(a0, 0.f);
        out[(j0 + 1) * TR_LS + lane] = fmaxf(a1, 0.f);
        out[(j0 + 2) * TR_LS + lane] = fmaxf(a2, 0.f);
        out[(j0 + 3) * TR_LS + lane] = fmaxf(a3, 0.f);
    }
}

// [n][K] row-major global tile (64 samples from `base`) -> LDS [K][TR_LS] (sample fastest); samples >= nv read as 0
template <int NT = 256>
__device__ __forceinline__ void tr_load_tile(const float* __restrict__ g, int64_t base, int K, int nv, float* lds, int t) {
    const float* const src = g + base * K;
    for (int i = t; i < 64 * K; i += NT) {
        const int s = i / K, k = i - s * K;
        lds[k * TR_LS + s] = s < nv ? src[i] : 0.f;
    }
}
template <int NT = 256>
__device__ __forceinline__ void tr_store_tile(const float* lds, float* __restrict__ g, int64_t base, int K, int nv, int t) {
    float* const dst = g + base * K;
    for (int i = t; i < 64 * K; i += NT) {
        const int s = i / K, k = i - s * K;
        if (s < nv) dst[i] = lds[k * TR_LS + s];
    }
}

// NT threads: 256, or 1024 for width >= 64 -- a training batch is a few hundred 64-sample tiles (one or two per workgroup), so a launch lasts as long as ONE tile's chain:
// with 4 waves (one per SIMD, nothing to hide the LDS / scalar-load latencies behind) the backward launch of the joint iteration's 37 k samples took 114 us and the forward
// 25 us; 16 waves split the same per-element FMA chains four times finer (same values, same order: cf. the SFT kernels below)
template <int W, int NT>
__global__ __launch_bounds__(NT) void k_rgbnet_fwd(const float* __restrict__ x, int64_t n, int dim0, int n_hidden,
                                                    const float* __restrict__ w1, const float* __restrict__ b1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                    const float* __restrict__ w3, const float* __restrict__ b3,
                                                    const float* __restrict__ add, float* __restrict__ h1g, float* __restrict__ h2g,
                                                    float* __restrict__ rgb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const xs = smem;                              // [dim0][LS]
    float* const h1s = xs + TR_MAX_DIM0 * TR_LS;         // [W][LS]
    float* const h2s = h1s + W * TR_LS;                  // [W][LS]
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nv = (n - base) < 64 ? (int)(n - base) : 64;
    tr_load_tile<NT>(x, base, dim0, nv, xs, t);
    __syncthreads();
    tr_dense_relu<W, NT>(xs, dim0, k4_const(w1), k4_const(b1), h1s, wv, lane);
    __syncthreads();
    if (h1g) tr_store_tile<NT>(h1s, h1g, base, W, nv, t);
    if (n_hidden) {
        tr_dense_relu<W, NT>(h1s, W, k4_const(w2), k4_const(b2), h2s, wv, lane);
        __syncthreads();
        if (h2g) tr_store_tile<NT>(h2s, h2g, base, W, nv, t);
    }
    const float* const hl = n_hidden ? h2s : h1s;
    if (wv < 3 && lane < nv) {
        const k4_cptr w3c = k4_const(w3) + wv * W;
        float acc = k4_const(b3)[wv];
        for (int j = 0; j < W; ++j) acc = fmaf(hl[j * TR_LS + lane], w3c[j], acc);
        if (add) acc += add[(base + lane) * 3 + wv];              // rgb_logit + k0_diffuse (lib/dvgo.py:412)
        rgb[(base + lane) * 3 + wv] = 1.f / (1.f + expf(-acc));   // torch.sigmoid
    }
}

// --------------------------------------------------------------------------------------------------------------------
// rgbnet backward.  Persistent workgroups walk the 64-sample tiles; per tile
//   g3  = grad_rgb * (1 - y) * y                                   (sigmoid)
//   gh_L = (h_L > 0) * g3 W3 ;  gh_1 = (h_1 > 0) * gh_2 W2 ;  gx = gh_1 W1          (lane = sample, scalar weights)
//   dW3 += g3^T [h_L | 1] ;  dW2 += gh_2^T [h_1 | 1] ;  dW1 += gh_1^T [x | 1]        (the appended ones row carries the bias gradient)
// The weight-gradient products are sums over the SAMPLE axis: every thread owns 4x4 blocks of (row of A, row of B) pairs and
// walks the 64 samples with ds_read_b128 (8 reads per 64 FMAs), accumulating in registers across all tiles of the workgroup;
// the per-workgroup partial sums go to `part` and k_rgbnet_reduce adds them in workgroup order: no atomics, the result
// depends on the grid size only.
// --------------------------------------------------------------------------------------------------------------------
struct TrBlock { int aoff, boff, ooff, ostride; };

template <int W>
struct TrBwdLayout {
    static constexpr int WB = W + 4;                                  // W rows + ones row + 3 zero rows
    static constexpr int GH2_ROWS = W > TR_MAX_DIM0 ? W : TR_MAX_DIM0; // also the staging area of gx
    static __host__ __device__ int d1b(int dim0) { return (dim0 + 4) & ~3; }     // dim0 rows + ones row, padded to 4
    static __host__ __device__ int lds_floats(int dim0) { return (d1b(dim0) + 2 * WB + 4 + W + GH2_ROWS) * TR_LS; }
    static __host__ __device__ int n_part(int dim0) { return 4 * WB + W * WB + W * d1b(dim0); }   // P3 [4][WB] | P2 [W][WB] | P1 [W][D1B]
};

template <int W, int MAXB, int NT>
__global__ __launch_bounds__(NT) void k_rgbnet_bwd(const float* __restrict__ x, int64_t n, int dim0, int n_hidden,
                                                    const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                                    const float* __restrict__ h1g, const float* __restrict__ h2g,
                                                    const float* __restrict__ rgb, const float* __restrict__ grgb,
                                                    float* __restrict__ gx, float* __restrict__ glogit, float* __restrict__ part) {
    typedef TrBwdLayout<W> L;
    constexpr int WB = L::WB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D1B = L::d1b(dim0);
    float* const xs = smem;                      // [D1B][LS]   x^T | ones | 0
    float* const h1s = xs + D1B * TR_LS;         // [WB][LS]    h1^T | ones | 0
    float* const h2s = h1s + WB * TR_LS;         // [WB][LS]
    float* const g3s = h2s + WB * TR_LS;         // [4][LS]     row 3 = 0
    float* const gh1 = g3s + 4 * TR_LS;          // [W][LS]
    float* const gh2 = gh1 + W * TR_LS;          // [GH2_ROWS][LS]; reused as the gx staging tile
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int NW = NT / 64;
    for (int i = t; i < L::lds_floats(dim0); i += NT) smem[i] = 0.f;       // the zero padding rows stay zero for the whole kernel

    float* const hls = n_hidden ? h2s : h1s;     // last hidden activation, its gradient:
    float* const ghl = n_hidden ? gh2 : gh1;

    // this thread's 4x4 blocks of the three weight-gradient products
    const int nb3 = WB / 4, nb2 = n_hidden ? (W / 4) * (WB / 4) : 0, nb1 = (W / 4) * (D1B / 4);
    TrBlock blk[MAXB];
    float acc[MAXB][16];
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
        int b = t + NT * q;
        blk[q].aoff = -1;
        if (b < nb3) {
            blk[q] = {(int)(g3s - smem), (int)(hls - smem) + 4 * b * TR_LS, 4 * b, WB};
        } else if (b < nb3 + nb2) {
            b -= nb3;
            const int ar = b / (WB / 4), bc = b - ar * (WB / 4);
            blk[q] = {(int)(gh2 - smem) + 4 * ar * TR_LS, (int)(h1s - smem) + 4 * bc * TR_LS, 4 * WB + 4 * ar * WB + 4 * bc, WB};
        } else if (b < nb3 + nb2 + nb1) {
            b -= nb3 + nb2;
            const int ar = b / (D1B / 4), bc = b - ar * (D1B / 4);
            blk[q] = {(int)(gh1 - smem) + 4 * ar * TR_LS, (int)(xs - smem) + 4 * bc * TR_LS, 4 * WB + W * WB + 4 * ar * D1B + 4 * bc, D1B};
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    }
    __syncthreads();

    const int64_t n_tiles = (n + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * 64;
        const int nv = (n - base) < 64 ? (int)(n - base) : 64;
        // ---- loads ----
        tr_load_tile<NT>(x, base, dim0, nv, xs, t);
        tr_load_tile<NT>(h1g, base, W, nv, h1s, t);
        if (n_hidden) tr_load_tile<NT>(h2g, base, W, nv, h2s, t);
        if (t < 64) {
            const float one = t < nv ? 1.f : 0.f;
            xs[dim0 * TR_LS + t] = one;
            h1s[W * TR_LS + t] = one;
            h2s[W * TR_LS + t] = one;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float g3 = 0.f;
                if (t < nv) {
                    const float y = rgb[(base + t) * 3 + c];
                    g3 = (grgb[(base + t) * 3 + c] * (1.f - y)) * y;            // sigmoid_backward: grad * (1 - y) * y
                    if (glogit) glogit[(base + t) * 3 + c] = g3;
                }
                g3s[c * TR_LS + t] = g3;
            }
        }
        __syncthreads();
        // ---- gh_L = relu'(h_L) * (g3 W3) ----
        {
            const k4_cptr w3c = k4_const(w3);
            const float g0 = g3s[lane], g1 = g3s[TR_LS + lane], g2 = g3s[2 * TR_LS + lane];
            constexpr int PER = W / NW;
            for (int j = wv * PER; j < (wv + 1) * PER; ++j) {
                float a = g0 * w3c[j];
                a = fmaf(g1, w3c[W + j], a);
                a = fmaf(g2, w3c[2 * W + j], a);
                ghl[j * TR_LS + lane] = hls[j * TR_LS + lane] > 0.f ? a : 0.f;
            }
        }
        __syncthreads();
        // ---- gh_1 = relu'(h_1) * (gh_2 W2) ----
        if (n_hidden) {
            const k4_cptr w2c = k4_const(w2);
            constexpr int PER = W / NW;
            for (int k0 = wv * PER; k0 < (wv + 1) * PER; k0 += 4) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int j = 0; j < W; ++j) {
                    const float v = gh2[j * TR_LS + lane];
                    a0 = fmaf(v, w2c[j * W + k0 + 0], a0);
                    a1 = fmaf(v, w2c[j * W + k0 + 1], a1);
                    a2 = fmaf(v, w2c[j * W + k0 + 2], a2);
                    a3 = fmaf(v, w2c[j * W + k0 + 3], a3);
                }
                gh1[(k0 + 0) * TR_LS + lane] = h1s[(k0 + 0) * TR_LS + lane] > 0.f ? a0 : 0.f;
                gh1[(k0 + 1) * TR_LS + lane] = h1s[(k0 + 1) * TR_LS + lane] > 0.f ? a1 : 0.f;
                gh1[(k0 + 2) * TR_LS + lane] = h1s[(k0 + 2) * TR_LS + lane] > 0.f ? a2 : 0.f;
                gh1[(k0 + 3) * TR_LS + lane] = h1s[(k0 + 3) * TR_LS + lane] > 0.f ? a3 : 0.f;
            }
            __syncthreads();
        }
        // ---- weight-gradient blocks: acc[a][b] += sum_s A[a][s] * B[b][s] ----
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            if (blk[q].aoff < 0) continue;
            const float4* const A = reinterpret_cast<const float4*>(smem + blk[q].aoff);
            const float4* const B = reinterpret_cast<const float4*>(smem + blk[q].boff);
            for (int s4 = 0; s4 < 16; ++s4) {
                float4 av[4], bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { av[r] = A[r * (TR_LS / 4) + s4]; bv[r] = B[r * (TR_LS / 4) + s4]; }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float v = acc[q][a * 4 + b];
                        v = fmaf(av[a].x, bv[b].x, v);
                        v = fmaf(av[a].y, bv[b].y, v);
                        v = fmaf(av[a].z, bv[b].z, v);
                        v = fmaf(av[a].w, bv[b].w, v);
                        acc[q][a * 4 + b] = v;
                    }
            }
        }
        __syncthreads();                                   // gh2 is free from here on: gx is staged in it
        // ---- gx = gh_1 W1 ----
        if (gx) {
            const k4_cptr w1c = k4_const(w1);
            const int nbx = (dim0 + 3) / 4;
            for (int bi = wv; bi < nbx; bi += NW) {
                const int i0 = bi * 4;
                const int i1 = min(i0 + 1, dim0 - 1), i2 = min(i0 + 2, dim0 - 1), i3 = min(i0 + 3, dim0 - 1);    // clamped: never out of W1
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int k = 0; k < W; ++k) {
                    const float v = gh1[k * TR_LS + lane];
                    a0 = fmaf(v, w1c[k * dim0 + i0], a0);
                    a1 = fmaf(v, w1c[k * dim0 + i1], a1);
                    a2 = fmaf(v, w1c[k * dim0 + i2], a2);
                    a3 = fmaf(v, w1c[k * dim0 + i3], a3);
                }
                gh2[i0 * TR_LS + lane] = a0;
                if (i0 + 1 < dim0) gh2[(i0 + 1) * TR_LS + lane] = a1;
                if (i0 + 2 < dim0) gh2[(i0 + 2) * TR_LS + lane] = a2;
                if (i0 + 3 < dim0) gh2[(i0 + 3) * TR_LS + lane] = a3;
            }
            __syncthreads();
            tr_store_tile<NT>(gh2, gx, base, dim0, nv, t);
        }
        __syncthreads();
    }

    float* const mine = part + (size_t)blockIdx.x * L::n_part(dim0);
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
        if (blk[q].aoff < 0) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) mine[blk[q].ooff + a * blk[q].ostride + b] = acc[q][a * 4 + b];
    }
}

// partial sums [n_wg][P3 | P2 | P1] -> gW3 [3][W], gb3 [3], gW2 [W][W], gb2 [W], gW1 [W][dim0], gb1 [W].  A workgroup = 16 output elements x
// 16 slices of the workgroup axis: every thread adds its slice in workgroup order (independent loads in flight instead of one
// 512-long dependent chain per element: 87 -> ~10 us), the 16 slice sums are added in slice order -- a fixed order, no atomics.
#define TR_RED_ELEMS 16
#define TR_RED_SLICES 16
__global__ __launch_bounds__(256) void k_rgbnet_reduce(const float* __restrict__ part, int n_wg, int n_part, int W, int WB, int dim0, int D1B, int n_hidden,
                                                       float* __restrict__ gw1, float* __restrict__ gb1, float* __restrict__ gw2, float* __restrict__ gb2,
                                                       float* __restrict__ gw3, float* __restrict__ gb3) {
    __shared__ float red[TR_RED_SLICES][TR_RED_ELEMS + 1];
    const int e = threadIdx.x & (TR_RED_ELEMS - 1), sl = threadIdx.x / TR_RED_ELEMS;
    const int i = blockIdx.x * TR_RED_ELEMS + e;
    const int n3 = 3 * (W + 1), n2 = n_hidden ? W * (W + 1) : 0, n1 = W * (dim0 + 1);
    const bool live = i < n3 + n2 + n1;
    int off = 0;
    float* dst = nullptr;
    if (live) {
        if (i < n3) {
            const int c = i / (W + 1), j = i - c * (W + 1);
            off = c * WB + j;
            dst = j < W ? gw3 + c * W + j : gb3 + c;
        } else if (i < n3 + n2) {
            const int r = i - n3, a = r / (W + 1), j = r - a * (W + 1);
            off = 4 * WB + a * WB + j;
            dst = j < W ? gw2 + a * W + j : gb2 + a;
        } else {
            const int r = i - n3 - n2, a = r / (dim0 + 1), j = r - a * (dim0 + 1);
            off = 4 * WB + W * WB + a * D1B + j;
            dst = j < dim0 ? gw1 + a * dim0 + j : gb1 + a;
        }
    }
    const int per = (n_wg + TR_RED_SLICES - 1) / TR_RED_SLICES;
    const int g0 = sl * per, g1 = min(g0 + per, n_wg);
    float s = 0.f;
    if (live)
        for (int g = g0; g < g1; ++g) s += part[(size_t)g * n_part + off];
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && live) {
        float t = red[0][e];
#pragma unroll
        for (int q = 1; q < TR_RED_SLICES; ++q) t += red[q][e];
        *dst = t;
    }
}

static int tr_bwd_grid(int64_t n) {
    const int64_t tiles = (n + 63) / 64;
    const int64_t cap = 2 * (int64_t)k4_num_cus();
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

template <int W>
static int tr_launch_fwd(const float* x, int64_t n, int dim0, int n_hidden, const float* w1, const float* b1, const float* w2, const float* b2,
                         const float* w3, const float* b3, const float* add, float* h1, float* h2, float* rgb, hipStream_t st) {
    const size_t lds = (size_t)(TR_MAX_DIM0 + 2 * W) * TR_LS * sizeof(float);
    constexpr int NT = W >= 64 ? 1024 : 256;
    K4_ENSURE_DYN_LDS((k_rgbnet_fwd<W, NT>), lds);
    hipLaunchKernelGGL((k_rgbnet_fwd<W, NT>), dim3((unsigned)((n + 63) / 64)), dim3(NT), lds, st, x, n, dim0, n_hidden, w1, b1, w2, b2, w3, b3, add, h1, h2, rgb);
    return k4_check_launch();
}

template <int W, int MAXB>
static int tr_launch_bwd(const float* x, int64_t n, int dim0, int n_hidden, const float* w1, const float* w2, const float* w3,
                         const float* h1, const float* h2, const float* rgb, const float* grgb, float* gx, float* glogit,
                         float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3, float* ws, hipStream_t st) {
    typedef TrBwdLayout<W> L;
    const size_t lds = (size_t)L::lds_floats(dim0) * sizeof(float);
    constexpr int NT = W == 64 ? 1024 : 256;                   // (width 128 at 1024 threads: two blocks per thread spill at the 128 registers a 16-wave workgroup leaves)
    K4_ENSURE_DYN_LDS((k_rgbnet_bwd<W, MAXB, NT>), lds);
    const int grid = tr_bwd_grid(n);
    hipLaunchKernelGGL((k_rgbnet_bwd<W, MAXB, NT>), dim3(grid), dim3(NT), lds, st, x, n, dim0, n_hidden, w1, w2, w3, h1, h2, rgb, grgb, gx, glogit, ws);
    int rc = k4_check_launch();
    if (rc) return rc;
    const int total = 3 * (W + 1) + (n_hidden ? W * (W + 1) : 0) + W * (dim0 + 1);
    hipLaunchKernelGGL(k_rgbnet_reduce, dim3((total + TR_RED_ELEMS - 1) / TR_RED_ELEMS), dim3(256), 0, st, ws, grid, L::n_part(dim0), W, L::WB, dim0, L::d1b(dim0), n_hidden,
                       gw1, gb1, gw2, gb2, gw3, gb3);
    return k4_check_launch();
}

static bool tr_shape_ok(int dim0, int width, int n_hidden) {
    return (width == 32 || width == 64 || width == 128) && (n_hidden == 0 || n_hidden == 1) && dim0 >= 1 && dim0 <= TR_MAX_DIM0;
}

extern "C" int64_t k4_rgbnet_bwd_workspace_bytes(int64_t n_pts, int32_t dim0, int32_t width, int32_t n_hidden) {
    if (!tr_shape_ok(dim0, width, n_hidden) || n_pts < 0) return -1;
    const int np = width == 32 ? TrBwdLayout<32>::n_part(dim0) : width == 64 ? TrBwdLayout<64>::n_part(dim0) : TrBwdLayout<128>::n_part(dim0);
    return (int64_t)tr_bwd_grid(n_pts) * np * (int64_t)sizeof(float);
}

extern "C" int k4_rgbnet_fwd(const float* x, int64_t n_pts, int32_t dim0, int32_t width, int32_t n_hidden,
                             const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                             const float* add, float* h1, float* h2, float* rgb, void* stream) {
    if (!tr_shape_ok(dim0, width, n_hidden)) return K4_ERR_UNSUPPORTED;
    if (n_pts < 0 || !w1 || !b1 || !w3 || !b3 || (n_hidden && (!w2 || !b2))) return K4_ERR_BAD_ARG;
    if (n_pts == 0) return K4_OK;
    if (!x || !rgb) return K4_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (width) {
        case 32: return tr_launch_fwd<32>(x, n_pts, dim0, n_hidden, w1, b1, w2, b2, w3, b3, add, h1, h2, rgb, st);
        case 64: return tr_launch_fwd<64>(x, n_pts, dim0, n_hidden, w1, b1, w2, b2, w3, b3, add, h1, h2, rgb, st);
        default: return tr_launch_fwd<128>(x, n_pts, dim0, n_hidden, w1, b1, w2, b2, w3, b3, add, h1, h2, rgb, st);
    }
}

extern "C" int k4_rgbnet_bwd(const float* x, int64_t n_pts, int32_t dim0, int32_t width, int32_t n_hidden,
                             const float* w1, const float* w2, const float* w3, const float* h1, const float* h2,
                             const float* rgb, const float* grad_rgb, float* grad_x, float* grad_logit,
                             float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3,
                             float* workspace, int64_t workspace_bytes, void* stream) {
    if (!tr_shape_ok(dim0, width, n_hidden)) return K4_ERR_UNSUPPORTED;
    if (n_pts < 0 || !w1 || !w3 || !gw1 || !gb1 || !gw3 || !gb3 || (n_hidden && (!w2 || !gw2 || !gb2))) return K4_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < k4_rgbnet_bwd_workspace_bytes(n_pts, dim0, width, n_hidden)) return K4_ERR_BAD_ARG;
    if (n_pts > 0 && (!x || !h1 || !rgb || !grad_rgb || (n_hidden && !h2))) return K4_ERR_BAD_ARG;      // n_pts == 0: the gradients are zeros
    hipStream_t st = (hipStream_t)stream;
    switch (width) {          // MAXB = ceil(blocks / threads) at dim0 = TR_MAX_DIM0: 32 -> 226 and 128 -> 1633 blocks (256 threads), 64 -> 561 (1024 threads)
        case 32: return tr_launch_bwd<32, 1>(x, n_pts, dim0, n_hidden, w1, w2, w3, h1, h2, rgb, grad_rgb, grad_x, grad_logit, gw1, gb1, gw2, gb2, gw3, gb3, workspace, st);
        case 64: return tr_launch_bwd<64, 1>(x, n_pts, dim0, n_hidden, w1, w2, w3, h1, h2, rgb, grad_rgb, grad_x, grad_logit, gw1, gb1, gw2, gb2, gw3, gb3, workspace, st);
        default: return tr_launch_bwd<128, 7>(x, n_pts, dim0, n_hidden, w1, w2, w3, h1, h2, rgb, grad_rgb, grad_x, grad_logit, gw1, gb1, gw2, gb2, gw3, gb3, workspace, st);
    }
}

// --------------------------------------------------------------------------------------------------------------------
// Distortion loss of flattened samples (run_sr.py:976-988 -> torch_efficient_distloss.flatten_eff_distloss(w, s, 1/n_max, ray_id)).
// Per ray, samples in ascending s:   L = sum_i [ interval/3 * w_i^2 + 2 w_i (s_i P_i - Q_i) ],  P/Q = EXCLUSIVE prefix sums of w / w s
// (= sum_ij w_i w_j |s_i - s_j| + 1/3 sum_i w_i^2 interval);   dL/dw_i = 2 (s_i (P_i - S_i) + (R_i - Q_i)) + 2/3 interval w_i with
// S/R the exclusive SUFFIX sums.  One wave per ray (its samples are contiguous: ray_id is ascending), segment bounds by binary
// search, 64 samples per scan step; value and gradient in one launch.  The caller divides by ray_id.max()+1 like the extension.
// --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int64_t tr_lower_bound(const int64_t* __restrict__ a, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_distloss(const float* __restrict__ w, const float* __restrict__ m, const int64_t* __restrict__ ray_id,
                                                  int64_t n_pts, int64_t n_rays, float interval, float* __restrict__ ray_loss,
                                                  float* __restrict__ grad) {
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = k4_lane();
    if (ray >= n_rays) return;
    const int64_t i0 = tr_lower_bound(ray_id, 0, n_pts, ray);
    const int64_t i1 = tr_lower_bound(ray_id, i0, n_pts, ray + 1);
    float tw = 0.f, twm = 0.f;
    for (int64_t i = i0 + lane; i < i1; i += 64) { const float wi = w[i]; tw += wi; twm = fmaf(wi, m[i], twm); }
    tw = tr_wave_sum(tw); twm = tr_wave_sum(twm);
    const float third = interval * (1.f / 3.f);
    float cw = 0.f, cwm = 0.f, lsum = 0.f;
    for (int64_t base = i0; base < i1; base += 64) {
        const int64_t i = base + lane;
        const bool act = i < i1;
        const float wi = act ? w[i] : 0.f, mi = act ? m[i] : 0.f, wm = wi * mi;
        float sw = wi, swm = wm;                                   // inclusive scans
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float u = __shfl_up(sw, off), v = __shfl_up(swm, off);
            if (lane >= off) { sw += u; swm += v; }
        }
        const float ew = __shfl_up(sw, 1), ewm = __shfl_up(swm, 1);
        const float wp = cw + (lane ? ew : 0.f), wmp = cwm + (lane ? ewm : 0.f);       // exclusive prefixes
        const float wsuf = tw - (wp + wi), wmsuf = twm - (wmp + wm);                    // exclusive suffixes
        if (act) {
            lsum += third * wi * wi + 2.f * wi * (mi * wp - wmp);
            grad[i] = 2.f * (mi * (wp - wsuf) + (wmsuf - wmp)) + 2.f * third * wi;
        }
        cw += __shfl(sw, 63); cwm += __shfl(swm, 63);
    }
    lsum = tr_wave_sum(lsum);
    if (lane == 0) ray_loss[ray] = lsum;
}

extern "C" int k4_distortion_loss(const float* w, const float* s, const int64_t* ray_id, int64_t n_pts, int64_t n_rays, float interval,
                                  float* ray_loss, float* grad_w, void* stream) {
    if (n_pts < 0 || n_rays < 0) return K4_ERR_BAD_ARG;
    if (n_rays == 0) return K4_OK;
    if (!ray_loss || (n_pts > 0 && (!w || !s || !ray_id || !grad_w))) return K4_ERR_BAD_ARG;
    const int64_t threads = n_rays * 64;
    hipLaunchKernelGGL(k_distloss, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, s, ray_id, n_pts, n_rays, interval,
                       ray_loss, grad_w);
    return k4_check_launch();
}

// --------------------------------------------------------------------------------------------------------------------
// The elementwise loss terms of the joint iteration (run_sr.py:877-995) in one launch each way:
//   photo   = weight_main * mean |rgb_feature - target|                                    (:877-881, F.l1_loss)
//   l1      = mean |rgb_sr - rgb_hr|,  psnr_sr = -10 log10 mean (clamp(rgb_sr, 0, 1) - rgb_hr)^2   (:925, :930)
//   entropy = -mean(p log p + (1 - p) log(1 - p)) * weight_entropy_last,  p = clamp(alphainv_last, 1e-6, 1 - 1e-6)      (:962-964)
//   rgbper  = weight_rgbper * sum_m [ |raw_rgb[m] - target[ray_id[m]]|^2 * weights[m] ] / n_rays                      (:993-995, weights detached)
// As tensor-library ops these are ~20 launches of 2-4 us forward and as many in the backward pass, each behind ~10 us of dispatch: the GPU sat
// idle for most of the 0.4 ms between the decoder's forward and its backward pass (profiles/r06_joint_phase_events.md, section 12).
// Per element the fp32 expression of the op sequence; the sums in fp64 (block partials, one atomic per block and term).
// --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double jl_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ void jl_block_add(double v, double* __restrict__ dst, double* sm) {
    v = jl_wave_sum(v);
    const int lane = k4_lane(), wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, sm[0] + sm[1] + sm[2] + sm[3]);
}
struct JlSeg { int b1, b2, b3, b4; };        // first block of the segments 1 .. 3 and the grid size (segment 0 starts at block 0)
__device__ __forceinline__ float jl_clamp_lo() { return 1e-6f; }
__device__ __forceinline__ float jl_clamp_hi() { return (float)(1.0 - 1e-6); }

__global__ __launch_bounds__(256) void k_joint_losses_sum(k4_joint_losses D, JlSeg S, double* __restrict__ acc) {
    __shared__ double sm[4];
    const int b = blockIdx.x;
    if (b < S.b1) {                                                        // photo
        const int64_t i = (int64_t)b * 256 + threadIdx.x;
        const float v = i < D.n_rays * 3 ? fabsf(D.rgb_feature[i] - D.target[i]) : 0.f;
        jl_block_add((double)v, acc + 0, sm);
    } else if (b < S.b2) {                                                 // decoder output against the 4x target
        const int64_t e = (int64_t)(b - S.b1) * 256 + threadIdx.x;
        float v = 0.f, q = 0.f;
        if (e < D.n_hr * 3) {
            const int64_t px = e / 3;
            const int c = (int)(e - px * 3);
            const float x = D.rgb_sr[c * D.sr_cstride + px * D.sr_pstride], t = D.target_4x[e];
            v = fabsf(x - t);
            const float d = fminf(fmaxf(x, 0.f), 1.f) - t;
            q = d * d;
        }
        jl_block_add((double)v, acc + 1, sm);
        jl_block_add((double)q, acc + 2, sm);
    } else if (b < S.b3) {                                                 // entropy of the last transmittance
        const int64_t i = (int64_t)(b - S.b2) * 256 + threadIdx.x;
        float v = 0.f;
        if (i < D.n_rays) {
            const float p = fminf(fmaxf(D.alphainv_last[i], jl_clamp_lo()), jl_clamp_hi());
            v = p * logf(p) + (1.f - p) * logf(1.f - p);
        }
        jl_block_add((double)v, acc + 3, sm);
    } else {                                                               // per-sample colour
        const int64_t m = (int64_t)(b - S.b3) * 256 + threadIdx.x;
        float v = 0.f;
        if (m < D.n_pts) {
            const int64_t r = D.ray_id[m];
            const float d0 = D.raw_rgb[m * 3] - D.target[r * 3], d1 = D.raw_rgb[m * 3 + 1] - D.target[r * 3 + 1], d2 = D.raw_rgb[m * 3 + 2] - D.target[r * 3 + 2];
            v = ((d0 * d0 + d1 * d1) + d2 * d2) * D.weights[m];
        }
        jl_block_add((double)v, acc + 4, sm);
    }
}
__global__ void k_joint_losses_finish(k4_joint_losses D, double* __restrict__ acc, float* __restrict__ terms, float* __restrict__ total) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float photo = D.weight_main * (float)(acc[0] / (double)(D.n_rays * 3));
    const float l1 = (float)(acc[1] / (double)(D.n_hr * 3));
    const float psnr = -10.f * log10f((float)(acc[2] / (double)(D.n_hr * 3)));
    const float ent = D.alphainv_last ? -(float)(acc[3] / (double)D.n_rays) * D.weight_entropy_last : 0.f;
    const float per = D.raw_rgb ? (D.weight_rgbper * (float)acc[4]) / (float)D.n_rays : 0.f;
    terms[0] = photo; terms[1] = l1; terms[2] = psnr; terms[3] = ent; terms[4] = per;
    float t = photo + l1;
    if (D.alphainv_last) t += ent;
    if (D.raw_rgb) t += per;
    total[0] = t;
    for (int k = 0; k < 5; ++k) acc[k] = 0.0;
}
__global__ __launch_bounds__(256) void k_joint_losses_bwd(k4_joint_losses D, JlSeg S, const float* __restrict__ gtot, float* __restrict__ g_feat, float* __restrict__ g_sr,
                                                          float* __restrict__ g_alpha, float* __restrict__ g_raw) {
    const float g = gtot[0];
    const int b = blockIdx.x;
    if (b < S.b1) {
        const int64_t i = (int64_t)b * 256 + threadIdx.x;
        if (g_feat && i < D.n_rays * 3) {
            const float gs = (g * D.weight_main) / (float)(D.n_rays * 3), d = D.rgb_feature[i] - D.target[i];
            g_feat[i] = d > 0.f ? gs : d < 0.f ? -gs : 0.f;
        }
    } else if (b < S.b2) {
        const int64_t e = (int64_t)(b - S.b1) * 256 + threadIdx.x;
        if (g_sr && e < D.n_hr * 3) {
            const int64_t px = e / 3;
            const int c = (int)(e - px * 3);
            const int64_t o = c * D.sr_cstride + px * D.sr_pstride;
            const float gs = g / (float)(D.n_hr * 3), d = D.rgb_sr[o] - D.target_4x[e];
            g_sr[o] = d > 0.f ? gs : d < 0.f ? -gs : 0.f;
        }
    } else if (b < S.b3) {
        const int64_t i = (int64_t)(b - S.b2) * 256 + threadIdx.x;
        if (g_alpha && i < D.n_rays) {
            const float a = D.alphainv_last[i];
            const bool inside = a >= jl_clamp_lo() && a <= jl_clamp_hi();
            const float p = fminf(fmaxf(a, jl_clamp_lo()), jl_clamp_hi());
            const float ge = -(g * D.weight_entropy_last) / (float)D.n_rays;
            g_alpha[i] = inside ? ge * (logf(p) - logf(1.f - p)) : 0.f;
        }
    } else {
        const int64_t m = (int64_t)(b - S.b3) * 256 + threadIdx.x;
        if (g_raw && m < D.n_pts) {
            const int64_t r = D.ray_id[m];
            const float gw = ((g / (float)D.n_rays) * D.weight_rgbper) * D.weights[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) g_raw[m * 3 + c] = gw * (2.f * (D.raw_rgb[m * 3 + c] - D.target[r * 3 + c]));
        }
    }
}
static int jl_segments(const k4_joint_losses* d, JlSeg* S) {
    if (!d || d->n_rays <= 0 || d->n_hr <= 0 || d->n_pts < 0 || !d->rgb_feature || !d->target || !d->rgb_sr || !d->target_4x) return K4_ERR_BAD_ARG;
    if (d->raw_rgb && (!d->weights || !d->ray_id)) return K4_ERR_BAD_ARG;
    const int64_t s0 = (d->n_rays * 3 + 255) / 256, s1 = (d->n_hr * 3 + 255) / 256, s2 = d->alphainv_last ? (d->n_rays + 255) / 256 : 0,
                  s3 = d->raw_rgb ? (d->n_pts + 255) / 256 : 0;
    if (s0 + s1 + s2 + s3 > 0x3fffffffLL) return K4_ERR_BAD_ARG;
    S->b1 = (int)s0; S->b2 = (int)(s0 + s1); S->b3 = (int)(s0 + s1 + s2); S->b4 = (int)(s0 + s1 + s2 + s3);
    return K4_OK;
}
extern "C" int k4_joint_losses_fwd(const k4_joint_losses* d, double* acc, float* terms, float* total, void* stream) {
    JlSeg S;
    const int rc = jl_segments(d, &S);
    if (rc) return rc;
    if (!acc || !terms || !total || (((uintptr_t)acc) & 7u)) return K4_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_joint_losses_sum, dim3((unsigned)S.b4), dim3(256), 0, (hipStream_t)stream, *d, S, acc);
    hipLaunchKernelGGL(k_joint_losses_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, *d, acc, terms, total);
    return k4_check_launch();
}
extern "C" int k4_joint_losses_bwd(const k4_joint_losses* d, const float* grad_total, float* grad_rgb_feature, float* grad_rgb_sr, float* grad_alphainv_last,
                                   float* grad_raw_rgb, void* stream) {
    JlSeg S;
    const int rc = jl_segments(d, &S);
    if (rc) return rc;
    if (!grad_total || (grad_alphainv_last && !d->alphainv_last) || (grad_raw_rgb && !d->raw_rgb)) return K4_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_joint_losses_bwd, dim3((unsigned)S.b4), dim3(256), 0, (hipStream_t)stream, *d, S, grad_total, grad_rgb_feature, grad_rgb_sr, grad_alphainv_last,
                       grad_raw_rgb);
    return k4_check_launch();
}

// --------------------------------------------------------------------------------------------------------------------
// The colour MLP's input of DirectMPIGO's training forward (lib/dmpigo.py:360-374) in ONE launch:
//   x[i] = [ vox_emb[i] (C) | pe_spa (3) | sin(pe_spa f) (3 P) | cos(pe_spa f) (3 P) | viewdirs[r] (3) | sin(viewdirs[r] g) (3 V) | cos(...) (3 V) ],  r = ray_id[i],
//   pe_spa[j] = ((p[2 - j] - min[2 - j]) / (max[2 - j] - min[2 - j])) * 2 - 1   (the reference's op sequence: sub, div, flip, mul, sub -- one rounding each),
//   pe_emb column 3 + d P + f = pe_spa[d] * posfreq[f] (`(pe_spa.unsqueeze(-1) * posfreq).flatten(-2)`), likewise the view directions.
// As PyTorch ops this was 16 launches (most of them on empty tensors in the LLFF configuration: no frequencies) issued by a host that paces the phase:
// the GPU needs 0.9 ms for the marcher's training forward and spent 1.9 ms in it (profiles/r06_joint_phase_events.md).
// --------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rgbnet_input_mpi(const float* __restrict__ vox, int C, const float* __restrict__ pts, const float* __restrict__ vd,
                                                          const int64_t* __restrict__ ray_id, int64_t n, const float* __restrict__ lo, const float* __restrict__ hi,
                                                          const float* __restrict__ pf, int P, const float* __restrict__ vf, int V, float* __restrict__ x, int dim0) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * dim0) return;
    const int64_t i = t / dim0;
    int col = (int)(t - i * dim0);
    float v;
    if (col < C) v = vox[i * C + col];
    else {
        col -= C;
        const int npe = 3 + 6 * P;
        const bool view = col >= npe;
        if (view) col -= npe;
        const int F = view ? V : P;
        const float* const fr = view ? vf : pf;
        // component d of the embedded 3-vector; which = 0 plain, 1 sin, 2 cos; f = frequency index
        int which = 0, d = col, f = 0;
        if (col >= 3) { const int c2 = col - 3; which = 1 + c2 / (3 * F); const int r = c2 % (3 * F); d = r / F; f = r % F; }
        float base;
        if (view) base = vd[ray_id[i] * 3 + d];
        else {
            const int a = 2 - d;                                                   // .flip((-1,))
            const float q = __fdiv_rn(__fsub_rn(pts[i * 3 + a], lo[a]), __fsub_rn(hi[a], lo[a]));
            base = __fsub_rn(__fmul_rn(q, 2.f), 1.f);
        }
        if (which == 0) v = base;
        else {
            const float arg = __fmul_rn(base, fr[f]);
            v = which == 1 ? sinf(arg) : cosf(arg);
        }
    }
    x[t] = v;
}
extern "C" int k4_rgbnet_input_mpi(const float* vox_emb, int32_t channels, const float* ray_pts, const float* viewdirs, const int64_t* ray_id, int64_t n_pts,
                                   const float* xyz_min, const float* xyz_max, const float* posfreq, int32_t n_posfreq, const float* viewfreq, int32_t n_viewfreq,
                                   float* x, int32_t dim0, void* stream) {
    if (n_pts < 0 || channels < 0 || n_posfreq < 0 || n_viewfreq < 0 || dim0 != channels + 3 + 6 * n_posfreq + 3 + 6 * n_viewfreq) return K4_ERR_BAD_ARG;
    if (n_pts == 0) return K4_OK;
    if (!ray_pts || !viewdirs || !ray_id || !xyz_min || !xyz_max || !x || (channels > 0 && !vox_emb) || (n_posfreq > 0 && !posfreq) || (n_viewfreq > 0 && !viewfreq)) return K4_ERR_BAD_ARG;
    const int64_t total = n_pts * dim0;
    if ((total + 255) / 256 > 0x7fffffffLL) return K4_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_rgbnet_input_mpi, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vox_emb, channels, ray_pts, viewdirs, ray_id, n_pts,
                       xyz_min, xyz_max, posfreq, n_posfreq, viewfreq, n_viewfreq, x, dim0);
    return k4_check_launch();
}

// --------------------------------------------------------------------------------------------------------------------
// SFTLayer of the VC-Decoder under autograd (lib/sr_esrnet.py:112-123): y = x * (scale(cond) + 1) + shift(cond) with
//   scale = W1s lrelu(W0s c + b0s) + b1s,  shift = W1h lrelu(W0h c + b0h) + b1h       (1x1 convolutions 32 -> 32 -> C, slope 0.2).
// The joint training step evaluates 36 of these per iteration on a 64x64 patch; as four convolution Functions + PyTorch elementwise
// glue each was ~13 launches forward and ~30 backward of 4-15 us apiece -- two thirds of the decoder's ~2500 launches.  Here:
//   k_sft_train_fwd : one launch, nothing saved but the inputs (the backward recomputes the 64 hidden activations);
//   k_sft_train_bwd : gx, gc (gradient of the 32-channel condition map) and the per-workgroup partial sums of the eight weight /
//                     bias gradients (4x2 register blocks over the pixel axis, cf. k_rgbnet_bwd: no atomics, fixed order);
//   k_sft_train_reduce : partials -> the eight gradient tensors.
// Tile = 64 pixels, lane = pixel, wave = a sixteenth of the neurons / channels; weights are the nn.Conv2d tensors as stored
// ([out][in] row-major), read with wave-uniform indices (scalar loads).  Exact fp32 FMA chains.
// --------------------------------------------------------------------------------------------------------------------
#define SFT_G 32                          // condition channels = hidden width
#define SFT_GB (SFT_G + 4)                // + ones row (bias gradient) + 3 zero rows: a multiple of 4
// Workgroup = 16 waves (4 per SIMD) on one 64-pixel tile.  A 64x64 training patch is 64 tiles -> 64 workgroups on 256 CUs whatever the
// workgroup size, so the time of a launch is the time of ONE tile: with 4 waves (one per SIMD, nothing to hide the LDS / scalar-load
// latencies behind) the backward took 60 us and the forward 25 us per layer, 36 layers per joint iteration (profiles/
// r05_joint_iteration_kernel_stats.csv).  16 waves split the same per-element FMA chains four times finer: same values, same order.
#define SFT_T 1024
#define SFT_NW (SFT_T / 64)

// [n][stride] global rows (K channels from each) -> LDS [K][TR_LS]; samples >= nv read as 0
__device__ __forceinline__ void sft_load_tile(const float* __restrict__ g, int64_t base, int stride, int K, int nv, float* lds, int t) {
    for (int i = t; i < 64 * K; i += SFT_T) {
        const int s = i / K, k = i - s * K;
        lds[k * TR_LS + s] = s < nv ? g[(base + s) * stride + k] : 0.f;
    }
}
__device__ __forceinline__ void sft_store_tile(const float* lds, float* __restrict__ g, int64_t base, int stride, int K, int nv, int t) {
    for (int i = t; i < 64 * K; i += SFT_T) {
        const int s = i / K, k = i - s * K;
        if (s < nv) g[(base + s) * stride + k] = lds[k * TR_LS + s];
    }
}
// The weight matrices are read below with dependent wave-uniform (scalar) loads.  Inside a training iteration every layer's weights are cold -- Adam
// rewrote them, 35 other layers ran since -- and each scalar miss went to HBM one after the other: a layer took 47-57 us in the iteration against 28 us
// when the same layer is launched repeatedly.  One coalesced vector sweep over the 24 KB at kernel entry brings them to L2 in a single round trip
// (the values are summed into a number that is never stored).
template <int C>
__device__ __forceinline__ float sft_warm_weights(const float* __restrict__ w0s, const float* __restrict__ w0h, const float* __restrict__ w1s, const float* __restrict__ w1h, int t) {
    float d = 0.f;
    if ((((uintptr_t)w0s | (uintptr_t)w0h | (uintptr_t)w1s | (uintptr_t)w1h) & 15u) != 0) return d;       // (workgroup-uniform)
    for (int i = t * 4; i < SFT_G * SFT_G; i += SFT_T * 4) {
        const float4 a = *reinterpret_cast<const float4*>(w0s + i), b = *reinterpret_cast<const float4*>(w0h + i);
        d += a.x + b.x;
    }
    for (int i = t * 4; i < C * SFT_G; i += SFT_T * 4) {
        const float4 a = *reinterpret_cast<const float4*>(w1s + i), b = *reinterpret_cast<const float4*>(w1h + i);
        d += a.x + b.x;
    }
    return d;
}
// hidden activations of both branches: hs[j] = lrelu(b0[j] + sum_k w0[j][k] c[k]), j < 32 scale branch, j >= 32 shift branch
__device__ __forceinline__ void sft_hidden(const float* cs, k4_cptr w0s, k4_cptr b0s, k4_cptr w0h, k4_cptr b0h, float slope,
                                           float* as, float* ah, int wv, int lane) {
    for (int j0 = wv * (2 * SFT_G / SFT_NW); j0 < (wv + 1) * (2 * SFT_G / SFT_NW); j0 += 4) {
        const bool sh = j0 >= SFT_G;
        const k4_cptr w = (sh ? w0h : w0s) + (j0 & (SFT_G - 1)) * SFT_G, b = (sh ? b0h : b0s) + (j0 & (SFT_G - 1));
        float a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3];
        for (int k = 0; k < SFT_G; ++k) {
            const float v = cs[k * TR_LS + lane];
            a0 = fmaf(v, w[k], a0); a1 = fmaf(v, w[SFT_G + k], a1); a2 = fmaf(v, w[2 * SFT_G + k], a2); a3 = fmaf(v, w[3 * SFT_G + k], a3);
        }
        float* const out = (sh ? ah : as) + (j0 & (SFT_G - 1)) * TR_LS + lane;
        out[0] = a0 > 0.f ? a0 : a0 * slope; out[TR_LS] = a1 > 0.f ? a1 : a1 * slope;
        out[2 * TR_LS] = a2 > 0.f ? a2 : a2 * slope; out[3 * TR_LS] = a3 > 0.f ? a3 : a3 * slope;
    }
}

template <int C>
__global__ __launch_bounds__(SFT_T) void k_sft_train_fwd(const float* __restrict__ x, int x_stride, const float* __restrict__ cond, int c_stride, int64_t n,
                                                       const float* __restrict__ w0s, const float* __restrict__ b0s, const float* __restrict__ w1s, const float* __restrict__ b1s,
                                                       const float* __restrict__ w0h, const float* __restrict__ b0h, const float* __restrict__ w1h, const float* __restrict__ b1h,
                                                       float slope, float* __restrict__ y, int y_stride,
                                                       const float* __restrict__ res, int res_stride, float res_scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const cs = smem;                       // [32][LS]
    float* const as = cs + SFT_G * TR_LS;         // [32][LS]
    float* const ah = as + SFT_G * TR_LS;         // [32][LS]
    float* const xs = ah + SFT_G * TR_LS;         // [C][LS]  x, then y in place
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nv = (n - base) < 64 ? (int)(n - base) : 64;
    const float warm = sft_warm_weights<C>(w0s, w0h, w1s, w1h, t);            // (see sft_warm_weights)
    sft_load_tile(cond, base, c_stride, SFT_G, nv, cs, t);
    sft_load_tile(x, base, x_stride, C, nv, xs, t);
    if (warm == 1.2345e38f) xs[0] = warm;                                     // (keeps the loads alive; never true)
    __syncthreads();
    sft_hidden(cs, k4_const(w0s), k4_const(b0s), k4_const(w0h), k4_const(b0h), slope, as, ah, wv, lane);
    __syncthreads();
    const k4_cptr w1sc = k4_const(w1s), w1hc = k4_const(w1h), b1sc = k4_const(b1s), b1hc = k4_const(b1h);
    for (int co = wv * (C / SFT_NW); co < (wv + 1) * (C / SFT_NW); co += 2) {
        float s0 = b1sc[co], s1 = b1sc[co + 1], h0 = b1hc[co], h1 = b1hc[co + 1];
        for (int k = 0; k < SFT_G; ++k) {
            const float a = as[k * TR_LS + lane], b = ah[k * TR_LS + lane];
            s0 = fmaf(a, w1sc[co * SFT_G + k], s0); s1 = fmaf(a, w1sc[(co + 1) * SFT_G + k], s1);
            h0 = fmaf(b, w1hc[co * SFT_G + k], h0); h1 = fmaf(b, w1hc[(co + 1) * SFT_G + k], h1);
        }
        xs[co * TR_LS + lane] = __fadd_rn(__fmul_rn(xs[co * TR_LS + lane], s0 + 1.f), h0);                  // x * (scale + 1) + shift   (lib/sr_esrnet.py:123)
        xs[(co + 1) * TR_LS + lane] = __fadd_rn(__fmul_rn(xs[(co + 1) * TR_LS + lane], s1 + 1.f), h1);   // two roundings, as the reference's ops
    }
    __syncthreads();
    if (res) {                                             // y = sft(x) * res_scale + res: the RRDB's skip connection (lib/sr_esrnet.py:181), two roundings as the reference's two ops
        for (int i = t; i < 64 * C; i += SFT_T) {
            const int s = i / C, k = i - s * C;
            if (s < nv) y[(base + s) * y_stride + k] = __fadd_rn(__fmul_rn(xs[k * TR_LS + s], res_scale), res[(base + s) * res_stride + k]);
        }
    } else sft_store_tile(xs, y, base, y_stride, C, nv, t);
}

template <int C>
struct SftBwdLayout {
    // LDS rows: cs [GB] | as [GB] | ah [GB] | gzs [G] | gzh [G] | gs [C] | gy [C] | gx [C] | gc [G]
    static constexpr int ROWS = 3 * SFT_GB + 2 * SFT_G + 3 * C + SFT_G;
    // partial sums: P1s [C][GB] | P1h [C][GB] | P0s [G][GB] | P0h [G][GB]   (column G = the bias gradient)
    static constexpr int N_PART = 2 * C * SFT_GB + 2 * SFT_G * SFT_GB;
    static constexpr int N_BLK = N_PART / 8;                  // 4 x 2 register blocks
    static constexpr int MAXB = (N_BLK + SFT_T - 1) / SFT_T;
};

template <int C>
__global__ __launch_bounds__(SFT_T) void k_sft_train_bwd(const float* __restrict__ x, int x_stride, const float* __restrict__ cond, int c_stride,
                                                       const float* __restrict__ gyg, int gy_stride, int64_t n,
                                                       const float* __restrict__ w0s, const float* __restrict__ b0s, const float* __restrict__ w1s, const float* __restrict__ b1s,
                                                       const float* __restrict__ w0h, const float* __restrict__ b0h, const float* __restrict__ w1h,
                                                       float slope, float* __restrict__ gxg, float* __restrict__ gcg, float* __restrict__ part,
                                                       const float* __restrict__ gxa, int gxa_stride, int gc_acc, int gx_lrelu, float gy_scale) {
    typedef SftBwdLayout<C> L;
    constexpr int MAXB = L::MAXB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const cs = smem;
    float* const as = cs + SFT_GB * TR_LS;
    float* const ah = as + SFT_GB * TR_LS;
    float* const gzs = ah + SFT_GB * TR_LS;
    float* const gzh = gzs + SFT_G * TR_LS;
    float* const gs = gzh + SFT_G * TR_LS;        // x on arrival, then gy * x
    float* const gy = gs + C * TR_LS;
    float* const gx = gy + C * TR_LS;
    float* const gc = gx + C * TR_LS;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
#ifndef K4_SFT_NO_WARM
    const float warm = sft_warm_weights<C>(w0s, w0h, w1s, w1h, t);
#else
    const float warm = 0.f;
#endif
    for (int i = t; i < L::ROWS * TR_LS; i += SFT_T) smem[i] = 0.f;           // the zero padding rows stay zero for the whole kernel

    TrBlock blk[MAXB];
    float acc[MAXB][8];
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
        int b = t + SFT_T * q;
        blk[q].aoff = -1;
        constexpr int NB1 = (C / 4) * (SFT_GB / 2), NB0 = (SFT_G / 4) * (SFT_GB / 2);
        if (b < 2 * NB1) {
            const bool sh = b >= NB1;
            b -= sh ? NB1 : 0;
            const int ar = b / (SFT_GB / 2), bc = b - ar * (SFT_GB / 2);
            blk[q] = {(int)((sh ? gy : gs) - smem) + 4 * ar * TR_LS, (int)((sh ? ah : as) - smem) + 2 * bc * TR_LS,
                      (sh ? C * SFT_GB : 0) + 4 * ar * SFT_GB + 2 * bc, SFT_GB};
        } else if (b < 2 * NB1 + 2 * NB0) {
            b -= 2 * NB1;
            const bool sh = b >= NB0;
            b -= sh ? NB0 : 0;
            const int ar = b / (SFT_GB / 2), bc = b - ar * (SFT_GB / 2);
            blk[q] = {(int)((sh ? gzh : gzs) - smem) + 4 * ar * TR_LS, (int)(cs - smem) + 2 * bc * TR_LS,
                      2 * C * SFT_GB + (sh ? SFT_G * SFT_GB : 0) + 4 * ar * SFT_GB + 2 * bc, SFT_GB};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
    }
    __syncthreads();

    const k4_cptr w0sc = k4_const(w0s), w0hc = k4_const(w0h), w1sc = k4_const(w1s), w1hc = k4_const(w1h), b1sc = k4_const(b1s);
    const int64_t n_tiles = (n + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * 64;
        const int nv = (n - base) < 64 ? (int)(n - base) : 64;
        sft_load_tile(cond, base, c_stride, SFT_G, nv, cs, t);
        sft_load_tile(x, base, x_stride, C, nv, gs, t);
        if (gy_scale != 1.f) {                             // the layer's output went through `* res_scale + res` (k4_sft_train_fwd_ex): gy = res_scale * incoming, one rounding
            for (int i = t; i < 64 * C; i += SFT_T) {
                const int s = i / C, k = i - s * C;
                gy[k * TR_LS + s] = s < nv ? gyg[(base + s) * gy_stride + k] * gy_scale : 0.f;
            }
        } else sft_load_tile(gyg, base, gy_stride, C, nv, gy, t);
        if (t < 64) {
            const float one = t < nv ? 1.f : 0.f;
            cs[SFT_G * TR_LS + t] = one; as[SFT_G * TR_LS + t] = one; ah[SFT_G * TR_LS + t] = one;
        }
        __syncthreads();
        sft_hidden(cs, w0sc, k4_const(b0s), w0hc, k4_const(b0h), slope, as, ah, wv, lane);
        __syncthreads();
        // gx = gy * (scale + 1);  gs = gy * x   (the shift branch's output gradient is gy itself)
        if (!gxg) {                                            // the chain's own launch (k_sft_train_bwd_gx) has written grad_x: gs only
            for (int co = wv * (C / SFT_NW); co < (wv + 1) * (C / SFT_NW); ++co) gs[co * TR_LS + lane] = gs[co * TR_LS + lane] * gy[co * TR_LS + lane];
        } else
        for (int co = wv * (C / SFT_NW); co < (wv + 1) * (C / SFT_NW); co += 2) {
            float s0 = b1sc[co], s1 = b1sc[co + 1];
            for (int k = 0; k < SFT_G; ++k) {
                const float a = as[k * TR_LS + lane];
                s0 = fmaf(a, w1sc[co * SFT_G + k], s0); s1 = fmaf(a, w1sc[(co + 1) * SFT_G + k], s1);
            }
            const float g0 = gy[co * TR_LS + lane], g1 = gy[(co + 1) * TR_LS + lane];
            const float x0 = gs[co * TR_LS + lane], x1 = gs[(co + 1) * TR_LS + lane];
            float o0 = g0 * (s0 + 1.f), o1 = g1 * (s1 + 1.f);
            if (gx_lrelu) {                                 // x = lrelu(z): the gradient in front of the activation (k4_lrelu_bwd's arithmetic)
                o0 = x0 > 0.f ? o0 : o0 * slope; o1 = x1 > 0.f ? o1 : o1 * slope;
            }
            gx[co * TR_LS + lane] = o0; gx[(co + 1) * TR_LS + lane] = o1;
            gs[co * TR_LS + lane] = x0 * g0; gs[(co + 1) * TR_LS + lane] = x1 * g1;
        }
        __syncthreads();
        // gz = lrelu'(z) * (W1^T g):  wave w -> hidden neurons 4 (w & 7) .. +3 of the scale (w < 8) or shift (w >= 8) branch
        {
            static_assert(SFT_NW == 16, "wave -> (branch, 4 neurons)");
            const bool sh = wv >= 8;
            const int k0 = (wv & 7) * 4;
            const float* const src = sh ? gy : gs;               // the shift branch's output gradient is gy itself
            const k4_cptr w1 = sh ? w1hc : w1sc;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int co = 0; co < C; ++co) {
                const float a = src[co * TR_LS + lane];
                s0 = fmaf(a, w1[co * SFT_G + k0], s0); s1 = fmaf(a, w1[co * SFT_G + k0 + 1], s1);
                s2 = fmaf(a, w1[co * SFT_G + k0 + 2], s2); s3 = fmaf(a, w1[co * SFT_G + k0 + 3], s3);
            }
            const float sv[4] = {s0, s1, s2, s3};
            const float* const act = sh ? ah : as;
            float* const out = sh ? gzh : gzs;
#pragma unroll
            for (int e = 0; e < 4; ++e)                          // leaky_relu_backward: grad where the INPUT is > 0, slope * grad elsewhere (lrelu(z) > 0 <=> z > 0)
                out[(k0 + e) * TR_LS + lane] = act[(k0 + e) * TR_LS + lane] > 0.f ? sv[e] : sv[e] * slope;
        }
        __syncthreads();
        // gc = W0s^T gz_s + W0h^T gz_h:  wave w -> condition channels 2w, 2w+1
        {
            const int k0 = wv * (SFT_G / SFT_NW);
            float a0 = 0.f, a1 = 0.f;
            for (int j = 0; j < SFT_G; ++j) {
                const float u = gzs[j * TR_LS + lane], v = gzh[j * TR_LS + lane];
                a0 = fmaf(u, w0sc[j * SFT_G + k0], a0); a1 = fmaf(u, w0sc[j * SFT_G + k0 + 1], a1);
                a0 = fmaf(v, w0hc[j * SFT_G + k0], a0); a1 = fmaf(v, w0hc[j * SFT_G + k0 + 1], a1);
            }
            gc[k0 * TR_LS + lane] = a0; gc[(k0 + 1) * TR_LS + lane] = a1;
        }
        // weight-gradient blocks: acc[a][b] += sum_s A[a][s] * B[b][s]   (A, B rows were complete at the barrier above)
#pragma unroll
        for (int q = 0; q < MAXB; ++q) {
            if (blk[q].aoff < 0) continue;
            const float4* const A = reinterpret_cast<const float4*>(smem + blk[q].aoff);
            const float4* const B = reinterpret_cast<const float4*>(smem + blk[q].boff);
            for (int s4 = 0; s4 < 16; ++s4) {
                float4 av[4], bv[2];
#pragma unroll
                for (int r = 0; r < 4; ++r) av[r] = A[r * (TR_LS / 4) + s4];
#pragma unroll
                for (int r = 0; r < 2; ++r) bv[r] = B[r * (TR_LS / 4) + s4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float v = acc[q][a * 2 + b];
                        v = fmaf(av[a].x, bv[b].x, v); v = fmaf(av[a].y, bv[b].y, v); v = fmaf(av[a].z, bv[b].z, v); v = fmaf(av[a].w, bv[b].w, v);
                        acc[q][a * 2 + b] = v;
                    }
            }
        }
        __syncthreads();
        if (!gxg) {
        } else if (gxa) {                                  // grad_x = this layer's gradient + gxa (the block's skip connection: one add kernel less per block)
            for (int i = t; i < 64 * C; i += SFT_T) {
                const int s = i / C, k = i - s * C;
                if (s < nv) gxg[(base + s) * C + k] = gx[k * TR_LS + s] + gxa[(base + s) * gxa_stride + k];
            }
        } else sft_store_tile(gx, gxg, base, C, C, nv, t);
        if (gc_acc) {                                      // grad_cond ACCUMULATES: every SFT layer of the decoder reads the same condition map
            for (int i = t; i < 64 * SFT_G; i += SFT_T) {
                const int s = i / SFT_G, k = i - s * SFT_G;
                if (s < nv) gcg[(base + s) * SFT_G + k] += gc[k * TR_LS + s];
            }
        } else sft_store_tile(gc, gcg, base, SFT_G, SFT_G, nv, t);
        __syncthreads();
    }
    float* const mine = part + (size_t)blockIdx.x * L::N_PART;
    if (warm == 1.2345e38f) mine[0] = warm;                                   // (keeps the warm-up loads alive; never true for finite weights of this size)
#pragma unroll
    for (int q = 0; q < MAXB; ++q) {
        if (blk[q].aoff < 0) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) mine[blk[q].ooff + a * blk[q].ostride + b] = acc[q][a * 2 + b];
    }
}

// The part of the layer's backward the CHAIN waits for: grad_x = gy * (scale(cond) + 1) [LeakyReLU mask] [+ gxa] -- the scale branch's 32 hidden neurons and its
// output layer, the arithmetic of k_sft_train_bwd's first two stages value for value (same FMA chains, same roundings).  Everything else of the layer
// (both branches' hidden gradients, the condition gradient, the eight parameter gradients) is read by nothing on the chain before the CondNet's backward pass /
// the optimizer: a caller with a third stream runs it there as k_sft_train_bwd with gxg == NULL (k4_sft_train_bwd_rest) while the chain moves on.  In the joint
// iteration the 36 full launches were 28-37 us each on the chain (1.2 ms of a 3.4 ms backward pass); this one is the size of the forward launch.
template <int C>
__global__ __launch_bounds__(SFT_T) void k_sft_train_bwd_gx(const float* __restrict__ x, int x_stride, const float* __restrict__ cond, int c_stride,
                                                          const float* __restrict__ gyg, int gy_stride, int64_t n,
                                                          const float* __restrict__ w0s, const float* __restrict__ b0s, const float* __restrict__ w1s, const float* __restrict__ b1s,
                                                          float slope, float* __restrict__ gxg, const float* __restrict__ gxa, int gxa_stride, int gx_lrelu, float gy_scale,
                                                          float* __restrict__ gs_out, float gs_scale, const float* __restrict__ add2, float* __restrict__ sum2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const cs = smem;                       // [32][LS]
    float* const as = cs + SFT_G * TR_LS;         // [32][LS]
    float* const gy = as + SFT_G * TR_LS;         // [C][LS]  gy, then grad_x in place
    float* const xs = gy + C * TR_LS;             // [C][LS]  (gx_lrelu only)
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t base = (int64_t)blockIdx.x * 64;
    const int nv = (n - base) < 64 ? (int)(n - base) : 64;
    const float warm = sft_warm_weights<C>(w0s, w0s, w1s, w1s, t);
    sft_load_tile(cond, base, c_stride, SFT_G, nv, cs, t);
    if (gy_scale != 1.f) {
        for (int i = t; i < 64 * C; i += SFT_T) {
            const int s = i / C, k = i - s * C;
            gy[k * TR_LS + s] = s < nv ? gyg[(base + s) * gy_stride + k] * gy_scale : 0.f;
        }
    } else sft_load_tile(gyg, base, gy_stride, C, nv, gy, t);
    if (gx_lrelu) sft_load_tile(x, base, x_stride, C, nv, xs, t);
    if (warm == 1.2345e38f) cs[0] = warm;                                     // (keeps the loads alive; never true)
    __syncthreads();
    {
        static_assert(SFT_NW == 16, "wave -> 2 hidden neurons of the scale branch");
        const int j0 = wv * 2;
        const k4_cptr w = k4_const(w0s) + j0 * SFT_G, b = k4_const(b0s) + j0;
        float a0 = b[0], a1 = b[1];
        for (int k = 0; k < SFT_G; ++k) {
            const float v = cs[k * TR_LS + lane];
            a0 = fmaf(v, w[k], a0); a1 = fmaf(v, w[SFT_G + k], a1);
        }
        as[j0 * TR_LS + lane] = a0 > 0.f ? a0 : a0 * slope; as[(j0 + 1) * TR_LS + lane] = a1 > 0.f ? a1 : a1 * slope;
    }
    __syncthreads();
    const k4_cptr w1sc = k4_const(w1s), b1sc = k4_const(b1s);
    for (int co = wv * (C / SFT_NW); co < (wv + 1) * (C / SFT_NW); co += 2) {
        float s0 = b1sc[co], s1 = b1sc[co + 1];
        for (int k = 0; k < SFT_G; ++k) {
            const float a = as[k * TR_LS + lane];
            s0 = fmaf(a, w1sc[co * SFT_G + k], s0); s1 = fmaf(a, w1sc[(co + 1) * SFT_G + k], s1);
        }
        const float g0 = gy[co * TR_LS + lane], g1 = gy[(co + 1) * TR_LS + lane];
        float o0 = g0 * (s0 + 1.f), o1 = g1 * (s1 + 1.f);
        if (gx_lrelu) {
            const float x0 = xs[co * TR_LS + lane], x1 = xs[(co + 1) * TR_LS + lane];
            o0 = x0 > 0.f ? o0 : o0 * slope; o1 = x1 > 0.f ? o1 : o1 * slope;
        }
        gy[co * TR_LS + lane] = o0; gy[(co + 1) * TR_LS + lane] = o1;
    }
    __syncthreads();
    // grad_x [+ gxa], and two by-products the chain's next launches would otherwise compute in launches of their own: gs_out = grad_x * gs_scale (the next dense
    // block's g5 = 0.2 grad_out) and sum2 = grad_x + add2 (the sum of the two gradients an RRDB's input receives) -- one rounding each, as k_scale_f32 / k_add_f32
    for (int i = t; i < 64 * C; i += SFT_T) {
        const int s = i / C, k = i - s * C;
        if (s >= nv) continue;
        float v = gy[k * TR_LS + s];
        if (gxa) v = v + gxa[(base + s) * gxa_stride + k];
        const int64_t o = (base + s) * C + k;
        gxg[o] = v;
        if (gs_out) gs_out[o] = v * gs_scale;
        if (sum2) sum2[o] = v + add2[o];
    }
}

// partials [n_wg][P1s | P1h | P0s | P0h] -> gw1s [C][32], gb1s [C], gw1h, gb1h, gw0s [32][32], gb0s [32], gw0h, gb0h; fixed order (see k_rgbnet_reduce)
__global__ __launch_bounds__(256) void k_sft_train_reduce(const float* __restrict__ part, int n_wg, int C, float* __restrict__ gw1s, float* __restrict__ gb1s,
                                                          float* __restrict__ gw1h, float* __restrict__ gb1h, float* __restrict__ gw0s, float* __restrict__ gb0s,
                                                          float* __restrict__ gw0h, float* __restrict__ gb0h) {
    __shared__ float red[TR_RED_SLICES][TR_RED_ELEMS + 1];
    const int e = threadIdx.x & (TR_RED_ELEMS - 1), sl = threadIdx.x / TR_RED_ELEMS;
    const int i = blockIdx.x * TR_RED_ELEMS + e;
    const int per_row = SFT_G + 1;
    const int n1 = C * per_row, n0 = SFT_G * per_row, n_part = 2 * C * SFT_GB + 2 * SFT_G * SFT_GB;
    const bool live = i < 2 * n1 + 2 * n0;
    int off = 0;
    float* dst = nullptr;
    if (live) {
        int r = i;
        if (r < 2 * n1) {
            const bool sh = r >= n1;
            r -= sh ? n1 : 0;
            const int a = r / per_row, j = r - a * per_row;
            off = (sh ? C * SFT_GB : 0) + a * SFT_GB + j;
            dst = j < SFT_G ? (sh ? gw1h : gw1s) + a * SFT_G + j : (sh ? gb1h : gb1s) + a;
        } else {
            r -= 2 * n1;
            const bool sh = r >= n0;
            r -= sh ? n0 : 0;
            const int a = r / per_row, j = r - a * per_row;
            off = 2 * C * SFT_GB + (sh ? SFT_G * SFT_GB : 0) + a * SFT_GB + j;
            dst = j < SFT_G ? (sh ? gw0h : gw0s) + a * SFT_G + j : (sh ? gb0h : gb0s) + a;
        }
    }
    const int per = (n_wg + TR_RED_SLICES - 1) / TR_RED_SLICES;
    const int g0 = sl * per, g1 = min(g0 + per, n_wg);
    float s = 0.f;
    if (live)
        for (int g = g0; g < g1; ++g) s += part[(size_t)g * n_part + off];
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && live) {
        float v = red[0][e];
#pragma unroll
        for (int q = 1; q < TR_RED_SLICES; ++q) v += red[q][e];
        *dst = v;
    }
}

static int sft_bwd_grid(int64_t n) {
    const int64_t tiles = (n + 63) / 64;
    const int64_t cap = (int64_t)k4_num_cus();
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

extern "C" int64_t k4_sft_train_bwd_workspace_bytes(int64_t n_pix, int32_t channels) {
    if ((channels != 32 && channels != 64) || n_pix < 0) return -1;
    return (int64_t)sft_bwd_grid(n_pix) * (channels == 64 ? SftBwdLayout<64>::N_PART : SftBwdLayout<32>::N_PART) * (int64_t)sizeof(float);
}

extern "C" int k4_sft_train_fwd_ex(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, int64_t n_pix, int32_t channels,
                                   const float* w0s, const float* b0s, const float* w1s, const float* b1s,
                                   const float* w0h, const float* b0h, const float* w1h, const float* b1h,
                                   float slope, float* y, int32_t y_stride, const float* res, int32_t res_stride, float res_scale, void* stream) {
    if (channels != 32 && channels != 64) return K4_ERR_UNSUPPORTED;
    if (n_pix < 0 || x_stride < channels || y_stride < channels || cond_stride < SFT_G || (res && res_stride < channels)) return K4_ERR_BAD_ARG;
    if (!w0s || !b0s || !w1s || !b1s || !w0h || !b0h || !w1h || !b1h) return K4_ERR_BAD_ARG;
    if (n_pix == 0) return K4_OK;
    if (!x || !cond || !y) return K4_ERR_BAD_ARG;
    return k4_taped(stream, [=](void* stream) -> int {                    // recordable (k4_tape.hip), as every entry point of the decoder's training pass below
        hipStream_t st = (hipStream_t)stream;
        const dim3 grid((unsigned)((n_pix + 63) / 64)), block(SFT_T);
        const size_t lds = (size_t)(3 * SFT_G + channels) * TR_LS * sizeof(float);
        if (channels == 64) hipLaunchKernelGGL(k_sft_train_fwd<64>, grid, block, lds, st, x, x_stride, cond, cond_stride, n_pix, w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h, slope, y, y_stride, res, res_stride, res_scale);
        else hipLaunchKernelGGL(k_sft_train_fwd<32>, grid, block, lds, st, x, x_stride, cond, cond_stride, n_pix, w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h, slope, y, y_stride, res, res_stride, res_scale);
        return k4_check_launch();
    });
}
extern "C" int k4_sft_train_fwd(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, int64_t n_pix, int32_t channels,
                                const float* w0s, const float* b0s, const float* w1s, const float* b1s,
                                const float* w0h, const float* b0h, const float* w1h, const float* b1h,
                                float slope, float* y, int32_t y_stride, void* stream) {
    return k4_sft_train_fwd_ex(x, x_stride, cond, cond_stride, n_pix, channels, w0s, b0s, w1s, b1s, w0h, b0h, w1h, b1h, slope, y, y_stride, nullptr, 0, 1.f, stream);
}

static int k4_wait_stream(hipStream_t waiter, hipStream_t signaller, hipStream_t waiter2 = nullptr);
// red_st: the stream of the partial sums' reduction (the eight parameter gradients).  Nothing on the chain reads those: with red_st != st the reduction is
// forked to red_st (which the caller joins before the gradients are read), one launch and one launch gap less on the chain per layer.
template <int C>
static int sft_launch_bwd(const float* x, int xs, const float* cond, int cs, const float* gy, int gys, int64_t n,
                          const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                          float slope, float* gx, float* gc, float* ws, float* const* gout, const float* gxa, int gxa_stride, int gc_acc, int gx_lrelu, float gy_scale, hipStream_t st,
                          hipStream_t red_st, bool do_reduce = true) {
    typedef SftBwdLayout<C> L;
    const size_t lds = (size_t)L::ROWS * TR_LS * sizeof(float);
    K4_ENSURE_DYN_LDS((k_sft_train_bwd<C>), lds);
    const int grid = sft_bwd_grid(n);
    hipLaunchKernelGGL((k_sft_train_bwd<C>), dim3(grid), dim3(SFT_T), lds, st, x, xs, cond, cs, gy, gys, n, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, gx, gc, ws, gxa, gxa_stride, gc_acc, gx_lrelu, gy_scale);
    int rc = k4_check_launch();
    if (rc || !do_reduce) return rc;
    const int total = 2 * C * (SFT_G + 1) + 2 * SFT_G * (SFT_G + 1);
    if (red_st != st) {
        rc = k4_wait_stream(red_st, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_sft_train_reduce, dim3((total + TR_RED_ELEMS - 1) / TR_RED_ELEMS), dim3(256), 0, red_st, ws, grid, C,
                       gout[0], gout[1], gout[2], gout[3], gout[4], gout[5], gout[6], gout[7]);
    return k4_check_launch();
}

static int sft_bwd_entry(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                         int64_t n_pix, int32_t channels,
                         const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                         float slope, float* grad_x, float* grad_cond,
                         float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                         float* workspace, int64_t workspace_bytes,
                         const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale,
                         void* side_stream, bool do_reduce, void* stream, bool rest_only = false) {
    if (channels != 32 && channels != 64) return K4_ERR_UNSUPPORTED;
    if (n_pix <= 0 || x_stride < channels || gy_stride < channels || cond_stride < SFT_G || (grad_x_add && gxa_stride < channels)) return K4_ERR_BAD_ARG;
    if (!x || !cond || !grad_y || (!grad_x && !rest_only) || !grad_cond || !w0s || !b0s || !w1s || !b1s || !w0h || !b0h || !w1h) return K4_ERR_BAD_ARG;
    if (!gw0s || !gb0s || !gw1s || !gb1s || !gw0h || !gb0h || !gw1h || !gb1h) return K4_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < k4_sft_train_bwd_workspace_bytes(n_pix, channels)) return K4_ERR_BAD_ARG;
    return k4_taped(stream, [=](void* stream) -> int {
        float* const gout[8] = {gw1s, gb1s, gw1h, gb1h, gw0s, gb0s, gw0h, gb0h};
        hipStream_t st = (hipStream_t)stream;
        const int acc = accumulate_grad_cond != 0;
        hipStream_t red = side_stream ? (hipStream_t)side_stream : st;
        if (channels == 64) return sft_launch_bwd<64>(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, grad_x, grad_cond, workspace, gout, grad_x_add, gxa_stride, acc, grad_x_lrelu != 0, grad_y_scale, st, red, do_reduce);
        return sft_launch_bwd<32>(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, grad_x, grad_cond, workspace, gout, grad_x_add, gxa_stride, acc, grad_x_lrelu != 0, grad_y_scale, st, red, do_reduce);
    });
}
extern "C" int k4_sft_train_bwd_side(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                                     int64_t n_pix, int32_t channels,
                                     const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                                     float slope, float* grad_x, float* grad_cond,
                                     float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                                     float* workspace, int64_t workspace_bytes,
                                     const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale,
                                     void* side_stream, void* stream) {
    return sft_bwd_entry(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, channels, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, grad_x, grad_cond,
                         gw0s, gb0s, gw1s, gb1s, gw0h, gb0h, gw1h, gb1h, workspace, workspace_bytes, grad_x_add, gxa_stride, accumulate_grad_cond, grad_x_lrelu,
                         grad_y_scale, side_stream, true, stream);
}
// The two halves of k4_sft_train_bwd_ex as calls of their own: the kernel that produces grad_x / grad_cond and the per-workgroup partial sums (`workspace`), and
// the reduction of the partials to the eight parameter gradients -- a caller that batches its side-stream work issues the reductions of several layers behind
// ONE fork (every hipEventRecord on the chain's stream costs ~7 us of the chain's time: profiles/r06_joint_phase_events.md, section 8).
extern "C" int k4_sft_train_bwd_main(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                                     int64_t n_pix, int32_t channels,
                                     const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                                     float slope, float* grad_x, float* grad_cond, float* workspace, int64_t workspace_bytes,
                                     const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale, void* stream) {
    float* const dummy = workspace;                                 // (the entry's NULL checks; the reduction that would write the gradients is not launched)
    return sft_bwd_entry(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, channels, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, grad_x, grad_cond,
                         dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, workspace, workspace_bytes, grad_x_add, gxa_stride, accumulate_grad_cond, grad_x_lrelu,
                         grad_y_scale, nullptr, false, stream);
}
// The layer's backward in two launches for a caller with a third stream (k_sft_train_bwd_gx above): k4_sft_train_bwd_gx on the chain, k4_sft_train_bwd_rest
// (+ k4_sft_train_reduce) wherever the caller likes, ordered behind the producer of grad_y and in front of the first reader of grad_cond / the gradients.
extern "C" int k4_sft_train_bwd_gx(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                                   int64_t n_pix, int32_t channels, const float* w0s, const float* b0s, const float* w1s, const float* b1s,
                                   float slope, float* grad_x, const float* grad_x_add, int32_t gxa_stride, int32_t grad_x_lrelu, float grad_y_scale,
                                   float* grad_x_scaled, float scaled_by, const float* add2, float* sum2, void* stream) {
    if (channels != 32 && channels != 64) return K4_ERR_UNSUPPORTED;
    if (n_pix <= 0 || gy_stride < channels || cond_stride < SFT_G || (grad_x_add && gxa_stride < channels) || (grad_x_lrelu && (!x || x_stride < channels))) return K4_ERR_BAD_ARG;
    if ((add2 != nullptr) != (sum2 != nullptr)) return K4_ERR_BAD_ARG;
    if (!cond || !grad_y || !grad_x || !w0s || !b0s || !w1s || !b1s) return K4_ERR_BAD_ARG;
    return k4_taped(stream, [=](void* stream) -> int {
        hipStream_t st = (hipStream_t)stream;
        const dim3 grid((unsigned)((n_pix + 63) / 64)), block(SFT_T);
        const size_t lds = (size_t)(2 * SFT_G + 2 * channels) * TR_LS * sizeof(float);
        if (channels == 64) {
            K4_ENSURE_DYN_LDS((k_sft_train_bwd_gx<64>), lds);
            hipLaunchKernelGGL(k_sft_train_bwd_gx<64>, grid, block, lds, st, x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, w0s, b0s, w1s, b1s, slope, grad_x, grad_x_add, gxa_stride, grad_x_lrelu, grad_y_scale, grad_x_scaled, scaled_by, add2, sum2);
        } else {
            K4_ENSURE_DYN_LDS((k_sft_train_bwd_gx<32>), lds);
            hipLaunchKernelGGL(k_sft_train_bwd_gx<32>, grid, block, lds, st, x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, w0s, b0s, w1s, b1s, slope, grad_x, grad_x_add, gxa_stride, grad_x_lrelu, grad_y_scale, grad_x_scaled, scaled_by, add2, sum2);
        }
        return k4_check_launch();
    });
}
extern "C" int k4_sft_train_bwd_rest(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                                     int64_t n_pix, int32_t channels,
                                     const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                                     float slope, float* grad_cond, float* workspace, int64_t workspace_bytes, int32_t accumulate_grad_cond, float grad_y_scale, void* stream) {
    float* const dummy = workspace;
    return sft_bwd_entry(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, channels, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, nullptr, grad_cond,
                         dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, workspace, workspace_bytes, nullptr, 0, accumulate_grad_cond, 0,
                         grad_y_scale, nullptr, false, stream, true);
}
extern "C" int k4_sft_train_reduce(const float* workspace, int64_t n_pix, int32_t channels,
                                   float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h, void* stream) {
    if ((channels != 32 && channels != 64) || n_pix <= 0 || !workspace || !gw0s || !gb0s || !gw1s || !gb1s || !gw0h || !gb0h || !gw1h || !gb1h) return K4_ERR_BAD_ARG;
    return k4_taped(stream, [=](void* stream) -> int {
        const int grid = sft_bwd_grid(n_pix);
        const int total = 2 * channels * (SFT_G + 1) + 2 * SFT_G * (SFT_G + 1);
        hipLaunchKernelGGL(k_sft_train_reduce, dim3((total + TR_RED_ELEMS - 1) / TR_RED_ELEMS), dim3(256), 0, (hipStream_t)stream, workspace, grid, channels,
                           gw1s, gb1s, gw1h, gb1h, gw0s, gb0s, gw0h, gb0h);
        return k4_check_launch();
    });
}
extern "C" int k4_sft_train_bwd_ex(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                                   int64_t n_pix, int32_t channels,
                                   const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                                   float slope, float* grad_x, float* grad_cond,
                                   float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                                   float* workspace, int64_t workspace_bytes,
                                   const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale, void* stream) {
    return k4_sft_train_bwd_side(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, channels, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, grad_x, grad_cond,
                                 gw0s, gb0s, gw1s, gb1s, gw0h, gb0h, gw1h, gb1h, workspace, workspace_bytes, grad_x_add, gxa_stride, accumulate_grad_cond, grad_x_lrelu,
                                 grad_y_scale, nullptr, stream);
}
extern "C" int k4_sft_train_bwd(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                                int64_t n_pix, int32_t channels,
                                const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                                float slope, float* grad_x, float* grad_cond,
                                float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                                float* workspace, int64_t workspace_bytes, void* stream) {
    return k4_sft_train_bwd_ex(x, x_stride, cond, cond_stride, grad_y, gy_stride, n_pix, channels, w0s, b0s, w1s, b1s, w0h, b0h, w1h, slope, grad_x, grad_cond,
                               gw0s, gb0s, gw1s, gb1s, gw0h, gb0h, gw1h, gb1h, workspace, workspace_bytes, nullptr, 0, 0, 0, 1.f, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// LeakyReLU backward on a channel slice (dense-block gradient image of the decoder's training graph, lib/sr_train.py K4RDB)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lrelu_bwd(const float* g, int g_stride, const float* __restrict__ y, int y_stride, int64_t n_pix,
                                                   int c4, float slope, float* out, int out_stride) {          // out may alias g
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix * c4) return;
    const int64_t p = i / c4;
    const int c = (int)(i - p * c4) * 4;
    const float4 yv = *reinterpret_cast<const float4*>(y + p * y_stride + c);
    float4 gv = *reinterpret_cast<const float4*>(g + p * g_stride + c);
    gv.x *= yv.x > 0.f ? 1.f : slope; gv.y *= yv.y > 0.f ? 1.f : slope; gv.z *= yv.z > 0.f ? 1.f : slope; gv.w *= yv.w > 0.f ? 1.f : slope;
    *reinterpret_cast<float4*>(out + p * out_stride + c) = gv;
}

extern "C" int k4_lrelu_bwd(const float* grad, int32_t g_stride, const float* y, int32_t y_stride, int64_t n_pix, int32_t channels, float slope,
                            float* out, int32_t out_stride, void* stream) {
    if (!grad || !y || !out || n_pix < 0 || channels <= 0 || (channels & 3) || (g_stride & 3) || (y_stride & 3) || (out_stride & 3) ||
        g_stride < channels || y_stride < channels || out_stride < channels || ((uintptr_t)grad & 15) || ((uintptr_t)y & 15) || ((uintptr_t)out & 15))
        return K4_ERR_BAD_ARG;
    if (n_pix == 0) return 0;
    return k4_taped(stream, [=](void* stream) -> int {
        const int64_t n = n_pix * (channels / 4);
        hipLaunchKernelGGL(k_lrelu_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad, g_stride, y, y_stride, n_pix, channels / 4, slope,
                           out, out_stride);
        return k4_check_launch();
    });
}

// ---------------------------------------------------------------------------------------------------------------------
// A dense block of the training graph issued natively (include/k4nerf.h, k4_rdb_train): the launch sequences of lib/sr_train.py K4RDB.forward /
// .backward, call for call -- the entry points below are the ones the host would call; what goes away is ~26 Python-to-C transitions per block.
// ---------------------------------------------------------------------------------------------------------------------
// Events come from a per-device ring created once (a dense block forks / joins the side stream six times: creating and destroying an event each
// time was ~12 runtime calls per block on the host path that paces the joint iteration).  Re-recording an event does not disturb a wait already
// queued on its previous record (hipStreamWaitEvent captures the record it was called after); the ring is far longer than a block's forks in flight.
#define K4_EV_RING 256
#define K4_EV_DEVS 16
static hipEvent_t k4_ev_ring[K4_EV_DEVS][K4_EV_RING];
static std::atomic<unsigned> k4_ev_next[K4_EV_DEVS];
static std::once_flag k4_ev_once[K4_EV_DEVS];
static bool k4_ev_ok[K4_EV_DEVS];
static int k4_wait_stream(hipStream_t waiter, hipStream_t signaller, hipStream_t waiter2) {      // everything queued on `signaller` so far completes before what `waiter` (and `waiter2`: the same event record) gets next
    if (waiter2 == signaller || waiter2 == waiter) waiter2 = nullptr;
    if (waiter == signaller) { if (!waiter2) return 0; waiter = waiter2; waiter2 = nullptr; }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= K4_EV_DEVS) {                                     // beyond the ring table: an event of its own
        hipEvent_t ev;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
        e = hipEventRecord(ev, signaller);
        if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
        if (e == hipSuccess && waiter2) e = hipStreamWaitEvent(waiter2, ev, 0);
        const hipError_t d = hipEventDestroy(ev);                          // (released by the runtime once the recorded work has completed)
        return (int)(e != hipSuccess ? e : d);
    }
    std::call_once(k4_ev_once[dev], [dev]() {
        bool ok = true;
        for (int i = 0; i < K4_EV_RING; ++i) ok = ok && hipEventCreateWithFlags(&k4_ev_ring[dev][i], hipEventDisableTiming) == hipSuccess;
        k4_ev_ok[dev] = ok;
    });
    if (!k4_ev_ok[dev]) return K4_ERR_BAD_ARG;
    hipEvent_t ev = k4_ev_ring[dev][k4_ev_next[dev].fetch_add(1u) % K4_EV_RING];
    e = hipEventRecord(ev, signaller);
    if (e == hipSuccess) e = hipStreamWaitEvent(waiter, ev, 0);
    if (e == hipSuccess && waiter2) e = hipStreamWaitEvent(waiter2, ev, 0);
    return (int)e;
}
template <bool VEC>
__global__ __launch_bounds__(256) void k_scale_f32(const float* __restrict__ in, float s, float* __restrict__ out, int64_t units) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= units) return;
    if (VEC) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
    } else out[i] = in[i] * s;
}
static bool k4_rdb_ok(const k4_rdb_train* p, bool bwd) {
    if (!p || p->H <= 0 || p->W <= 0 || p->g != 32 || (p->nf != 32 && p->nf != 64) || !p->t || !p->c || !p->buf || !p->x4) return false;
    for (int k = 0; k < 5; ++k) if (!p->w_fwd[k] && !bwd) return false;
    for (int k = 0; k < 8; ++k) if (!p->sft0[k] || !p->sft1[k]) return false;
    if (!bwd) return p->out != nullptr;
    if (!p->g5 || !p->G || !p->gx4 || !p->gx0 || !p->ws0 || !p->ws1 || (!p->gc_acc && (!p->gc0 || !p->gc1)) || p->dwdb_span_floats < 0) return false;
    for (int k = 0; k < 5; ++k) if (!p->w_bwd[k] || !p->b_bwd[k] || !p->dwdb[k]) return false;
    for (int k = 0; k < 8; ++k) if (!p->gsft0[k] || !p->gsft1[k]) return false;
    return true;
}
#define K4_RDB_TRY(CALL) do { const int rc_ = (CALL); if (rc_ != 0) return rc_; } while (0)

extern "C" int k4_rdb_train_fwd(const k4_rdb_train* p_in, void* stream) {
    if (!k4_rdb_ok(p_in, false)) return K4_ERR_BAD_ARG;
    const k4_rdb_train desc = *p_in;                                       // (a tape keeps the descriptor by value)
    return k4_taped(stream, [desc](void* stream) -> int {
    const k4_rdb_train* const p = &desc;
    const int H = p->H, W = p->W, nf = p->nf, g = p->g, bw = nf + 4 * g;
    const int64_t n = (int64_t)H * W;
    K4_RDB_TRY(k4_sft_train_fwd(p->t, nf, p->c, 32, n, nf, p->sft0[0], p->sft0[1], p->sft0[2], p->sft0[3], p->sft0[4], p->sft0[5], p->sft0[6], p->sft0[7],
                                0.2f, p->buf, bw, stream));
    for (int k = 1; k <= 3; ++k)
        K4_RDB_TRY(k4_conv2d_nhwc_bf16x6(p->buf, nf + (k - 1) * g, bw, p->w_fwd[k - 1], p->b_fwd[k - 1], 3, p->buf + nf + (k - 1) * g, g, bw, H, W,
                                         K4_EPI_LRELU | K4_CONV_SMALL, 0.2f, nullptr, 0, 0.f, nullptr, 0, stream));
    K4_RDB_TRY(k4_conv2d_nhwc_bf16x6(p->buf, nf + 3 * g, bw, p->w_fwd[3], p->b_fwd[3], 3, p->x4, g, g, H, W, K4_EPI_LRELU | K4_CONV_SMALL, 0.2f, nullptr, 0, 0.f, nullptr, 0, stream));
    K4_RDB_TRY(k4_sft_train_fwd(p->x4, g, p->c, 32, n, g, p->sft1[0], p->sft1[1], p->sft1[2], p->sft1[3], p->sft1[4], p->sft1[5], p->sft1[6], p->sft1[7],
                                0.2f, p->buf + nf + 3 * g, bw, stream));
    return k4_conv2d_nhwc_bf16x6(p->buf, bw, bw, p->w_fwd[4], p->b_fwd[4], 3, p->out, nf, nf, H, W, K4_EPI_RES | K4_CONV_SMALL, 0.2f, p->t, nf, 0.2f, nullptr, 0, stream);
    });
}

extern "C" int k4_rdb_train_bwd(const k4_rdb_train* p_in, void* stream) {
    if (!k4_rdb_ok(p_in, true)) return K4_ERR_BAD_ARG;
    const k4_rdb_train desc = *p_in;
    return k4_taped(stream, [desc](void* stream) -> int {
    const k4_rdb_train* const p = &desc;
    const int H = p->H, W = p->W, nf = p->nf, g = p->g, bw = nf + 4 * g;
    const int64_t n = (int64_t)H * W;
    hipStream_t main_s = (hipStream_t)stream, side = p->side_stream ? (hipStream_t)p->side_stream : main_s;
    // every exit joins the side stream back into `stream`: after a failed launch the caller frees the gradient buffers on `stream` while weight
    // gradients already forked to the side stream may still be running on them
    int rc = 0;
    const bool fl = p->fused_lrelu != 0;
    if (p->g5_from_gx0_add && !p->gx0_add) return K4_ERR_BAD_ARG;
#undef K4_RDB_TRY
#define K4_RDB_TRY(CALL) do { rc = (CALL); if (rc != 0) goto join; } while (0)
    // a weight gradient on the side stream: forked behind everything queued on the main stream so far (= the producer of the gradient slice it reads)
    // defer_side: every side-stream launch of the block (zero-fill, five weight gradients, two SFT reductions) is issued at the END of the block behind ONE fork.
    // Forking per weight gradient put a hipEventRecord on the chain's stream in front of every dgrad launch: ~7 us of the chain's time each, 1.0 ms of a 4.1 ms
    // backward pass (profiles/r06_joint_phase_events.md, section 8).  Needs a caller that does not join per block (no_join): the weight gradients of block b then
    // run beside the chain of block b - 1.
    const bool defer = p->defer_side != 0 && side != main_s;
    // aux_stream (ABI 14; with defer_side + no_join + gc_acc): the chain runs only the grad_x part of the two SFT layers' backward (k4_sft_train_bwd_gx); the rest of
    // both layers goes to aux_stream at the end of the block.  The caller joins aux_stream before grad_cond's first reader and before the optimizer.
    hipStream_t aux = (hipStream_t)p->aux_stream;
    const bool split = defer && p->no_join != 0 && p->gc_acc != nullptr && aux != nullptr && aux != main_s;
    struct WgradQ { int cin; const float* gy; int cout; int gys; int k; } wq[5];
    int nwq = 0;
#define K4_RDB_WGRAD(CIN, GY, COUT, GYS, K) do { \
        if (defer) { wq[nwq].cin = (CIN); wq[nwq].gy = (GY); wq[nwq].cout = (COUT); wq[nwq].gys = (GYS); wq[nwq].k = (K); ++nwq; break; } \
        K4_RDB_TRY(k4_wait_stream(side, main_s)); \
        if (p->dwdb_span_floats > 0) K4_RDB_TRY(k4_conv2d_wgrad_dbias_bf16x6_acc(p->buf, (CIN), bw, (GY), (COUT), (GYS), 3, H, W, p->dwdb[K], (void*)side)); \
        else K4_RDB_TRY(k4_conv2d_wgrad_dbias_bf16x6(p->buf, (CIN), bw, (GY), (COUT), (GYS), 3, H, W, p->dwdb[K], (void*)side)); } while (0)
    // G[:, 0:COUT'] (+)= dgrad: the output is its own residual
    // MASK (fused_lrelu): the last 32 channels this launch completes are x_k's gradient slice -- LeakyReLU backward from buf's slice in the epilogue
#define K4_RDB_DGRAD(K, SRC, SS, CIN_OF_LAYER, ACC, MASK) \
        K4_RDB_TRY(k4_conv2d_nhwc_bf16x6((SRC), (K) == 4 ? nf : g, (SS), p->w_bwd[K], p->b_bwd[K], 3, p->G, (CIN_OF_LAYER), bw, H, W, \
                                         ((ACC) ? K4_EPI_RES : 0u) | ((MASK) ? K4_EPI_LRELU_BWD : 0u) | K4_CONV_SMALL, 0.2f, \
                                         (ACC) ? p->G : nullptr, (ACC) ? bw : 0, 1.f, (MASK) ? p->buf : nullptr, (MASK) ? bw : 0, stream))
    // dwdb_span_floats > 0: the five [dW | dbias] buffers are one span starting at dwdb_span: ONE zero-fill on the side stream (ordered before every
    // weight gradient there) instead of one per layer
    if (p->dwdb_span_floats > 0) {
        if (!p->dwdb_span) return K4_ERR_BAD_ARG;
        if (!defer) {
            K4_RDB_TRY(k4_wait_stream(side, main_s));                              // (the span may be memory the main stream's earlier work still reads)
            K4_RDB_TRY(k4_zero_f32(p->dwdb_span, p->dwdb_span_floats, (void*)side));
        }
    }
    if (!split && (p->g5_next || p->gx0_add2 || p->gx0_sum2)) return K4_ERR_BAD_ARG;        // by-products of the chain's grad_x launch (aux_stream form only)
    if (p->g5_from_gx0_add && !p->g5_given) {                                       // g5 = 0.2 grad_out (conv5's output is scaled by 0.2 in the forward pass)
        const bool vec = (((uintptr_t)p->gx0_add | (uintptr_t)p->g5) & 15u) == 0;
        const int64_t units = vec ? n * nf / 4 : n * nf;
        if (vec) hipLaunchKernelGGL(k_scale_f32<true>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, main_s, p->gx0_add, 0.2f, (float*)p->g5, units);
        else hipLaunchKernelGGL(k_scale_f32<false>, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, main_s, p->gx0_add, 0.2f, (float*)p->g5, units);
        K4_RDB_TRY(k4_check_launch());
    }
    // conv5: out = 0.2 conv5(buf) + t
    K4_RDB_WGRAD(bw, p->g5, nf, nf, 4);
    K4_RDB_DGRAD(4, p->g5, nf, bw, false, false);                                  // G = dgrad (every channel)
    // xc1 = sft1(x4), x4 = lrelu(conv4(buf[:, 0:nf+3g]))
    if (split) K4_RDB_TRY(k4_sft_train_bwd_gx(p->x4, g, p->c, 32, p->G + nf + 3 * g, bw, n, g, p->sft1[0], p->sft1[1], p->sft1[2], p->sft1[3], 0.2f, p->gx4, nullptr, 0, fl, 1.f,
                                              nullptr, 0.f, nullptr, nullptr, stream));
    else if (defer) K4_RDB_TRY(k4_sft_train_bwd_main(p->x4, g, p->c, 32, p->G + nf + 3 * g, bw, n, g, p->sft1[0], p->sft1[1], p->sft1[2], p->sft1[3], p->sft1[4], p->sft1[5], p->sft1[6],
                                                0.2f, p->gx4, p->gc_acc ? p->gc_acc : p->gc1, p->ws1, p->ws1_bytes, nullptr, 0, p->gc_acc != nullptr, fl, 1.f, stream));
    else K4_RDB_TRY(k4_sft_train_bwd_side(p->x4, g, p->c, 32, p->G + nf + 3 * g, bw, n, g, p->sft1[0], p->sft1[1], p->sft1[2], p->sft1[3], p->sft1[4], p->sft1[5], p->sft1[6],
                                     0.2f, p->gx4, p->gc_acc ? p->gc_acc : p->gc1, p->gsft1[0], p->gsft1[1], p->gsft1[2], p->gsft1[3], p->gsft1[4], p->gsft1[5], p->gsft1[6], p->gsft1[7],
                                     p->ws1, p->ws1_bytes, nullptr, 0, p->gc_acc != nullptr, fl, 1.f, p->side_stream, stream));
    if (!fl) K4_RDB_TRY(k4_lrelu_bwd(p->gx4, g, p->x4, g, n, g, 0.2f, p->gx4, g, stream));
    K4_RDB_WGRAD(nf + 3 * g, p->gx4, g, g, 3);
    K4_RDB_DGRAD(3, p->gx4, g, nf + 3 * g, true, fl);                              // G[:, 0:nf+3g] += dgrad (+ the mask of x3's slice, its last 32 channels)
    for (int k = 3; k >= 1; --k) {                                                  // x_k = lrelu(conv_k(buf[:, 0:off]))
        const int off = nf + (k - 1) * g;
        if (!fl) K4_RDB_TRY(k4_lrelu_bwd(p->G + off, bw, p->buf + off, bw, n, g, 0.2f, p->G + off, bw, stream));
        K4_RDB_WGRAD(off, p->G + off, g, bw, k - 1);
        K4_RDB_DGRAD(k - 1, p->G + off, bw, off, true, fl && k > 1);               // (k == 1 completes xc0's slice: sft0's output, no activation)
    }
    // gx0_add != NULL: gx0 = the gradient through sft0 + gx0_add (the block's skip connection: grad_out itself)
    if (split) K4_RDB_TRY(k4_sft_train_bwd_gx(p->t, nf, p->c, 32, p->G, bw, n, nf, p->sft0[0], p->sft0[1], p->sft0[2], p->sft0[3], 0.2f, p->gx0, p->gx0_add, nf, 0, 1.f,
                                              p->g5_next, 0.2f, p->gx0_add2, p->gx0_sum2, stream));
    else if (defer) K4_RDB_TRY(k4_sft_train_bwd_main(p->t, nf, p->c, 32, p->G, bw, n, nf, p->sft0[0], p->sft0[1], p->sft0[2], p->sft0[3], p->sft0[4], p->sft0[5], p->sft0[6],
                                                0.2f, p->gx0, p->gc_acc ? p->gc_acc : p->gc0, p->ws0, p->ws0_bytes, p->gx0_add, nf, p->gc_acc != nullptr, 0, 1.f, stream));
    else K4_RDB_TRY(k4_sft_train_bwd_side(p->t, nf, p->c, 32, p->G, bw, n, nf, p->sft0[0], p->sft0[1], p->sft0[2], p->sft0[3], p->sft0[4], p->sft0[5], p->sft0[6],
                                     0.2f, p->gx0, p->gc_acc ? p->gc_acc : p->gc0, p->gsft0[0], p->gsft0[1], p->gsft0[2], p->gsft0[3], p->gsft0[4], p->gsft0[5], p->gsft0[6], p->gsft0[7],
                                     p->ws0, p->ws0_bytes, p->gx0_add, nf, p->gc_acc != nullptr, 0, 1.f, p->side_stream, stream));
    if (defer) {                                                                    // the block's side-stream work behind ONE fork (one event record, both streams wait for it)
        K4_RDB_TRY(k4_wait_stream(side, main_s, split ? aux : nullptr));
        // aux_wgrad: conv1's weight gradient (the first piece of the span) runs on the third stream behind the SFT layers' deferred launches -- per block the
        // weight gradients' stream carried ~130 us of launches against ~70 us on the third stream and ~110 us on the chain: it had become the pace of the pass
        const bool wg_aux = split && p->aux_wgrad != 0 && p->dwdb_span_floats > 0 && p->dwdb[0] == p->dwdb_span && p->dwdb[1] > p->dwdb[0] &&
                            (p->dwdb[1] - p->dwdb[0]) < p->dwdb_span_floats;
        const int64_t n_aux = wg_aux ? (int64_t)(p->dwdb[1] - p->dwdb[0]) : 0;
        if (p->dwdb_span_floats > 0) K4_RDB_TRY(k4_zero_f32(p->dwdb_span + n_aux, p->dwdb_span_floats - n_aux, (void*)side));
        for (int q = 0; q < nwq; ++q) {
            if (wg_aux && wq[q].k == 0) continue;
            if (p->dwdb_span_floats > 0) K4_RDB_TRY(k4_conv2d_wgrad_dbias_bf16x6_acc(p->buf, wq[q].cin, bw, wq[q].gy, wq[q].cout, wq[q].gys, 3, H, W, p->dwdb[wq[q].k], (void*)side));
            else K4_RDB_TRY(k4_conv2d_wgrad_dbias_bf16x6(p->buf, wq[q].cin, bw, wq[q].gy, wq[q].cout, wq[q].gys, 3, H, W, p->dwdb[wq[q].k], (void*)side));
        }
        // split: what the chain did not wait for -- both SFT layers' hidden / condition gradients and partial sums, then their reductions -- on the third stream,
        // in the order the one-launch form adds into gc_acc (sft1, then sft0)
        hipStream_t red = split ? aux : side;
        if (split) K4_RDB_TRY(k4_sft_train_bwd_rest(p->x4, g, p->c, 32, p->G + nf + 3 * g, bw, n, g, p->sft1[0], p->sft1[1], p->sft1[2], p->sft1[3], p->sft1[4], p->sft1[5], p->sft1[6],
                                                    0.2f, p->gc_acc, p->ws1, p->ws1_bytes, 1, 1.f, (void*)aux));
        K4_RDB_TRY(k4_sft_train_reduce(p->ws1, n, g, p->gsft1[0], p->gsft1[1], p->gsft1[2], p->gsft1[3], p->gsft1[4], p->gsft1[5], p->gsft1[6], p->gsft1[7], (void*)red));
        if (split) K4_RDB_TRY(k4_sft_train_bwd_rest(p->t, nf, p->c, 32, p->G, bw, n, nf, p->sft0[0], p->sft0[1], p->sft0[2], p->sft0[3], p->sft0[4], p->sft0[5], p->sft0[6],
                                                    0.2f, p->gc_acc, p->ws0, p->ws0_bytes, 1, 1.f, (void*)aux));
        K4_RDB_TRY(k4_sft_train_reduce(p->ws0, n, nf, p->gsft0[0], p->gsft0[1], p->gsft0[2], p->gsft0[3], p->gsft0[4], p->gsft0[5], p->gsft0[6], p->gsft0[7], (void*)red));
        if (wg_aux) {
            K4_RDB_TRY(k4_zero_f32(p->dwdb_span, n_aux, (void*)aux));
            for (int q = 0; q < nwq; ++q)
                if (wq[q].k == 0) K4_RDB_TRY(k4_conv2d_wgrad_dbias_bf16x6_acc(p->buf, wq[q].cin, bw, wq[q].gy, wq[q].cout, wq[q].gys, 3, H, W, p->dwdb[0], (void*)aux));
        }
    }
join:
#undef K4_RDB_WGRAD
#undef K4_RDB_DGRAD
    {
        // the wgrads are done before anything queued on `stream` after this call -- unless the caller joins itself, once (no_join; a failed launch joins anyway)
        const int rj = (p->no_join && rc == 0) ? 0 : k4_wait_stream(main_s, side);
        if (rc != 0 && split) (void)k4_wait_stream(main_s, aux);
        return rc != 0 ? rc : rj;
    }
    });
}
#undef K4_RDB_TRY

// Fork / join of a second stream for callers that place launches there themselves (lib/sr_tape.py: the weight gradients of the layers outside the
// dense blocks).  Recordable: `stream` is the call's main stream (the replaying stream on a replay), `side` stays as given.
// Side streams that REALLY run beside the streams they are meant to overlap.  The runtime multiplexes HIP streams onto a few hardware queues per
// priority level (four by default) and hands a new stream the least-referenced queue: in a process that has already created a handful of streams a
// "second stream" may share its hardware queue with the main stream -- its kernels then run in queue order with the main stream's, and every
// fork / join between the two becomes a full serialisation.  Measured (profiles/r06_joint_phase_events.md, section 6): the joint training iteration took
// 9.1 ms in a fresh process and 12.5-20.9 ms behind 6+ earlier streams, same code.  k4_stream_create_overlapping creates candidates until one demonstrably
// overlaps `main_stream` and every stream of `others`: a kernel that spins for ~200 us is launched on the one, a time stamp kernel on the candidate -- the
// stamp lands before the spin ends iff the two are on different hardware queues.  Rejected candidates are destroyed afterwards (while they live they keep
// their queue referenced, so the next candidate goes elsewhere).
__global__ void k_overlap_spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void k_overlap_stamp(unsigned long long* out) {
    if (threadIdx.x == 0) out[0] = wall_clock64();
}
static int k4_overlap_probe(hipStream_t busy, hipStream_t cand, unsigned long long* dbuf, bool* overlap) {       // dbuf: device, 3 x u64
    hipLaunchKernelGGL(k_overlap_spin, dim3(1), dim3(64), 0, busy, 20000ull, dbuf);                               // wall_clock64: 100 MHz -> 200 us
    hipLaunchKernelGGL(k_overlap_stamp, dim3(1), dim3(64), 0, cand, dbuf + 2);
    hipError_t e = hipStreamSynchronize(busy);
    if (e == hipSuccess) e = hipStreamSynchronize(cand);
    if (e != hipSuccess) return (int)e;
    unsigned long long h[3];
    e = hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    *overlap = h[2] < h[1];                                                                                       // stamped before the spin ended
    return k4_check_launch();
}
extern "C" int k4_streams_overlap(void* a, void* b) {          // 1: kernels of b run beside kernels of a; 0: they share a hardware queue; < 0: error
    if (a == b) return 0;
    unsigned long long* dbuf = nullptr;
    if (hipMalloc((void**)&dbuf, 3 * sizeof(unsigned long long)) != hipSuccess) return -1;
    bool ov = false;
    const int rc = k4_overlap_probe((hipStream_t)a, (hipStream_t)b, dbuf, &ov);
    (void)hipFree(dbuf);
    return rc != 0 ? -1 : (ov ? 1 : 0);
}
extern "C" void* k4_stream_create_overlapping(void* main_stream, void* const* others, int32_t n_others, int32_t low_priority) {
    if (n_others < 0 || (n_others > 0 && !others)) return nullptr;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    unsigned long long* dbuf = nullptr;
    if (hipMalloc((void**)&dbuf, 3 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    constexpr int MAX_TRIES = 12;
    hipStream_t tried[MAX_TRIES];
    int n_tried = 0;
    hipStream_t good = nullptr;
    while (n_tried < MAX_TRIES && !good) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithPriority(&c, hipStreamNonBlocking, low_priority ? least : 0) != hipSuccess) break;
        tried[n_tried++] = c;
        bool ok = true;
        int rc = k4_overlap_probe((hipStream_t)main_stream, c, dbuf, &ok);
        for (int i = 0; rc == 0 && ok && i < n_others; ++i) rc = k4_overlap_probe((hipStream_t)others[i], c, dbuf, &ok);
        if (rc != 0) break;
        if (ok) good = c;
    }
    if (!good && n_tried > 0) good = tried[n_tried - 1];        // none verified (one hardware queue? profiling layer?): a plain second stream, as before
    for (int i = 0; i < n_tried; ++i)
        if (tried[i] != good) (void)hipStreamDestroy(tried[i]);
    (void)hipFree(dbuf);
    return (void*)good;
}
extern "C" int k4_side_wait_main(void* side, void* stream) {
    return k4_taped(stream, [side](void* stream) -> int { return k4_wait_stream((hipStream_t)side, (hipStream_t)stream); });
}
extern "C" int k4_main_wait_side(void* side, void* stream) {
    return k4_taped(stream, [side](void* stream) -> int { return k4_wait_stream((hipStream_t)stream, (hipStream_t)side); });
}

// ---------------------------------------------------------------------------------------------------------------------
// Touched voxels of a grid gradient [C][nvox] (any channel non-zero) as a compact int32 index list: what the data-parallel exchange of
// the joint step sends instead of the dense 1.36 GB tensor (joint_train.sparse_grad_allreduce).  One pass over the gradient, no
// temporaries; the list is in arbitrary order (wave-aggregated append), *counter receives the TOTAL number of touched voxels even when
// it exceeds `cap` (the caller retries with a larger list).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_touched_voxels(const float* __restrict__ g, int C, int64_t nvox, int32_t* __restrict__ out, int64_t cap,
                                                        unsigned long long* __restrict__ counter) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool t = false;
    if (v < nvox)
        for (int ch = 0; ch < C; ++ch) t |= g[(size_t)ch * nvox + v] != 0.f;
    const unsigned long long m = __ballot(t);
    if (m == 0) return;
    const int lane = k4_lane();
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = __shfl(base, leader);
    if (t) {
        const unsigned long long pos = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        if ((int64_t)pos < cap) out[pos] = (int32_t)v;
    }
}

extern "C" int k4_touched_voxels(const float* grad, int32_t channels, int64_t n_vox, int32_t* idx_out, int64_t cap, int64_t* counter, void* stream) {
    if (!grad || channels <= 0 || n_vox < 0 || n_vox >= (1ll << 31) || cap < 0 || (cap > 0 && !idx_out) || !counter) return K4_ERR_BAD_ARG;
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(int64_t), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (n_vox == 0) return 0;
    hipLaunchKernelGGL(k_touched_voxels, dim3((unsigned)((n_vox + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad, channels, n_vox, idx_out, cap,
                       reinterpret_cast<unsigned long long*>(counter));
    return k4_check_launch();
}
