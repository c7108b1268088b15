// Backward of the VC-Decoder convolutions on the gfx950 matrix cores (the joint training loop of
// /root/reference/run_sr.py:869-1014 back-propagates the L1 / perceptual loss through SFTNet, lib/sr_esrnet.py:112-182,446-465).
//
//   dgrad  dX[p][ci] = sum_{co,t} dY[p - off(t)][co] * W[co][ci][t]  is itself a stride-1 "same" convolution of dY with the
//          flipped, transposed filter W'[ci][co][2-dy][2-dx] = W[co][ci][dy][dx]: it runs on the FORWARD kernels of k4_sr.hip
//          (k4_conv2d_nhwc_bf16x6) with host-packed W' -- no separate kernel.
//   wgrad  dW[co][ci][t] = sum_p dY[p][co] * X[p + off(t)][ci]   -- this file.  A GEMM whose K dimension is the PIXELS:
//          D[ci 32][co 32] per (tap, ci block, co block) on v_mfma_f32_32x32x16_bf16 with the exact 3-term bf16 split of both
//          operands (6 partial products, fp32 accumulation: fp32-equivalent, as the forward pass).  A fragment = 8 consecutive
//          pixels of a row for one input channel (lanes = channels: 128-byte coalesced reads per pixel), B fragment = the same 8
//          pixels of dY for one output channel.  The image is cut in bands of K4_WG_BAND rows (split-K): one workgroup per
//          (tap, ci block, co block, band), its 4 waves take rows round-robin, are reduced through LDS and added to dW with fp32
//          atomics (dW must be zero-initialised by the caller; <= H/K4_WG_BAND addends per element).
//   dbias  = sum_p dY[p][co]: k4_conv2d_bias_grad (one workgroup per 32 channels, wave shuffles + LDS).
#include "k4_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));

#define K4_WG_BAND 16

__device__ __forceinline__ unsigned wg_pk_bf16(float lo, float hi) {               // v_cvt_pk_bf16_f32 (RNE)
    const wg_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wg_bf16x2));
}
__device__ __forceinline__ void wg_split3(const float (&v)[8], uint4& t0, uint4& t1, uint4& t2) {
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        p0[i] = wg_pk_bf16(a, b);
        const float ra = a - __uint_as_float(p0[i] << 16), rb = b - __uint_as_float(p0[i] & 0xffff0000u);
        p1[i] = wg_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(p1[i] << 16), sb = rb - __uint_as_float(p1[i] & 0xffff0000u);
        p2[i] = wg_pk_bf16(sa, sb);
    }
    t0 = make_uint4(p0[0], p0[1], p0[2], p0[3]); t1 = make_uint4(p1[0], p1[1], p1[2], p1[3]); t2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
}

struct WgradParams {
    const float* x; int cin; int x_stride;
    const float* gy; int cout; int gy_stride;
    int ks, H, W;
    float* dw;                                  // [cout][cin][ks][ks]
    int ci_blocks, co_blocks, bands;
};

__global__ __launch_bounds__(256) void k4_conv_wgrad_kernel(const WgradParams P) {
    __shared__ float red[3][32 * 32];
    const int lane = k4_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int b = (int)blockIdx.x;
    const int band = b % P.bands; b /= P.bands;
    const int cob = b % P.co_blocks; b /= P.co_blocks;
    const int cib = b % P.ci_blocks; b /= P.ci_blocks;
    const int tap = b;                                                  // 0 .. ks*ks-1
    const int pad = P.ks / 2;
    const int dy = tap / P.ks - pad, dx = tap % P.ks - pad;
    const int ci = cib * 32 + l31, co = cob * 32 + l31;
    const bool ci_ok = ci < P.cin, co_ok = co < P.cout;
    f32x16 acc = (f32x16)(0.f);
    const int y_end = min((band + 1) * K4_WG_BAND, P.H);
    for (int y = band * K4_WG_BAND + wv; y < y_end; y += 4) {
        const int sy = y + dy;
        const bool row_ok = sy >= 0 && sy < P.H;
        for (int x0 = 0; x0 < P.W; x0 += 16) {
            float a8[8], b8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int px = x0 + half * 8 + e;                        // output pixel column of this k element
                const int sx = px + dx;
                const bool a_ok = ci_ok && row_ok && px < P.W && sx >= 0 && sx < P.W;
                const bool b_ok = co_ok && px < P.W;
                a8[e] = a_ok ? P.x[((size_t)sy * P.W + sx) * P.x_stride + ci] : 0.f;
                b8[e] = b_ok ? P.gy[((size_t)y * P.W + px) * P.gy_stride + co] : 0.f;
            }
            uint4 a0, a1, a2, b0, b1, b2;
            wg_split3(a8, a0, a1, a2);
            wg_split3(b8, b0, b1, b2);
#define WG_MFMA(A, B) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, A), __builtin_bit_cast(wg_bf16x8, B), acc, 0, 0, 0)
            WG_MFMA(a2, b0); WG_MFMA(a0, b2); WG_MFMA(a1, b1); WG_MFMA(a1, b0); WG_MFMA(a0, b1); WG_MFMA(a0, b0);
#undef WG_MFMA
        }
    }
    // reduce the 4 waves' tiles through LDS: accumulator register r of lane l = D[i = (r&3)+8*(r>>2)+4*half][j = l31]
    if (wv > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv - 1][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float v = acc[r] + red[0][i * 32 + l31] + red[1][i * 32 + l31] + red[2][i * 32 + l31];
            const int cii = cib * 32 + i;
            if (cii < P.cin && co_ok) atomicAdd(&P.dw[(((size_t)co * P.cin + cii) * P.ks + (dy + pad)) * P.ks + (dx + pad)], v);
        }
    }
}

// dbias[co] = sum over pixels of gy[p][co]: one workgroup per 32 channels, lanes = channel x 8 pixel phases
__global__ __launch_bounds__(256) void k4_bias_grad_kernel(const float* __restrict__ gy, int cout, int gy_stride, int64_t n_pix, float* __restrict__ db) {
    __shared__ float part[8][32];
    const int c = (int)blockIdx.x * 32 + (int)(threadIdx.x & 31);
    const int ph = (int)(threadIdx.x >> 5);                              // 0..7
    float s = 0.f;
    if (c < cout)
        for (int64_t p = ph; p < n_pix; p += 8) s += gy[p * gy_stride + c];
    part[ph][threadIdx.x & 31] = s;
    __syncthreads();
    if (ph == 0 && c < cout) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][threadIdx.x & 31];
        db[c] = t;
    }
}

extern "C" int k4_conv2d_wgrad_bf16x6(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                                      int32_t ksize, int32_t H, int32_t W, float* dw, void* stream) {
    if (!x || !gy || !dw || cin <= 0 || cout <= 0 || x_stride < cin || gy_stride < cout || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3))
        return K4_ERR_BAD_ARG;
    WgradParams P{};
    P.x = x; P.cin = cin; P.x_stride = x_stride; P.gy = gy; P.cout = cout; P.gy_stride = gy_stride;
    P.ks = ksize; P.H = H; P.W = W; P.dw = dw;
    P.ci_blocks = (cin + 31) / 32; P.co_blocks = (cout + 31) / 32; P.bands = (H + K4_WG_BAND - 1) / K4_WG_BAND;
    const unsigned grid = (unsigned)(ksize * ksize * P.ci_blocks * P.co_blocks * P.bands);
    hipLaunchKernelGGL(k4_conv_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
    return k4_check_launch();
}

extern "C" int k4_conv2d_bias_grad(const float* gy, int32_t cout, int32_t gy_stride, int64_t n_pix, float* dbias, void* stream) {
    if (!gy || !dbias || cout <= 0 || gy_stride < cout || n_pix <= 0) return K4_ERR_BAD_ARG;
    hipLaunchKernelGGL(k4_bias_grad_kernel, dim3((unsigned)((cout + 31) / 32)), dim3(256), 0, (hipStream_t)stream, gy, cout, gy_stride, n_pix, dbias);
    return k4_check_launch();
}
