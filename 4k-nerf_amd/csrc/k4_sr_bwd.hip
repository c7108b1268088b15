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
//          atomics (the entry point zeroes dW first; <= H/K4_WG_BAND addends per element).
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

// zero-fill as a KERNEL (not hipMemsetAsync): inside a hipGraph capture (lib/sr_train.GraphedDecoder) the memset of this ROCm build ran
// once at capture time instead of becoming a node -- replays then added their split-K partial sums to stale contents
__global__ void wg_zero_kernel(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
static inline void wg_zero(float* p, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(wg_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
}

struct WgradParams {
    const float* x; int cin; int x_stride;
    const float* gy; int cout; int gy_stride;
    int ks, H, W;
    float* dw;                                  // [cout][cin][ks][ks]
    float* db;                                  // NULL, or [cout]: the bias gradient, summed by the workgroups of tap 0 / input block 0
    int ci_blocks, co_blocks, bands;
    int units, upw, groups;                     // nine-tap kernel: (row, 16-pixel chunk) units, units per wave, workgroups per (ci block, co block)
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
    const bool do_bias = P.db != nullptr && tap == 0 && cib == 0;        // workgroup-uniform: these workgroups see every dY value of their band once
    double bsum = 0.0;                                                  // the band's dY sum in fp64: one fp32 rounding per band
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
            if (do_bias) bsum += (double)(((b8[0] + b8[1]) + (b8[2] + b8[3])) + ((b8[4] + b8[5]) + (b8[6] + b8[7])));
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
    if (do_bias) {                                                      // dbias[co] += this band's sum of dY[.][co]: both pixel halves, four waves
        bsum += __shfl_xor(bsum, 32);
        double* const redd = reinterpret_cast<double*>(&red[0][0]);
        __syncthreads();                                                // `red` is free again
        if (half == 0) redd[wv * 32 + l31] = bsum;
        __syncthreads();
        if (wv == 0 && half == 0 && co_ok) unsafeAtomicAdd(P.db + co, (float)((redd[l31] + redd[32 + l31]) + (redd[64 + l31] + redd[96 + l31])));
    }
}

// The 3x3 layers: ALL NINE TAPS in one workgroup.  In the kernel above every (tap, ci block) workgroup fetches and splits the same dY fragment and every
// (tap, co block) workgroup the same input pixels (one pixel to the side): 16 fetches and two 3-term splits (~100 vector instructions) per 6 matrix
// instructions -- the weight gradients of the 64x64 training patch ran at 37 us a layer, 85 layers an iteration, and were the longer of the two chains of the
// decoder's backward pass (profiles/r06_joint_phase_events.md, section 13).  Here a wave takes units of (row y, 16 pixels): the dY fragment is fetched and
// split ONCE for the nine taps; of each of the three input rows y - 1, y, y + 1 it fetches the 10 pixels px - 1 .. px + 8 once and splits them once as the
// pairs (0,1) (2,3) (4,5) (6,7) (8,9): pairs 0..3 are the fragment of tap dx = -1, pairs 1..4 that of dx = +1, and dx = 0 is one v_alignbit per dword --
// 38 fetches and ~280 vector instructions per 54 matrix instructions.  The nine 32 x 32 tiles of the four waves are summed in LDS in dW's own order
// ([co][ci][tap]: 288 consecutive floats per output channel) and added to dW with fp32 atomics on CONSECUTIVE addresses (the tiles of the kernel above go
// out with one cache line per lane).  Units per wave (P.upw) are chosen by the launcher: 4 on the 64 x 64 patch (96 workgroups for 192 -> 32), more on the
// decoder's 256 x 256 layers.  Same products, same 3-term splits; the order of the fp32 sums differs (as it does from run to run with the atomics).
__device__ __forceinline__ void wg_split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = wg_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = wg_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
    p2 = wg_pk_bf16(sa, sb);
}
#define K4_WG9_PITCH 73
__global__ __launch_bounds__(256, 2) void k4_conv_wgrad9_kernel(const WgradParams P) {
    __shared__ float red[4 * 32 * K4_WG9_PITCH];
    const int lane = k4_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int b = (int)blockIdx.x;
    const int grp = b % P.groups; b /= P.groups;
    const int cob = b % P.co_blocks;
    const int cib = b / P.co_blocks;
    const int ci = cib * 32 + l31, co = cob * 32 + l31;
    const bool ci_ok = ci < P.cin, co_ok = co < P.cout;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x16)(0.f);
    const bool do_bias = P.db != nullptr && cib == 0;                    // workgroup-uniform: these workgroups see every dY value of their units once
    double bsum = 0.0;
    const int xchunks = (P.W + 15) >> 4;
    for (int u = 0; u < P.upw; ++u) {
        const int unit = (grp * 4 + wv) * P.upw + u;                     // a wave's units are consecutive: neighbouring chunks of a row
        if (unit >= P.units) break;                                      // (wave-uniform)
        const int y = unit / xchunks, x0 = (unit - y * xchunks) * 16 + half * 8;
        float b8[8], a10[3][10];
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[e] = (co_ok && x0 + e < P.W) ? P.gy[((size_t)y * P.W + x0 + e) * P.gy_stride + co] : 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int sy = y + r - 1;
            const bool row_ok = ci_ok && sy >= 0 && sy < P.H;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int sx = x0 - 1 + j;
                a10[r][j] = (row_ok && sx >= 0 && sx < P.W) ? P.x[((size_t)sy * P.W + sx) * P.x_stride + ci] : 0.f;
            }
        }
        if (do_bias) bsum += (double)(((b8[0] + b8[1]) + (b8[2] + b8[3])) + ((b8[4] + b8[5]) + (b8[6] + b8[7])));
        uint4 b0, b1, b2;
        wg_split3(b8, b0, b1, b2);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            unsigned t0[5], t1[5], t2[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) wg_split_pair(a10[r][2 * k], a10[r][2 * k + 1], t0[k], t1[k], t2[k]);
#define WG_MFMA(T, A, B) acc[T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, A), __builtin_bit_cast(wg_bf16x8, B), acc[T], 0, 0, 0)
#define WG_TAP(T, A0, A1, A2) WG_MFMA(T, A2, b0); WG_MFMA(T, A0, b2); WG_MFMA(T, A1, b1); WG_MFMA(T, A1, b0); WG_MFMA(T, A0, b1); WG_MFMA(T, A0, b0)
            {
                const uint4 a0 = make_uint4(t0[0], t0[1], t0[2], t0[3]), a1 = make_uint4(t1[0], t1[1], t1[2], t1[3]), a2 = make_uint4(t2[0], t2[1], t2[2], t2[3]);
                WG_TAP(r * 3 + 0, a0, a1, a2);                             // dx = -1: input pixels px - 1 .. px + 6
            }
            {
#define WG_AL(t, k) __builtin_amdgcn_alignbit(t[k + 1], t[k], 16)
                const uint4 a0 = make_uint4(WG_AL(t0, 0), WG_AL(t0, 1), WG_AL(t0, 2), WG_AL(t0, 3)), a1 = make_uint4(WG_AL(t1, 0), WG_AL(t1, 1), WG_AL(t1, 2), WG_AL(t1, 3)),
                            a2 = make_uint4(WG_AL(t2, 0), WG_AL(t2, 1), WG_AL(t2, 2), WG_AL(t2, 3));
#undef WG_AL
                WG_TAP(r * 3 + 1, a0, a1, a2);                             // dx = 0: px .. px + 7
            }
            {
                const uint4 a0 = make_uint4(t0[1], t0[2], t0[3], t0[4]), a1 = make_uint4(t1[1], t1[2], t1[3], t1[4]), a2 = make_uint4(t2[1], t2[2], t2[3], t2[4]);
                WG_TAP(r * 3 + 2, a0, a1, a2);                             // dx = +1: px + 1 .. px + 8
            }
#undef WG_TAP
#undef WG_MFMA
        }
    }
    // The four waves' tiles summed through LDS and added to dW, eight input channels at a time: accumulator register r of lane l = D[i = (r&3)+8*(r>>2)+4*half][j = l31],
    // so the registers 4p .. 4p+3 of the nine taps are the input channels 8p .. 8p+7 -- in dW ([co][ci][tap]) 72 consecutive floats per output channel.  Plain stores
    // into the wave's own region, plain loads of the four regions (LDS float atomics and a read-modify-write per wave in turn both cost tens of microseconds).
    const int ci_left = P.cin - cib * 32;                                // input channels of this block that exist
    for (int p = 0; p < 4; ++p) {
        float* const mine = red + wv * (32 * K4_WG9_PITCH);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // (a runtime p would index the accumulator vectors dynamically: the four cases are spelled out)
            const float v0 = p == 0 ? acc[t][0] : p == 1 ? acc[t][4] : p == 2 ? acc[t][8] : acc[t][12];
            const float v1 = p == 0 ? acc[t][1] : p == 1 ? acc[t][5] : p == 2 ? acc[t][9] : acc[t][13];
            const float v2 = p == 0 ? acc[t][2] : p == 1 ? acc[t][6] : p == 2 ? acc[t][10] : acc[t][14];
            const float v3 = p == 0 ? acc[t][3] : p == 1 ? acc[t][7] : p == 2 ? acc[t][11] : acc[t][15];
            float* const d = mine + l31 * K4_WG9_PITCH + (4 * half) * 9 + t;
            d[0] = v0; d[9] = v1; d[18] = v2; d[27] = v3;
        }
        __syncthreads();
        for (int idx = (int)threadIdx.x; idx < 32 * 72; idx += 256) {
            const int j = idx / 72, q = idx - j * 72;
            const int co_ = cob * 32 + j;
            const int o = j * K4_WG9_PITCH + q;
            const float v = (red[o] + red[32 * K4_WG9_PITCH + o]) + (red[2 * 32 * K4_WG9_PITCH + o] + red[3 * 32 * K4_WG9_PITCH + o]);
            if (co_ < P.cout && p * 72 + q < ci_left * 9) unsafeAtomicAdd(&P.dw[((size_t)co_ * P.cin + cib * 32) * 9 + p * 72 + q], v);
        }
        __syncthreads();
    }
    if (do_bias) {                                                      // dbias[co] += the dY sums of this workgroup's units: both pixel halves, four waves
        bsum += __shfl_xor(bsum, 32);
        double* const redd = reinterpret_cast<double*>(&red[0]);
        __syncthreads();                                                // `red` is free again
        if (half == 0) redd[wv * 32 + l31] = bsum;
        __syncthreads();
        if (wv == 0 && half == 0 && co_ok) unsafeAtomicAdd(P.db + co, (float)((redd[l31] + redd[32 + l31]) + (redd[64 + l31] + redd[96 + l31])));
    }
}

// dbias[co] = sum over pixels of gy[p][co].  Workgroup = 32 channels x a slab of K4_BG_SLAB pixels; lanes = channel x 8 pixel phases, 8
// independent loads in flight per thread; the slab sums are added to the zeroed dbias with one fp32 atomic per channel and workgroup
// (like wgrad's split-K).  The first form -- ONE workgroup per 32 channels walking the whole image with a dependent add per load -- took
// 120 us per layer on the 64x64 training patch and was HALF of the joint iteration's GPU time (229 layers, profiles/r02_final_tree.md).
#define K4_BG_SLAB 512
__global__ __launch_bounds__(256) void k4_bias_grad_kernel(const float* __restrict__ gy, int cout, int gy_stride, int64_t n_pix, float* __restrict__ db) {
    __shared__ float part[8][32];
    const int cb = (int)blockIdx.x, c = cb * 32 + (int)(threadIdx.x & 31);
    const int ph = (int)(threadIdx.x >> 5);                              // 0..7
    const int64_t p0 = (int64_t)blockIdx.y * K4_BG_SLAB, p1 = (p0 + K4_BG_SLAB < n_pix) ? p0 + K4_BG_SLAB : n_pix;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < cout)
        for (int64_t p = p0 + ph; p < p1; p += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t q = p + 8 * u;
                if (q < p1) s[u] += gy[q * gy_stride + c];
            }
        }
    part[ph][threadIdx.x & 31] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (ph == 0 && c < cout) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][threadIdx.x & 31];
        unsafeAtomicAdd(db + c, t);
    }
}

static int wgrad_launch(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                        int32_t ksize, int32_t H, int32_t W, float* dw, float* dbias, int64_t zero_floats, void* stream) {
    if (!x || !gy || !dw || cin <= 0 || cout <= 0 || x_stride < cin || gy_stride < cout || H <= 0 || W <= 0 || (ksize != 1 && ksize != 3))
        return K4_ERR_BAD_ARG;
    WgradParams P{};
    P.x = x; P.cin = cin; P.x_stride = x_stride; P.gy = gy; P.cout = cout; P.gy_stride = gy_stride;
    P.ks = ksize; P.H = H; P.W = W; P.dw = dw; P.db = dbias;
    P.ci_blocks = (cin + 31) / 32; P.co_blocks = (cout + 31) / 32; P.bands = (H + K4_WG_BAND - 1) / K4_WG_BAND;
    if (zero_floats > 0) wg_zero(dw, zero_floats, (hipStream_t)stream);                                                 // split-K partial sums are ADDED
    if (ksize == 3 && !(k4_env().sr_debug & 4096)) {
        // units per wave: as many workgroups as ~3 per CU allow, at least 4 units (the tile reduction amortises over them)
        P.units = H * ((W + 15) / 16);
        const int want_groups = 768 / (P.ci_blocks * P.co_blocks) > 0 ? 768 / (P.ci_blocks * P.co_blocks) : 1;
        P.upw = (P.units + 4 * want_groups - 1) / (4 * want_groups);
        if (P.upw < 4) P.upw = 4;            // (64 x 64 patch, isolated launch: 24 / 29 / 39 us at 2 / 4 / 8 units; inside the iteration, beside the other stream's kernels: 8.05 / 7.88 / 8.5 ms)
        P.groups = (P.units + 4 * P.upw - 1) / (4 * P.upw);
        hipLaunchKernelGGL(k4_conv_wgrad9_kernel, dim3((unsigned)(P.ci_blocks * P.co_blocks * P.groups)), dim3(256), 0, (hipStream_t)stream, P);
        return k4_check_launch();
    }
    const unsigned grid = (unsigned)(ksize * ksize * P.ci_blocks * P.co_blocks * P.bands);
    hipLaunchKernelGGL(k4_conv_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P);
    return k4_check_launch();
}

// (the entry points of this file that the decoder's training pass calls are recordable: k4_tape.hip)
extern "C" int k4_conv2d_wgrad_bf16x6(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                                      int32_t ksize, int32_t H, int32_t W, float* dw, void* stream) {
    return k4_taped(stream, [=](void* stream) -> int {
        return wgrad_launch(x, cin, x_stride, gy, cout, gy_stride, ksize, H, W, dw, nullptr, (int64_t)cout * cin * ksize * ksize, stream);
    });
}

extern "C" int k4_conv2d_wgrad_dbias_bf16x6(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                                            int32_t ksize, int32_t H, int32_t W, float* dw_db, void* stream) {
    // dw_db = [cout*cin*k*k floats of dW | cout floats of dbias], ONE buffer: one zero-fill, one launch
    if (!dw_db) return K4_ERR_BAD_ARG;
    return k4_taped(stream, [=](void* stream) -> int {
        const int64_t nw = (int64_t)cout * cin * ksize * ksize;
        return wgrad_launch(x, cin, x_stride, gy, cout, gy_stride, ksize, H, W, dw_db, dw_db + nw, nw + cout, stream);
    });
}

// ... ADDED to dw_db (no zero-fill): the caller has zeroed it -- k4_rdb_train_bwd zeroes the five buffers of a dense block with one launch
extern "C" int k4_conv2d_wgrad_dbias_bf16x6_acc(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                                                int32_t ksize, int32_t H, int32_t W, float* dw_db, void* stream) {
    if (!dw_db) return K4_ERR_BAD_ARG;
    return k4_taped(stream, [=](void* stream) -> int {
        const int64_t nw = (int64_t)cout * cin * ksize * ksize;
        return wgrad_launch(x, cin, x_stride, gy, cout, gy_stride, ksize, H, W, dw_db, dw_db + nw, 0, stream);
    });
}
extern "C" int k4_zero_f32(float* p, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && !p) || (n + 255) / 256 > 0x7fffffffLL) return K4_ERR_BAD_ARG;
    if (n == 0) return K4_OK;
    return k4_taped(stream, [=](void* stream) -> int {
        wg_zero(p, n, (hipStream_t)stream);
        return k4_check_launch();
    });
}

extern "C" int k4_conv2d_bias_grad(const float* gy, int32_t cout, int32_t gy_stride, int64_t n_pix, float* dbias, void* stream) {
    if (!gy || !dbias || cout <= 0 || gy_stride < cout || n_pix <= 0) return K4_ERR_BAD_ARG;
    const int64_t slabs = (n_pix + K4_BG_SLAB - 1) / K4_BG_SLAB;
    if (slabs > 65535) return K4_ERR_UNSUPPORTED;
    return k4_taped(stream, [=](void* stream) -> int {
        wg_zero(dbias, cout, (hipStream_t)stream);
        hipLaunchKernelGGL(k4_bias_grad_kernel, dim3((unsigned)((cout + 31) / 32), (unsigned)slabs), dim3(256), 0, (hipStream_t)stream, gy, cout, gy_stride, n_pix, dbias);
        return k4_check_launch();
    });
}

// ------------------------------------------------------------------------------------------------------------------
// Weight packing on the device.  Under training every optimizer step changes every weight, so each iteration re-packs all 260
// convolutions of SFTNet twice (forward operand and the flipped-transposed dgrad operand): as PyTorch ops (zeros / permute / three
// bf16 casts and subtractions / stack / contiguous ...) that was ~6000 tiny launches per iteration, most of the joint step's wall
// time.  One launch per (layer, operand) here, bit-identical to the host packer (lib/sr_esrnet.py::_Packed):
//   out[ch][term][tap][g2][n][e] (8 x bf16 = one 16-byte unit per (ch, term, tap, g2, n)),  channel c = 16 ch + 8 g2 + e,
//   form 0: v = W[n][c][tap]                       the layer as stored, [cout][cin][k][k]
//   form 1: v = W[c][n][taps-1-tap]                dgrad operand: the layer (cin -> cout) seen as (cout -> cin), taps flipped
//   form 2: v = W[n % cout][c][n / cout]           3x3 layer with cout <= 3 as a 1x1 layer whose outputs are (tap, co)  (K4_W_TAPS_AS_COUT)
//   form 3: v = W[c][n % cin][taps-1 - n / cin]    form 2 of the dgrad operand (layers with cin <= 3: conv_first, CondNet.0)
//   term q of v: t0 = bf16(v), t1 = bf16(v - t0), t2 = bf16(v - t0 - t1)   (round to nearest even, exact residuals)
// ------------------------------------------------------------------------------------------------------------------
__global__ void k4_pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int cin, int taps, int form,
                                    int nch, int taps_l, int NOUT, uint4* __restrict__ out, float* __restrict__ bias_out, int n_bias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_bias) bias_out[idx] = (bias && (form == 0 || form == 2) && idx < cout) ? bias[idx] : 0.f;
    if (idx >= nch * taps_l * 2 * NOUT) return;
    const int n = idx % NOUT;
    int r = idx / NOUT;
    const int g2 = r & 1; r >>= 1;
    const int tap = r % taps_l, ch = r / taps_l;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 16 + g2 * 8 + e;
        float q = 0.f;
        if (form == 0) { if (n < cout && c < cin) q = w[((size_t)n * cin + c) * taps + tap]; }
        else if (form == 1) { if (n < cin && c < cout) q = w[((size_t)c * cin + n) * taps + (taps - 1 - tap)]; }
        else if (form == 2) { if (n < taps * cout && c < cin) q = w[((size_t)(n % cout) * cin + c) * taps + n / cout]; }
        else { if (n < taps * cin && c < cout) q = w[((size_t)c * cin + n % cin) * taps + (taps - 1 - n / cin)]; }
        v[e] = q;
    }
    uint4 t0, t1, t2;
    wg_split3(v, t0, t1, t2);
    const size_t plane = (size_t)taps_l * 2 * NOUT;                   // 16-byte units per (ch, term)
    uint4* const o = out + ((size_t)ch * 3) * plane + ((size_t)tap * 2 + g2) * NOUT + n;
    o[0] = t0; o[plane] = t1; o[2 * plane] = t2;
}

// The same packing for MANY layers in one launch: the job table travels as a kernel argument (blockIdx.y = job; workgroups past a
// job's size leave at once).  Under training all ~170 operands of SFTNet (every convolution, forward and dgrad form) are re-packed
// once per iteration: 170 launches of 5 us -> 3.
struct PackJobDev { const float* w; const float* bias; uint4* out; float* bias_out; int cout, cin, taps, form, nch, taps_l, NOUT, n_bias; };
struct PackMultiArgs { PackJobDev j[K4_PACK_MULTI_MAX]; };

__global__ __launch_bounds__(256) void k4_pack_conv_multi_kernel(const PackMultiArgs A) {
    const PackJobDev& J = A.j[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = J.nch * J.taps_l * 2 * J.NOUT;
    if (idx >= total && idx >= J.n_bias) return;
    if (idx < J.n_bias) J.bias_out[idx] = (J.bias && (J.form == 0 || J.form == 2) && idx < J.cout) ? J.bias[idx] : 0.f;
    if (idx >= total) return;
    const int cout = J.cout, cin = J.cin, taps = J.taps, form = J.form, taps_l = J.taps_l, NOUT = J.NOUT;
    const float* __restrict__ w = J.w;
    const int n = idx % NOUT;
    int r = idx / NOUT;
    const int g2 = r & 1; r >>= 1;
    const int tap = r % taps_l, ch = r / taps_l;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 16 + g2 * 8 + e;
        float q = 0.f;
        if (form == 0) { if (n < cout && c < cin) q = w[((size_t)n * cin + c) * taps + tap]; }
        else if (form == 1) { if (n < cin && c < cout) q = w[((size_t)c * cin + n) * taps + (taps - 1 - tap)]; }
        else if (form == 2) { if (n < taps * cout && c < cin) q = w[((size_t)(n % cout) * cin + c) * taps + n / cout]; }
        else { if (n < taps * cin && c < cout) q = w[((size_t)c * cin + n % cin) * taps + (taps - 1 - n / cin)]; }
        v[e] = q;
    }
    uint4 t0, t1, t2;
    wg_split3(v, t0, t1, t2);
    const size_t plane = (size_t)taps_l * 2 * NOUT;
    uint4* const o = J.out + ((size_t)ch * 3) * plane + ((size_t)tap * 2 + g2) * NOUT + n;
    o[0] = t0; o[plane] = t1; o[2 * plane] = t2;
}

extern "C" int k4_pack_conv_weight_bf16x6_multi(const k4_pack_job* jobs, int32_t n_jobs, void* stream) {
    if (!jobs || n_jobs < 0) return K4_ERR_BAD_ARG;
    const std::vector<k4_pack_job> table(jobs, jobs + n_jobs);            // the closure owns a copy of the caller's host array
    return k4_taped(stream, [table, n_jobs](void* stream) -> int {
    const k4_pack_job* jobs = table.data();
    for (int base = 0; base < n_jobs; base += K4_PACK_MULTI_MAX) {
        const int nj = n_jobs - base < K4_PACK_MULTI_MAX ? n_jobs - base : K4_PACK_MULTI_MAX;
        PackMultiArgs A{};
        int max_threads = 0;
        for (int q = 0; q < nj; ++q) {
            const k4_pack_job& S = jobs[base + q];
            const int cout = S.cout, cin = S.cin, ksize = S.ksize, form = S.form;
            if (!S.w || !S.w_split || !S.bias_out || cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3) || form < 0 || form > 3) return K4_ERR_BAD_ARG;
            if ((form == 2 && (ksize != 3 || cout > 3)) || (form == 3 && (ksize != 3 || cin > 3))) return K4_ERR_BAD_ARG;
            const int taps = ksize * ksize;
            const int cout_l = form == 0 ? cout : form == 1 ? cin : form == 2 ? taps * cout : taps * cin;
            const int cin_l = (form == 1 || form == 3) ? cout : cin;
            PackJobDev& D = A.j[q];
            D.w = S.w; D.bias = S.bias; D.out = reinterpret_cast<uint4*>(S.w_split); D.bias_out = S.bias_out;
            D.cout = cout; D.cin = cin; D.taps = taps; D.form = form;
            D.taps_l = form >= 2 ? 1 : taps;
            D.NOUT = (cout_l + 31) / 32 * 32; D.nch = (cin_l + 15) / 16;
            D.n_bias = form >= 2 ? 32 : D.NOUT;
            const int total = D.nch * D.taps_l * 2 * D.NOUT;
            const int threads = total > D.n_bias ? total : D.n_bias;
            if (threads > max_threads) max_threads = threads;
        }
        hipLaunchKernelGGL(k4_pack_conv_multi_kernel, dim3((unsigned)((max_threads + 255) / 256), (unsigned)nj), dim3(256), 0, (hipStream_t)stream, A);
    }
    return k4_check_launch();
    });
}

extern "C" int k4_pack_conv_weight_bf16x6(const float* w, const float* bias, int32_t cout, int32_t cin, int32_t ksize, int32_t form,
                                          void* w_split, float* bias_out, void* stream) {
    if (!w || !w_split || !bias_out || cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3) || form < 0 || form > 3) return K4_ERR_BAD_ARG;
    if ((form == 2 && (ksize != 3 || cout > 3)) || (form == 3 && (ksize != 3 || cin > 3))) return K4_ERR_BAD_ARG;
    const int taps = ksize * ksize;
    const int cout_l = form == 0 ? cout : form == 1 ? cin : form == 2 ? taps * cout : taps * cin;     // the logical layer the kernels see
    const int cin_l = (form == 1 || form == 3) ? cout : cin;
    const int taps_l = form >= 2 ? 1 : taps;
    const int NOUT = (cout_l + 31) / 32 * 32, nch = (cin_l + 15) / 16;
    const int n_bias = form >= 2 ? 32 : NOUT;                                   // taps forms: the bias of the 3x3 layer itself (<= 3 channels)
    const int total = nch * taps_l * 2 * NOUT;
    const int threads = total > n_bias ? total : n_bias;
    return k4_taped(stream, [=](void* stream) -> int {
        hipLaunchKernelGGL(k4_pack_conv_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, bias, cout, cin, taps, form, nch, taps_l, NOUT,
                           reinterpret_cast<uint4*>(w_split), bias_out, n_bias);
        return k4_check_launch();
    });
}
