// VC-Decoder (SFTNet, lib/sr_esrnet.py) convolutions for gfx950: NHWC fp32 implicit GEMM on the matrix cores.
//
// One kernel template covers every layer of SFTNet.forward (lib/sr_esrnet.py:446-465):
//   3x3 / 1x1, stride 1, zero "same" padding; C_in 1..192 read from a channel slice of an NHWC buffer
//   (the dense block's torch.cat((x, x1, ..), 1) of lib/sr_esrnet.py:152-156 is a [H][W][192] buffer whose
//   slices the convs write in place -- no concat copies); fused epilogues: bias, LeakyReLU(0.2), residual
//   (y*s + res: x5*0.2+x :158, out*0.2+x :182, body_feat += feat :458), SFT modulation
//   (x*(scale+1)+shift :123) and nearest x2 upsampling folded into the loader (F.interpolate :461-463).
//
// GEMM view per workgroup: M = 256 output pixels (8 rows x 32 columns), N = C_out (32*NT), K = taps*C_in,
// walked in chunks of KC = 8 input channels.  v_mfma_f32_32x32x2_f32 (exact fp32: the result is a k-ordered
// fmaf chain): A = 32 pixels of one image row x 2 channels, B = 2 channels x 32 output channels.
//   * the chunk's haloed input tile sits in LDS channel-major [KC][10][36]: the A operand of tap (dy,dx) is
//     32 consecutive floats of one row -> one conflict-free ds_read_b32, shifted by the tap;
//   * the chunk's weights sit in LDS as [tap][KC][32*NT]: the B operand is 32 consecutive floats;
//   * a wave owns 2 rows x NT column blocks: 2*NT accumulators of 16 VGPRs; every operand read feeds 2 MFMAs.
// fp32 MFMA issues at the fp32 vector rate (157 TFLOP/s peak), so LDS/HBM are far from limiting: the kernel is
// matrix-pipe bound; bf16 would be 16x faster but breaks the fp32 parity contract.
#include "k4_common.h"
#include "k4_p16.h"

// a * b + c with TWO roundings, as the reference's separate PyTorch ops (`x5 * 0.2 + x`, `x * (scale + 1) + shift`, lib/sr_esrnet.py:123 /
// :158 / :182) -- hipcc would contract the expression into one FMA.  One rounding less is not "wrong", but a LeakyReLU input within an ulp
// of zero downstream then takes the other branch than the reference's, and the training gradients differ visibly at that pixel
// (tests/test_sr_train_gpu.py; tools/sr_rdb_debug4.py finds such elements).
__device__ __forceinline__ float k4s_mul_add(float a, float b, float c) { return __fadd_rn(__fmul_rn(a, b), c); }
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KC 8
#define TILE_W 32
#define TILE_H 8

struct ConvParams {
    const float* x; int cin; int cin_stride;
    const float* w; const float* bias;
    float* y; int cout; int cout_stride;
    int H, W;            // output size
    int srcH, srcW;      // input size (H/2, W/2 with K4_PRE_UPSAMPLE2X)
    unsigned flags; float slope;
    const float* res; int res_stride; float res_scale;
    const float* modx; int mod_stride;
    int tiles_x, tiles_y;
    int debug;           // K4_SR_DEBUG ablation bits of the v2 3x3 kernel (profiling only, WRONG results; 0 in production): 2 = no epilogue stores,
                         // 4 = no MFMA phase, 8 = no split / LDS stores after the first chunk, 16 = no activation loads after the first chunk
};

// Grouped launch: one grid covers up to K4_MAX_JOBS windows (the tiles of SFTNet.tile_process are independent images that share every
// weight): a layer of all windows is ONE launch, its workgroups fill the chip together (a 520x520 window alone is 561 workgroups on
// 256 CUs = 73 % tail efficiency; the four windows of a frame together are 1649 = 92 %) and the launch count per frame drops 4x.
struct ConvMulti {
    ConvParams base;                       // everything the windows share (x / y / res / modx / H / W / srcH / srcW / tiles_* are per job)
    int n;
    int blk_end[K4_MAX_JOBS];              // exclusive prefix of the jobs' workgroup counts
    const float* x[K4_MAX_JOBS]; float* y[K4_MAX_JOBS]; const float* res[K4_MAX_JOBS]; const float* modx[K4_MAX_JOBS];
    int H[K4_MAX_JOBS], W[K4_MAX_JOBS], tiles_x[K4_MAX_JOBS];
    int total;                             // workgroup-tiles of the launch
};
// -> this workgroup's window parameters and its tile index inside that window (workgroup-uniform)
__device__ __forceinline__ ConvParams k4_select_job_of(const ConvMulti& M, int b, int& local);
__device__ __forceinline__ ConvParams k4_select_job(const ConvMulti& M, int& local) {
    return k4_select_job_of(M, k4_xcd_remap((int)blockIdx.x, (int)gridDim.x), local);
}
__device__ __forceinline__ ConvParams k4_select_job_of(const ConvMulti& M, int b, int& local) {
    int g = 0;
    while (g + 1 < M.n && b >= M.blk_end[g]) ++g;
    local = b - (g ? M.blk_end[g - 1] : 0);
    ConvParams P = M.base;
    P.x = M.x[g]; P.y = M.y[g]; P.res = M.res[g]; P.modx = M.modx[g];
    P.H = M.H[g]; P.W = M.W[g]; P.tiles_x = M.tiles_x[g];
    const bool ups = (P.flags & K4_PRE_UPSAMPLE2X) != 0;
    P.srcH = ups ? P.H / 2 : P.H; P.srcW = ups ? P.W / 2 : P.W;
    return P;
}

// shared epilogue: lane holds output channel n*32+l31 of pixels x0 + row(r,half), rows y0 + wv*2 + m
template <int NT>
__device__ __forceinline__ void k4_conv_epilogue(const ConvParams& P, f32x16 (&acc)[2][NT], int x0, int y0, int wv, int half, int l31) {
    const bool modulate = (P.flags & K4_EPI_MODULATE) != 0;
    constexpr int NW = NT;                    // N tiles written (MODULATE: only the first NT/2)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int gy = y0 + wv * 2 + m;
        if (gy >= P.H) continue;
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            if (modulate && n >= NT / 2) continue;
            const int co = n * 32 + l31;
            if (co >= P.cout) continue;
            const float bsc = P.bias[co];
            const float bsh = modulate ? P.bias[co + (NT / 2) * 32] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gx = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (gx >= P.W) continue;
                const size_t pix = (size_t)gy * P.W + gx;
                float v = acc[m][n][r] + bsc;
                if (modulate) {
                    // SFTLayer: x * (scale + 1) + shift      (lib/sr_esrnet.py:123)
                    const float sh = acc[m][(n + NT / 2) % NT][r] + bsh;
                    v = k4s_mul_add(P.modx[pix * P.mod_stride + co], v + 1.f, sh);
                }
                if (P.flags & K4_EPI_LRELU) v = v > 0.f ? v : v * P.slope;
                if (P.flags & K4_EPI_RES) v = k4s_mul_add(v, P.res_scale, P.res[pix * P.res_stride + co]);
                P.y[pix * P.cout_stride + co] = v;
            }
        }
    }
}

template <int KS, int NT>
__global__ __launch_bounds__(256) void k4_conv_kernel(const ConvParams P) {
    constexpr int TAPS = KS * KS;
    constexpr int PADW = KS / 2;
    constexpr int ROWS = TILE_H + 2 * PADW;
    constexpr int COLS_USED = TILE_W + 2 * PADW;
    constexpr int COLS = KS == 3 ? 36 : 32;
    constexpr int NOUT = NT * 32;
    __shared__ float in_s[KC][ROWS][COLS];
    __shared__ float w_s[TAPS][KC][NOUT];

    const int lane = k4_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int tile = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int x0 = tx * TILE_W, y0 = ty * TILE_H;
    const bool ups = (P.flags & K4_PRE_UPSAMPLE2X) != 0;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x16)(0.f);

    const int nchunks = (P.cin + KC - 1) / KC;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * KC;
        // ---- stage the haloed input tile: NHWC global -> channel-major LDS ----
        for (int p = (int)threadIdx.x; p < ROWS * COLS_USED; p += 256) {
            const int py = p / COLS_USED, px = p - py * COLS_USED;
            const int gy = y0 - PADW + py, gx = x0 - PADW + px;
            float v[KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) v[c] = 0.f;
            if (gy >= 0 && gy < P.H && gx >= 0 && gx < P.W) {
                const int sy = ups ? (gy >> 1) : gy, sx = ups ? (gx >> 1) : gx;
                const float* src = P.x + ((size_t)sy * P.srcW + sx) * P.cin_stride + c0;
                if (c0 + KC <= P.cin && ((((size_t)src) & 15) == 0)) {
                    const float4 a = *reinterpret_cast<const float4*>(src);
                    const float4 b = *reinterpret_cast<const float4*>(src + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c) if (c0 + c < P.cin) v[c] = src[c];
                }
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) in_s[c][py][px] = v[c];
        }
        // ---- stage this chunk's weights (already in LDS order: [chunk][tap][KC][NOUT]) ----
        {
            const float4* src = reinterpret_cast<const float4*>(P.w + (size_t)ch * TAPS * KC * NOUT);
            float4* dst = reinterpret_cast<float4*>(&w_s[0][0][0]);
            for (int i = (int)threadIdx.x; i < TAPS * KC * NOUT / 4; i += 256) dst[i] = src[i];
        }
        __syncthreads();
        // ---- 2 rows x NT column blocks per wave ----
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int dy = t / KS, dx = t - dy * KS;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                const float a0 = in_s[2 * kk + half][wv * 2 + 0 + dy][l31 + dx];
                const float a1 = in_s[2 * kk + half][wv * 2 + 1 + dy][l31 + dx];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float b = w_s[t][2 * kk + half][n * 32 + l31];
                    acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    k4_conv_epilogue<NT>(P, acc, x0, y0, wv, half, l31);
}

// ------------------------------------------------------------------------------------------------------------------
// Split-bf16 arithmetic on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16): K is walked in chunks of KC2 = 16 input channels
// (one MFMA K-step per tap); activations stay fp32 in HBM, the split happens while a chunk is staged into LDS.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define KC2 16

template <int KS, int NT>
static int launch_conv(const ConvParams& P, hipStream_t st) {
    const dim3 grid((unsigned)(P.tiles_x * P.tiles_y)), block(256);
    hipLaunchKernelGGL((k4_conv_kernel<KS, NT>), grid, block, 0, st, P);
    return k4_check_launch();
}

extern "C" int64_t k4_conv_weight_floats(int32_t cout, int32_t cin, int32_t ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return -1;
    const int64_t nt = (cout + 31) / 32;
    if (nt != 1 && nt != 2 && nt != 4) return -1;
    return (int64_t)((cin + KC - 1) / KC) * ksize * ksize * KC * nt * 32;
}

extern "C" int k4_conv2d_nhwc(const float* x, int32_t cin, int32_t cin_stride,
                              const float* w_packed, const float* bias, int32_t ksize,
                              float* y, int32_t cout, int32_t cout_stride,
                              int32_t H, int32_t W, uint32_t flags, float slope,
                              const float* res, int32_t res_stride, float res_scale,
                              const float* mod_x, int32_t mod_stride, void* stream) {
    if (!x || !w_packed || !bias || !y || cin <= 0 || cout <= 0 || H <= 0 || W <= 0) return K4_ERR_BAD_ARG;
    if (cin_stride < cin || (ksize != 1 && ksize != 3)) return K4_ERR_BAD_ARG;
    if ((flags & K4_EPI_RES) && (!res || res_stride <= 0)) return K4_ERR_BAD_ARG;
    const bool modulate = (flags & K4_EPI_MODULATE) != 0;
    if (modulate && (!mod_x || mod_stride <= 0 || cout % 32 != 0)) return K4_ERR_BAD_ARG;
    if ((flags & K4_PRE_UPSAMPLE2X) && ((H & 1) || (W & 1))) return K4_ERR_BAD_ARG;
    // MODULATE: the GEMM produces 2*cout channels ([scale | shift]); cout is what gets written
    const int gemm_n = modulate ? 2 * cout : cout;
    const int nt = (gemm_n + 31) / 32;
    if (cout_stride < cout) return K4_ERR_BAD_ARG;
    ConvParams P{};
    P.x = x; P.cin = cin; P.cin_stride = cin_stride; P.w = w_packed; P.bias = bias;
    P.y = y; P.cout = cout; P.cout_stride = cout_stride; P.H = H; P.W = W;
    P.srcH = (flags & K4_PRE_UPSAMPLE2X) ? H / 2 : H; P.srcW = (flags & K4_PRE_UPSAMPLE2X) ? W / 2 : W;
    P.flags = flags; P.slope = slope; P.res = res; P.res_stride = res_stride; P.res_scale = res_scale;
    P.modx = mod_x; P.mod_stride = mod_stride;
    P.tiles_x = (W + TILE_W - 1) / TILE_W; P.tiles_y = (H + TILE_H - 1) / TILE_H;
    hipStream_t st = (hipStream_t)stream;
    if (ksize == 3) {
        if (nt == 1) return launch_conv<3, 1>(P, st);
        if (nt == 2) return launch_conv<3, 2>(P, st);
    } else {
        if (nt == 1) return launch_conv<1, 1>(P, st);
        if (nt == 2) return launch_conv<1, 2>(P, st);
        if (nt == 4) return launch_conv<1, 4>(P, st);
    }
    return K4_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------------
// 3-term split ("bf16x6", the default decoder arithmetic): fp32-EQUIVALENT convolutions on the bf16 matrix pipe.
// Every fp32 operand is split exactly into three bf16 terms v = v0 + v1 + v2 (8+8+8 significant bits, RNE of the running
// remainder, v_cvt_pk_bf16_f32); a product is accumulated as x0w0 + x0w1 + x1w0 + x1w1 + x0w2 + x2w0 (smallest first) in
// the fp32 accumulator.  The three dropped partial products are <= 2^-23 |x w| -- one fp32 rounding of the product -- so
// the result differs from the fp32-input MFMA kernel above the way one fp32 summation order differs from another, while
// 6 x v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace 8 x v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 2.67x less
// matrix-pipe time.  Workgroup = 8 waves, tile 16 rows x 32 columns (one workgroup per CU: 114 KB of LDS for the split
// 3x3 chunk); a wave owns 2 rows x NT column blocks.  The NEXT chunk's activations and pre-split weights are fetched into
// registers before the current chunk's MFMAs and split/stored after them, so HBM/L2 latency hides behind the matrix pipe.
// ------------------------------------------------------------------------------------------------------------------
#define TILE_HB 16
typedef __bf16 k4s_bf16x2 __attribute__((ext_vector_type(2)));
typedef float k4s_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned k4s_pk_bf16(float lo, float hi) {              // v_cvt_pk_bf16_f32 (RNE)
    const k4s_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, k4s_bf16x2));
}
__device__ __forceinline__ void k4s_split3(const float (&v)[8], uint4& t0, uint4& t1, uint4& t2) {
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        p0[i] = k4s_pk_bf16(a, b);
        const float ra = a - __uint_as_float(p0[i] << 16), rb = b - __uint_as_float(p0[i] & 0xffff0000u);
        p1[i] = k4s_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(p1[i] << 16), sb = rb - __uint_as_float(p1[i] & 0xffff0000u);
        p2[i] = k4s_pk_bf16(sa, sb);
    }
    t0 = make_uint4(p0[0], p0[1], p0[2], p0[3]); t1 = make_uint4(p1[0], p1[1], p1[2], p1[3]); t2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 k4s_f16x2 __attribute__((ext_vector_type(2)));
// 4 channels, scaled by the power of two `sc` -> fp16 hi (RNE) and fp16 lo = RNE(x*sc - hi): x*sc == hi + lo up to 2^-22 relative
// hi = RNE_fp16(v * sc), lo = RNE_fp16(v * sc - hi) for four values, sc a power of two: 8 x v_fma_mix{lo,hi}_f16 (the fp32 FMA is exact here,
// one rounding to fp16 each; the high term adds -0.0 so that a negative zero keeps its sign).  The plain C++ form compiles to 16
// instructions (hipcc does not form fma_mix); the staging phase they sit in runs between two barriers, outside the matrix phase, so every
// instruction there is exposed (profiles/r04_mfma_valu_overlap.md): bit-identical results (tools/micro/fma_mix_split.hip).
#ifndef K4S_ASM_SPLIT
#define K4S_ASM_SPLIT 1
#endif
#ifndef K4S_INT_MAX
#define K4S_INT_MAX 1
#endif
__device__ __forceinline__ void k4s_split2h_x4(const float4& v, float sc, uint2& hi, uint2& lo) {
#if !K4S_ASM_SPLIT
    const float f[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    unsigned ph[2], pl[2];
    for (int i = 0; i < 2; ++i) {
        const k4s_f32x2 x = {f[2 * i], f[2 * i + 1]};
        const k4s_f16x2 h = __builtin_convertvector(x, k4s_f16x2);
        const k4s_f32x2 r = x - __builtin_convertvector(h, k4s_f32x2);
        const k4s_f16x2 l = __builtin_convertvector(r, k4s_f16x2);
        ph[i] = __builtin_bit_cast(unsigned, h); pl[i] = __builtin_bit_cast(unsigned, l);
    }
    hi = make_uint2(ph[0], ph[1]); lo = make_uint2(pl[0], pl[1]);
    return;
#endif
    unsigned h0, h1, l0, l1;
    const float nz = -0.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(h0) : "v"(v.x), "v"(sc), "v"(nz));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(h0) : "v"(v.y), "v"(sc), "v"(nz));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(h1) : "v"(v.z), "v"(sc), "v"(nz));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(h1) : "v"(v.w), "v"(sc), "v"(nz));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "v"(sc), "v"(h0));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "v"(sc), "v"(h0));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "v"(sc), "v"(h1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "v"(sc), "v"(h1));
    hi = make_uint2(h0, h1); lo = make_uint2(l0, l1);
}

__device__ __forceinline__ void k4s_split3x4(const float4& v, uint2& t0, uint2& t1, uint2& t2) {   // 4 channels -> 3 terms x 4 bf16
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned p0[2], p1[2], p2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = f[2 * i], b = f[2 * i + 1];
        p0[i] = k4s_pk_bf16(a, b);
        const float ra = a - __uint_as_float(p0[i] << 16), rb = b - __uint_as_float(p0[i] & 0xffff0000u);
        p1[i] = k4s_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(p1[i] << 16), sb = rb - __uint_as_float(p1[i] & 0xffff0000u);
        p2[i] = k4s_pk_bf16(sa, sb);
    }
    t0 = make_uint2(p0[0], p0[1]); t1 = make_uint2(p1[0], p1[1]); t2 = make_uint2(p2[0], p2[1]);
}

// NW = waves per workgroup (tile = 2*NW rows x 32 columns).  NW = 8 for 64 output channels (one 114 KB workgroup per CU);
// NW = 4 for 32 output channels: 60 KB, two workgroups per CU whose staging / MFMA phases interleave.
template <int KS, int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k4_conv_b6_kernel(const ConvMulti M) {
    int tile;
    const ConvParams P = k4_select_job(M, tile);
    constexpr int THREADS = 64 * NW;
    constexpr int TILE_ROWS = 2 * NW;
    constexpr int TAPS = KS * KS;
    constexpr int PADW = KS / 2;
    constexpr int ROWS = TILE_ROWS + 2 * PADW;
    constexpr int COLS = TILE_W + 2 * PADW;
    constexpr int NOUT = NT * 32;
    constexpr int IN_ITEMS = ROWS * COLS * 2;                 // (pixel, channel group of 8)
    constexpr int IN_PER = (IN_ITEMS + THREADS - 1) / THREADS;
    constexpr int W_ITEMS = 3 * TAPS * 2 * NOUT;              // 16-byte units of one chunk's split weights
    constexpr int W_PER = (W_ITEMS + THREADS - 1) / THREADS;
    constexpr int IN_PLANE = 2 * ROWS * COLS;                 // uint4 per term
    extern __shared__ uint4 k4_b6_smem[];
    uint4* const in_s = k4_b6_smem;                           // [term][channel group][row][col] x 8 bf16
    uint4* const w_s = k4_b6_smem + 3 * IN_PLANE;             // [term][tap][channel group][cout] x 8 bf16

    const int tid = (int)threadIdx.x;
    const int lane = k4_lane();
    const int wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int x0 = tx * TILE_W, y0 = ty * TILE_ROWS;
    const bool ups = (P.flags & K4_PRE_UPSAMPLE2X) != 0;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x16)(0.f);

    // per-thread source of each staged item (chunk independent part)
    const float* isrc[IN_PER];
    bool iin[IN_PER];
    int ikg[IN_PER], idst[IN_PER];
#pragma unroll
    for (int i = 0; i < IN_PER; ++i) {
        const int it = tid + i * THREADS;
        const int itc = it < IN_ITEMS ? it : 0;
        const int kg = itc & 1, pp = itc >> 1;
        const int py = pp / COLS, px = pp - py * COLS;
        const int gy = y0 - PADW + py, gx = x0 - PADW + px;
        const bool inside = it < IN_ITEMS && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        const int sy = inside ? (ups ? (gy >> 1) : gy) : 0, sx = inside ? (ups ? (gx >> 1) : gx) : 0;
        isrc[i] = P.x + ((size_t)sy * P.srcW + sx) * P.cin_stride + kg * 8;
        iin[i] = inside; ikg[i] = kg;
        idst[i] = it < IN_ITEMS ? (kg * ROWS + py) * COLS + px : -1;
    }
    const int nchunks = (P.cin + KC2 - 1) / KC2;
    const uint4* const wsrc_all = reinterpret_cast<const uint4*>(P.w);

    // staged items live in registers between the fetch (before the MFMAs) and the split/store (after them); written
    // as macros, not lambdas, so that the arrays are promoted to registers (captured arrays ended up in scratch)
    float4 rva[IN_PER], rvb[IN_PER];
    uint4 wr[W_PER];
    // all pixels of the tensor share their alignment when the channel stride is a multiple of 4 floats: the vector path
    // is then a WORKGROUP-uniform decision per chunk (no divergent branches; out-of-image items fetch a valid dummy
    // address and are zeroed by a select)
    const bool vec_ok = (P.cin_stride & 3) == 0 && (((size_t)P.x) & 15) == 0;
#define K4_B6_LOAD(CH) do { \
        const int c0_ = (CH) * KC2; \
        if (vec_ok && c0_ + KC2 <= P.cin) { \
            _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) { \
                const float* src = iin[i] ? isrc[i] + c0_ : P.x; \
                const float4 va = *reinterpret_cast<const float4*>(src); \
                const float4 vb = *reinterpret_cast<const float4*>(src + 4); \
                rva[i] = iin[i] ? va : make_float4(0.f, 0.f, 0.f, 0.f); \
                rvb[i] = iin[i] ? vb : make_float4(0.f, 0.f, 0.f, 0.f); \
            } \
        } else { \
            _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) { \
                const int cb = c0_ + ikg[i] * 8; \
                const float* src = isrc[i] + c0_; \
                float e8[8]; \
                _Pragma("unroll") for (int c = 0; c < 8; ++c) { \
                    const bool ok_ = iin[i] && cb + c < P.cin; \
                    const float q_ = *(ok_ ? src + c : P.x); \
                    e8[c] = ok_ ? q_ : 0.f; \
                } \
                rva[i] = make_float4(e8[0], e8[1], e8[2], e8[3]); rvb[i] = make_float4(e8[4], e8[5], e8[6], e8[7]); \
            } \
        } \
        const uint4* wsrc_ = wsrc_all + (size_t)(CH) * W_ITEMS; \
        _Pragma("unroll") for (int j = 0; j < W_PER; ++j) { \
            const int it = tid + j * THREADS; \
            wr[j] = wsrc_[it < W_ITEMS ? it : 0]; \
        } } while (0)
#define K4_B6_STORE() do { \
        _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) { \
            const float v8[8] = {rva[i].x, rva[i].y, rva[i].z, rva[i].w, rvb[i].x, rvb[i].y, rvb[i].z, rvb[i].w}; \
            uint4 t0, t1, t2; \
            k4s_split3(v8, t0, t1, t2); \
            if (idst[i] >= 0) { in_s[idst[i]] = t0; in_s[IN_PLANE + idst[i]] = t1; in_s[2 * IN_PLANE + idst[i]] = t2; } \
        } \
        _Pragma("unroll") for (int j = 0; j < W_PER; ++j) { \
            const int it = tid + j * THREADS; \
            if (it < W_ITEMS) w_s[it] = wr[j]; \
        } } while (0)

    K4_B6_LOAD(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        K4_B6_STORE();
        __syncthreads();
        { const int chn = ch + 1 < nchunks ? ch + 1 : ch; K4_B6_LOAD(chn); }     // unconditional: flies during the MFMAs below
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int dy = t / KS, dx = t - dy * KS;
            bf16x8 a[3][2];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    a[q][m] = __builtin_bit_cast(bf16x8, in_s[q * IN_PLANE + (half * ROWS + wv * 2 + m + dy) * COLS + l31 + dx]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bf16x8 b[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, w_s[((q * TAPS + t) * 2 + half) * NOUT + n * 32 + l31]);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][m], b[0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[2], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[0], acc[m][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#undef K4_B6_LOAD
#undef K4_B6_STORE
    k4_conv_epilogue<NT>(P, acc, x0, y0, wv, half, l31);
}

// ------------------------------------------------------------------------------------------------------------------
// v2 of the 3x3 bf16x6 convolution (default).  rocprofv3 + the ISA of the kernel above showed why it sat at 0.22-0.35 of its
// matrix floor: hipcc waited lgkmcnt(0) in front of nearly every group of 1-4 MFMAs (no operand prefetch distance), spilled the
// prefetched next-chunk weights to scratch (134 VGPRs chosen for an occupancy the 114 KB of LDS forbids anyway), and the 8 waves
// of the single workgroup per CU staged and computed in lock step, so the matrix pipe idled through every staging phase.
// Here:
//   * a workgroup = 4 waves, 16 rows x 32 columns x 32 OUTPUT CHANNELS (layers with 64 output channels run two workgroups per
//     tile, adjacent in launch order: the second finds the tile in L2); LDS holds only the split input tile (58.7 KB) -> TWO
//     workgroups per CU whose staging and MFMA phases interleave;
//   * weights never touch LDS: a wave loads the 3 pre-split fragments of one tap straight from L1/L2 into registers, a ring of
//     3 filled two taps ahead (across the chunk boundary too); activations are fetched with the non-temporal hint so that they do
//     not evict the 28 KB of weight fragments from the 32 KB L1;
//   * a wave owns 4 output rows; the chunk is 36 sub-stages (tap x output row r) of 6 MFMAs; the 3 A fragments of sub-stage u+1
//     are read from LDS while the MFMAs of sub-stage u run (explicit software pipeline, order pinned with sched_barrier);
//   * 2 waves per SIMD, nothing spills.
// Same arithmetic, same operand order per accumulator as above: results are bit-identical to the v1 kernel (tests).
// ------------------------------------------------------------------------------------------------------------------
typedef unsigned k4s_u32x4 __attribute__((ext_vector_type(4)));
#ifndef K4_V2_ARING
#define K4_V2_ARING 3      // A-fragment ring: filled ARING-1 sub-stages ahead
#endif
#ifndef K4_V2_BRING
#define K4_V2_BRING 2      // weight-fragment ring: filled BRING-1 taps ahead
#endif
#ifndef K4_V2_NT
#define K4_V2_NT 0         // 1: non-temporal activation loads
#endif
typedef float k4_f4 __attribute__((ext_vector_type(4)));
// per-tile scalars (workgroup-uniform)
struct V2Tile { const float* x; float* y; const float* res; const float* modx; int H, W, srcW, x0, y0, nb; };
template <int TROWS>
__device__ __forceinline__ V2Tile k4_v2_tile(const ConvMulti& M, int b, int nb_count, bool ups) {
    int g = 0;
    while (g + 1 < M.n && b >= M.blk_end[g]) ++g;
    const int local = b - (g ? M.blk_end[g - 1] : 0);
    const int tile = local / nb_count;
    V2Tile T;
    T.nb = local - tile * nb_count;
    T.x = M.x[g]; T.y = M.y[g]; T.res = M.res[g]; T.modx = M.modx[g]; T.H = M.H[g]; T.W = M.W[g];
    T.srcW = ups ? T.W / 2 : T.W;
    const int tiles_x = M.tiles_x[g];
    T.x0 = (tile % tiles_x) * TILE_W; T.y0 = (tile / tiles_x) * TROWS;
    return T;
}

// RPW = output rows per wave: 4 (16-row tiles) by default; 2 (8-row tiles, half the serial work per workgroup) for launches too
// small to fill the chip with 16-row tiles -- the 209x209 windows of the 8-GPU job are 98 tiles per layer each.
// (A form with the following SFTLayer fused into this kernel's epilogue was bit-identical and measured neutral -- 51.35 / 52.09 ms per
// 4K frame against 51.88 / 51.92 layer by layer, round 2 -- and was removed in round 3: DESIGN.md.)
// NTERM = 2 (flag K4_ARITH_2TERM, the decoder's opt-in 'bf16x3' arithmetic): only the two leading split terms of both operands are
// staged / loaded and 3 of the 6 products are formed (a1 b0 + a0 b1 + a0 b0, ~2^-16 relative per product) -- half the matrix
// instructions, a third less LDS and register traffic, same packed weights (their third term is simply not read).
// F16 (flag K4_ARITH_F16X3, the decoder's 'f16x3' arithmetic): 2-term splits in fp16 instead of bf16 -- 11 + 11 = 22 significant bits per
// operand, 3 products (a_lo b_hi + a_hi b_lo + a_hi b_hi) on v_mfma_f32_32x32x16_f16, ~2^-21 relative per product against 2^-23 of the
// 6-product bf16 form and 2^-16 of the 3-product bf16 form, at half the matrix instructions of the former.  fp16 has 5 exponent bits, so
// both operands are scaled by powers of two (exact): the weights at packing time by 2^a[co] 2^b[chunk] (largest |w| of an output channel
// -> [2^13, 2^14), then the largest scaled |w| of a 16-input-channel chunk -> [2^13, 2^14); the tables ride behind the packed
// terms), the activations per 16-channel chunk of the tile being staged (largest magnitude
// of the chunk's haloed tile -> [2^13, 2^14), found with one wave reduction + 4 LDS words under the previous chunk's MFMAs).  When the
// scale changes between two chunks of a tile the accumulators are re-based by the exact power-of-two ratio.  Elements 2^-16 below the
// largest one of their chunk start to lose bits of the low term (fp16 subnormals): an absolute error 2^-38 of the chunk's scale.
// Same LDS footprint and loop structure as NTERM = 2.
#ifndef K4_V2_MINWG_2T
#define K4_V2_MINWG_2T 2   // workgroups per CU the 2-term instantiation's register allocation is bounded for (39 KB of LDS would allow 3-4)
#endif
#ifndef K4_V2_MINWG_F16
#define K4_V2_MINWG_F16 3  // 8-row tiles (RPW = 2) of the fp16 form: 167 VGPRs, 21.8 KB of LDS -> THREE workgroups per CU (4K frame 39.5 ms against 44.5 at
#endif                     // two); the 12- / 16-row forms would spill 28 / 178 registers at that bound and keep two
// |x|, or 0 for inf / NaN: non-finite activations must not set a chunk's scale (they stay non-finite through the fp16 conversion
// and poison exactly the outputs they reach, as in fp32)
__device__ __forceinline__ float k4s_finite_abs(float x) { const float a = fabsf(x); return a <= 3.4028234e38f ? a : 0.f; }
// largest value of a wave in every lane, for v >= 0 (the bit patterns of non-negative floats order like unsigned integers): 6 DPP
// moves + v_max_u32 instead of 6 ds_bpermute round trips
__device__ __forceinline__ float k4s_wave_max_nonneg(float v) {
    unsigned x = __float_as_uint(v);
#define K4_DPP_MAX(CTRL, ROWMASK) x = max(x, (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROWMASK, 0xf, false))
    K4_DPP_MAX(0x111, 0xf);      // row_shr:1
    K4_DPP_MAX(0x112, 0xf);      // row_shr:2
    K4_DPP_MAX(0x114, 0xf);      // row_shr:4
    K4_DPP_MAX(0x118, 0xf);      // row_shr:8   -> lane 15 of every row of 16 holds the row's maximum
    K4_DPP_MAX(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    K4_DPP_MAX(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef K4_DPP_MAX
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)x, 63));
}
// largest finite magnitude of four values on their bit patterns (non-negative floats order like unsigned integers): 4 x v_and + 2 x v_max3_u32.
// A non-finite value (bits >= 0x7f800000) makes the result >= 0x7f800000: the caller then takes the slow path below.
__device__ __forceinline__ unsigned k4s_absmax4_bits(const float4& v, unsigned m) {
    const unsigned a0 = __float_as_uint(v.x) & 0x7fffffffu, a1 = __float_as_uint(v.y) & 0x7fffffffu, a2 = __float_as_uint(v.z) & 0x7fffffffu, a3 = __float_as_uint(v.w) & 0x7fffffffu;
    return max(max(max(m, a0), max(a1, a2)), a3);
}
// WL = the chunk's weight fragments go through LDS: fetched ONCE per workgroup together with the next chunk's activations (a whole MFMA phase
// ahead), stored beside the input tile, read by every wave with ds_read_b128 one tap ahead.  Without it each wave fetches its fragments
// from L1 / L2 one tap ahead -- 192 cycles of matrix work at 8-row tiles with 3 products, against an L2 hit of ~500+ cycles (a layer's
// weights, 18 KB per chunk x 4..12 chunks, do not stay in the 32 KB L1 beside three workgroups' activation traffic): every layer shape
// of the decoder sat at ~30 % of its matrix floor whatever its size (profiles/r03_sr_kernel_stats.md).
// K4_SR_TIMING (profiling builds only, profiles/r03_commands/r03_call8.sh): s_memtime stamps at the phase boundaries of the chunk loop, summed over all
// waves into k4_sr_timing[] (read and reset through k4_debug_sr_timing).  Not compiled into the product library.
#ifdef K4_SR_TIMING
__device__ unsigned long long k4_sr_timing[16];
#define K4_SR_TSTAMP(SLOT) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                                tacc[SLOT] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define K4_SR_TSTAMP(SLOT) do { } while (0)
#endif
template <int RPW, int NTERM = 3, bool F16 = false, bool WL = false>
__global__ __launch_bounds__(256, (F16 ? (RPW == 2 ? K4_V2_MINWG_F16 : 2) : NTERM == 2 ? K4_V2_MINWG_2T : 2)) void k4_conv_b6v2_kernel(const ConvMulti M) {
    constexpr int NBK = 1;                                    // 32-channel output blocks per workgroup (two were measured neutral: profiles/r03_sr_kernel_stats.md)
    static_assert(!WL || F16, "weights through LDS: fp16 form");
    static_assert(NTERM == 3 || NTERM == 2, "3-term (6 products) or 2-term (3 products) splits");
    static_assert(!F16 || NTERM == 2, "the fp16 arithmetic is a 2-term split");
    constexpr int THREADS = 256;
    constexpr int TROWS = 4 * RPW;
    constexpr int NSUB = 9 * RPW;                             // sub-stages per chunk
    constexpr int ROWS = TROWS + 2, COLS = TILE_W + 2;
    constexpr int NPIX = ROWS * COLS;                         // haloed input tile
    constexpr int IN_PER = (NPIX * 4 + THREADS - 1) / THREADS;   // staged items per thread: (pixel, QUARTER of the 16-channel chunk)
    constexpr int IN_PLANE = 2 * NPIX;                        // uint4 per term
    __shared__ uint4 in_s[NTERM * IN_PLANE];                  // [term][channel group][row][col] x 8 bf16 (F16: 8 fp16)
    __shared__ float smax[2][4];                              // F16: largest |activation| of the chunk being staged, per wave (double buffered)
    constexpr int WCH = NTERM * 9 * 2 * 32 * NBK;             // WL: 16-byte units of one chunk's weight fragments for this workgroup's NBK blocks
    constexpr int W_PER = (WCH + THREADS - 1) / THREADS;
    __shared__ uint4 w_s[WL ? WCH : 1];                       // [term][tap][channel group][32 NBK output channels] x 8 fp16
    const ConvParams& P = M.base;                             // shared by every window: cin, strides, weights, bias, cout, flags ...
    const int nb_count = (P.cout + 31) >> 5;
    const int NOUT = nb_count * 32;
    const int tid = (int)threadIdx.x;
    const int lane = k4_lane();
    const int wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const bool ups = (P.flags & K4_PRE_UPSAMPLE2X) != 0;
    const int nchunks = (P.cin + KC2 - 1) / KC2;
    const int W_ITEMS = (F16 ? 2 : 3) * 9 * 2 * NOUT;                       // 16-byte units of one chunk's split weights
    const bool vec_base = (P.cin_stride & 3) == 0;

#ifdef K4_SR_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    const int bcur = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (bcur < M.total) {
    const V2Tile T = k4_v2_tile<TROWS>(M, bcur, nb_count / NBK, ups);      // T.nb = index of this workgroup's group of NBK output blocks

    // Staging map: thread -> quarter q = tid&3 (4 channels = 16 bytes) of pixels pp = (tid>>2) + 64*i.  Four adjacent lanes read the
    // 64 contiguous bytes of one pixel's chunk, so a wave's load instruction touches 16 cache lines with 64 bytes each (the
    // (pixel, 8-channel) map of the v1 kernel touched 64 pieces of 16 bytes: the L1 tag rate, not bytes, bounded the staging --
    // TCP_TOTAL_CACHE_ACCESSES 2.2e8 per 2080x2080 launch, 0.7 per CU-cycle with the MFMA phase compiled out).
    // Per-thread state of the tile being STAGED (chunk independent): element offsets from the image base, an inside-the-image mask.
    // ioff: offset of the item from the image base -- in BYTES when the window's image is below 2 GB (buf_ok), else in elements -- or
    // K4_V2_OOB for an item outside the image / past the tile.  Whole chunks of 16-byte aligned images below 2 GB are fetched through a
    // buffer descriptor of the window: an offset beyond its range returns zeros in hardware -- no select per component, no 64-bit
    // address arithmetic per load (the staging phase is not covered by matrix work: every vector instruction in it is exposed).
#define K4_V2_OOB 0x80000000u
    unsigned ioff[IN_PER];
    bool vec_ok, buf_ok;
    const float* xbase;
    __amdgpu_buffer_rsrc_t xrsrc;
    const int sq = tid & 3, sp0 = tid >> 2;
#define K4_V2_SETUP(TT) do { \
        const long long xbytes_ = (long long)(ups ? (TT).H / 2 : (TT).H) * (TT).srcW * P.cin_stride * 4; \
        buf_ok = F16 && xbytes_ <= (long long)K4_V2_OOB;           /* the bf16 instantiations keep flat loads: their register budgets are full */ \
        _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) { \
            const int pp = sp0 + 64 * i; \
            const int ppc = pp < NPIX ? pp : 0; \
            const int py = ppc / COLS, px = ppc - py * COLS; \
            const int gy = (TT).y0 - 1 + py, gx = (TT).x0 - 1 + px; \
            const bool inside = pp < NPIX && gy >= 0 && gy < (TT).H && gx >= 0 && gx < (TT).W; \
            const int sy = ups ? (gy >> 1) : gy, sx = ups ? (gx >> 1) : gx; \
            ioff[i] = inside ? (unsigned)((sy * (TT).srcW + sx) * P.cin_stride + sq * 4) << (buf_ok ? 2 : 0) : K4_V2_OOB;      /* elements < 2^31 (launch check) */ \
        } \
        xbase = (TT).x; \
        vec_ok = vec_base && (((size_t)(TT).x) & 15) == 0; \
        const unsigned long long xa_ = (unsigned long long)(TT).x; \
        const unsigned xlo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa_), xhi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xa_ >> 32)); \
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)xhi_ << 32) | xlo_), 0, \
                                                  __builtin_amdgcn_readfirstlane((int)(unsigned)(buf_ok ? xbytes_ : 0)), 0x00020000); } while (0)
    K4_V2_SETUP(T);

    // raw fp32 of the NEXT chunk's tile items: fetched before the MFMA phase of the current chunk, split + stored after it, so that
    // the HBM/L2 latency of the staging is hidden (the two workgroups of a CU start in phase and stay in phase: a synchronous
    // staging phase was fully exposed -- SQ_VALU_MFMA_BUSY_CYCLES showed the matrix pipe 48 % busy)
    float4 rv[IN_PER];
#define K4_V2_LOADRAW(CH) do { \
        const int c0_ = (CH) * KC2; \
        if (vec_ok && !buf_ok && c0_ + KC2 <= P.cin) { \
            _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) {          /* images of 2 GB and more: flat loads + select */ \
                const bool in_ = ioff[i] != K4_V2_OOB; \
                const k4_f4 va = *reinterpret_cast<const k4_f4*>(xbase + (in_ ? ioff[i] : 0u) + c0_); \
                rv[i] = in_ ? make_float4(va.x, va.y, va.z, va.w) : make_float4(0.f, 0.f, 0.f, 0.f); \
            } \
        } else if (vec_ok && c0_ + KC2 <= P.cin) { \
            _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) { \
                const k4s_u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)ioff[i], c0_ * 4, 0); \
                rv[i] = make_float4(__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w)); \
            } \
        } else { \
            _Pragma("unroll") for (int i = 0; i < IN_PER; ++i) { \
                const bool in_ = ioff[i] != K4_V2_OOB; \
                const int cb = c0_ + sq * 4; \
                const float* src = xbase + (in_ ? (ioff[i] >> (buf_ok ? 2 : 0)) : 0u) + c0_; \
                float e4[4]; \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) { \
                    const bool ok_ = in_ && cb + c < P.cin; \
                    const float q_ = *(ok_ ? src + c : xbase); \
                    e4[c] = ok_ ? q_ : 0.f; \
                } \
                rv[i] = make_float4(e4[0], e4[1], e4[2], e4[3]); \
            } \
        } } while (0)
    K4_V2_LOADRAW(0);

    // operand registers: B = the 3 split terms of ONE tap, a ring filled BRING-1 taps ahead (over chunk and tile boundaries: the
    // weights do not depend on the tile);  A = the 3 split terms of one (input row, dx), a ring filled ARING-1 sub-stages ahead
    uint4 bbuf[K4_V2_BRING][NBK][3];
    const uint4* wlane = reinterpret_cast<const uint4*>(P.w) + half * NOUT + T.nb * 32 * NBK + l31;
    const uint4* const wlds = w_s + half * (32 * NBK) + l31;
#define K4_V2_LOADB(DST, CH, TAP) do { \
        if constexpr (WL) { \
            _Pragma("unroll") for (int j_ = 0; j_ < NBK; ++j_) \
            _Pragma("unroll") for (int q_ = 0; q_ < NTERM; ++q_) DST[j_][q_] = wlds[((q_ * 9 + (TAP)) * 2) * (32 * NBK) + j_ * 32]; \
        } else { \
            const uint4* wp_ = wlane + (size_t)(CH) * W_ITEMS; \
            _Pragma("unroll") for (int j_ = 0; j_ < NBK; ++j_) \
            _Pragma("unroll") for (int q_ = 0; q_ < NTERM; ++q_) DST[j_][q_] = wp_[((q_ * 9 + (TAP)) * 2) * NOUT + j_ * 32]; \
        } } while (0)
    // WL: this thread's share of a chunk's fragments, global -> registers (with the activations of the same chunk) -> w_s
    // (nine named registers, not an array: hipcc left `uint4 wr[W_PER]` in scratch memory -- every fragment went global -> VGPR -> scratch
    // -> VGPR -> LDS with the global latency exposed at the scratch store)
    static_assert(W_PER <= 9, "weight staging registers");
    uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
#define K4_V2_WR_LIST(X) X(0, wr0) X(1, wr1) X(2, wr2) X(3, wr3) X(4, wr4) X(5, wr5) X(6, wr6) X(7, wr7) X(8, wr8)
    // byte offsets of this thread's items inside a chunk's fragments (chunk independent); the chunk's base rides in the scalar offset of a
    // buffer load over the packed weights (< 4 GB): no per-chunk address arithmetic in vector registers
    unsigned woffb[WL ? W_PER : 1];
    __amdgpu_buffer_rsrc_t wrsrc;
    if constexpr (WL) {
#pragma unroll
        for (int k = 0; k < W_PER; ++k) { const int it_ = tid + k * THREADS, itc_ = it_ < WCH ? it_ : 0; woffb[k] = (unsigned)(((itc_ / (32 * NBK)) * NOUT + itc_ % (32 * NBK) + T.nb * 32 * NBK) * 16); }
        const unsigned long long wa_ = (unsigned long long)P.w;
        const unsigned wlo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wa_), whi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wa_ >> 32));
        wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)whi_ << 32) | wlo_), 0,
                                                  __builtin_amdgcn_readfirstlane(nchunks * W_ITEMS * 16), 0x00020000);
    }
#ifndef K4_V2_WBUF
#define K4_V2_WBUF 1
#endif
#define K4_V2_LOADW_ONE(K, R) if constexpr (W_PER > K) { \
            if (K4_V2_WBUF) { const k4s_u32x4 wv_ = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)woffb[K], wsoff_, 0); \
                              R = make_uint4(wv_.x, wv_.y, wv_.z, wv_.w); } \
            else R = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(P.w) + (size_t)wsoff_ + woffb[K]); }
#define K4_V2_STOREW_ONE(K, R) if constexpr (W_PER > K) { if (tid + K * THREADS < WCH) w_s[tid + K * THREADS] = R; }
#define K4_V2_LOADW(CH) do { \
        const int wsoff_ = (CH) * W_ITEMS * 16; \
        K4_V2_WR_LIST(K4_V2_LOADW_ONE) } while (0)
    if constexpr (WL) K4_V2_LOADW(0);
    else {
        K4_V2_LOADB(bbuf[0], 0, 0);
        if (K4_V2_BRING == 3) K4_V2_LOADB(bbuf[1], 0, 1);
    }

    uint2* const in2 = reinterpret_cast<uint2*>(in_s);
    const int sdst = ((sq >> 1) * NPIX + sp0) * 2 + (sq & 1);
    const uint4* const arow = in_s + (half * ROWS + wv * RPW) * COLS + l31;   // A fragment of (term q, input row i, dx): arow[q*IN_PLANE + i*COLS + dx]
#define K4_V2_READA(DST, U) do { \
        const int t_ = (U) / RPW, r_ = (U) % RPW; \
        _Pragma("unroll") for (int q_ = 0; q_ < NTERM; ++q_) DST[q_] = arow[q_ * IN_PLANE + (r_ + t_ / 3) * COLS + t_ % 3]; } while (0)

    f32x16 acc[NBK][RPW];
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[j][r] = (f32x16)(0.f);
    // F16: power-of-two scale of the staged activations, one per 16-channel chunk of this tile (see the header comment): 2^sexp maps the
    // chunk's largest magnitude into [2^13, 2^14) -- but the accumulators' base never rises more than 2^60 above the smallest base a
    // chunk of this tile has had, so that re-basing them (an exact multiplication by 2^(new - old), either direction) cannot overflow
    // fp32: a chunk adds at most 144 x 2^14 x 2^14 < 2^36 in its own units
    // The weights carry their own power-of-two factors (packing time): 2^a[co] per output channel (undone per lane in the epilogue) and
    // 2^b[chunk] per 16-input-channel chunk, which simply adds to the chunk's activation exponent: t = sexp + b is what the
    // accumulators are based on.
    // largest FINITE magnitude of this wave's share of the staged chunk -> smax[SLOT][wave]: on the bit patterns (6 instructions per four
    // values); only when a non-finite value is among them (the wave maximum says so, uniformly) the filtering form runs
#define K4_V2_CHUNKMAX(SLOT) do { \
        unsigned mb_ = 0u; \
        _Pragma("unroll") for (int i_ = 0; i_ < IN_PER; ++i_) mb_ = k4s_absmax4_bits(rv[i_], mb_); \
        float m_ = k4s_wave_max_nonneg(__uint_as_float(mb_)); \
        if (__float_as_uint(m_) >= 0x7f800000u || !K4S_INT_MAX) { \
            float mf_ = 0.f; \
            _Pragma("unroll") for (int i_ = 0; i_ < IN_PER; ++i_) mf_ = fmaxf(fmaxf(fmaxf(mf_, k4s_finite_abs(rv[i_].x)), k4s_finite_abs(rv[i_].y)), fmaxf(k4s_finite_abs(rv[i_].z), k4s_finite_abs(rv[i_].w))); \
            m_ = k4s_wave_max_nonneg(mf_); \
        } \
        if (lane == 0) smax[SLOT][wv] = m_; } while (0)
    int sexp = 0, tcur = 0, tmin = 1000;
    const float* const wtail = reinterpret_cast<const float*>(reinterpret_cast<const uint4*>(P.w) + (size_t)nchunks * W_ITEMS);     // [NOUT] 2^-a | [nchunks] b
    if constexpr (F16) {
        K4_V2_CHUNKMAX(0);
        __syncthreads();
    }
    K4_SR_TSTAMP(0);                                     // prologue: tile setup, first chunk's loads issued, (F16) its maximum + barrier
    {
        for (int ch = 0; ch < nchunks; ++ch) {
            if (!WL && K4_V2_BRING == 2 && ch > 0) {  // 9 taps per chunk, ring of 2: the tap prefetched across the chunk boundary sits in the odd buffer
#pragma unroll
                for (int j = 0; j < NBK; ++j)
#pragma unroll
                    for (int q = 0; q < NTERM; ++q) bbuf[0][j][q] = bbuf[1][j][q];
            }
            // ---- split + store this chunk's haloed input tile ----
            if constexpr (F16) {
                const float m = fmaxf(fmaxf(smax[ch & 1][0], smax[ch & 1][1]), fmaxf(smax[ch & 1][2], smax[ch & 1][3]));
                const int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);                 // biased exponent of the chunk's largest magnitude
                const int enat = m > 0.f ? min(max(13 - (eb - 127), -100), 100) : 100;
                const int bch = F16 ? reinterpret_cast<const int*>(wtail + NOUT)[ch] : 0;
                tmin = min(tmin, enat + bch);
                const int tnew = min(enat + bch, tmin + 60);
                if (tnew != tcur && ch > 0) {                                              // workgroup-uniform
                    const int d = tnew - tcur;
#pragma unroll
                    for (int j = 0; j < NBK; ++j)
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[j][r][e] = ldexpf(acc[j][r][e], d);
                }
                tcur = tnew;
                sexp = tnew - bch;                                                         // <= enat: the chunk fits fp16
                const float sc = ldexpf(1.f, sexp);
                if (!(P.debug & 8) || ch == 0)
#pragma unroll
                for (int i = 0; i < IN_PER; ++i) {
                    uint2 t0, t1;
                    k4s_split2h_x4(rv[i], sc, t0, t1);
                    if (sp0 + 64 * i < NPIX) {
                        uint2* const d = in2 + sdst + i * 128;
                        d[0] = t0; d[2 * IN_PLANE] = t1;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < IN_PER; ++i) {
                    uint2 t0, t1, t2;
                    k4s_split3x4(rv[i], t0, t1, t2);
                    if (sp0 + 64 * i < NPIX) {           // 8-byte unit of (term, channel group kg = q>>1, pixel, q&1); i*128 is a constant offset
                        uint2* const d = in2 + sdst + i * 128;
                        d[0] = t0; d[2 * IN_PLANE] = t1;
                        if (NTERM == 3) d[4 * IN_PLANE] = t2;
                    }
                }
            }
            if constexpr (WL) { K4_V2_WR_LIST(K4_V2_STOREW_ONE) }
            K4_SR_TSTAMP(1);                             // wait for the chunk's raw activations, scale, split, LDS stores
            __syncthreads();
            K4_SR_TSTAMP(2);                             // barrier A
            // next staging unit: the next chunk of this tile (its loads fly during the MFMAs below)
            if (ch + 1 < nchunks && !(P.debug & 16)) {
                K4_V2_LOADRAW(ch + 1);
                if constexpr (WL) K4_V2_LOADW(ch + 1);
            }
            if constexpr (WL) {
                K4_V2_LOADB(bbuf[0], ch, 0);
                if (K4_V2_BRING == 3) K4_V2_LOADB(bbuf[1], ch, 1);
            }
            // ---- 9*RPW sub-stages u = tap*RPW + r: 6 (3) MFMAs each into acc[r]; A(u+AD) is fetched from LDS and B(tap+BD) from L1/L2 under them ----
            constexpr int AD = K4_V2_ARING - 1, BD = K4_V2_BRING - 1;
            const int chn = ch + 1 == nchunks ? 0 : ch + 1;                   // the tap ring runs over the chunk boundary
            uint4 abuf[K4_V2_ARING][3];
            K4_V2_READA(abuf[0], 0);
            if (AD == 2) K4_V2_READA(abuf[1], 1);
            K4_SR_TSTAMP(3);                             // next chunk's loads issued, first fragments requested
            if (!(P.debug & 4))
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                const int t = u / RPW, r = u % RPW;
                if (r == 0) {                                                // weights of tap t+BD
                    if (t + BD < 9) K4_V2_LOADB(bbuf[(t + BD) % K4_V2_BRING], ch, t + BD);
                    else if constexpr (!WL) K4_V2_LOADB(bbuf[(t + BD) % K4_V2_BRING], chn, t + BD - 9);
                }
                if (u + AD < NSUB) K4_V2_READA(abuf[(u + AD) % K4_V2_ARING], u + AD);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (F16) {
                    const f16x8 a0 = __builtin_bit_cast(f16x8, abuf[u % K4_V2_ARING][0]), a1 = __builtin_bit_cast(f16x8, abuf[u % K4_V2_ARING][1]);
                    // lo x hi, hi x lo, hi x hi -- per accumulator in this order, the NBK blocks' instructions interleaved
#pragma unroll
                    for (int j = 0; j < NBK; ++j) acc[j][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, __builtin_bit_cast(f16x8, bbuf[t % K4_V2_BRING][j][0]), acc[j][r], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NBK; ++j) acc[j][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, bbuf[t % K4_V2_BRING][j][1]), acc[j][r], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NBK; ++j) acc[j][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, bbuf[t % K4_V2_BRING][j][0]), acc[j][r], 0, 0, 0);
                } else {
                    const bf16x8 a0 = __builtin_bit_cast(bf16x8, abuf[u % K4_V2_ARING][0]), a1 = __builtin_bit_cast(bf16x8, abuf[u % K4_V2_ARING][1]),
                                 a2 = __builtin_bit_cast(bf16x8, abuf[u % K4_V2_ARING][NTERM - 1]);
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, bbuf[t % K4_V2_BRING][0][0]), b1 = __builtin_bit_cast(bf16x8, bbuf[t % K4_V2_BRING][0][1]),
                                 b2 = __builtin_bit_cast(bf16x8, bbuf[t % K4_V2_BRING][0][NTERM - 1]);
                    if (NTERM == 3) {
                        acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[0][r], 0, 0, 0);
                        acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[0][r], 0, 0, 0);
                        acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[0][r], 0, 0, 0);
                    }
                    acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[0][r], 0, 0, 0);
                    acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][r], 0, 0, 0);
                    acc[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][r], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            K4_SR_TSTAMP(4);                             // the MFMA phase
            if constexpr (F16) {
                if (ch + 1 < nchunks) {              // the next chunk's largest magnitude (its raw values have landed under the MFMAs)
                    K4_V2_CHUNKMAX((ch + 1) & 1);
                }
            }
            K4_SR_TSTAMP(5);                             // next chunk's maximum (waits for its raw values)
            __syncthreads();
            K4_SR_TSTAMP(6);                             // barrier B
        }
        // ---- epilogue of tile T: lane holds output channel nb*32 + l31 of pixels x0 + row(reg, half) in rows y0 + wv*4 + r ----
        if (P.debug & 2) {
            float sum_ = 0.f;
#pragma unroll
            for (int j = 0; j < NBK; ++j)
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum_ += acc[j][r][e];
            if (sum_ == 123.456f) T.y[lane] = sum_;
        } else if (!(RPW == 4 && NTERM == 3) &&                       // (that instantiation has no register left for it: it would spill)
                   T.x0 + TILE_W <= T.W && P.slope >= 0.f && P.slope <= 1.f && (long long)T.H * T.W * P.cout_stride * 4 <= 0x80000000LL &&
                   (!(P.flags & K4_EPI_RES) || (long long)T.H * T.W * P.res_stride * 4 <= 0x80000000LL)) {
            // Fast path (every tile but the last column of an image): the tile is whole in x and the images are below 2 GB -> buffer
            // stores / residual loads with the lane's offset computed ONCE per row and the element's pixel offset as a scalar (e is a
            // compile-time index).  The general path below spends ~10 vector instructions per stored element on 64-bit addresses, none of
            // them covered by matrix work (the epilogue has none).  Same values: max(v, v * slope) == LeakyReLU for
            // 0 <= slope <= 1.
            const unsigned long long ya_ = (unsigned long long)T.y, ra_ = (unsigned long long)T.res;
            const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ya_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ya_)),
                0, __builtin_amdgcn_readfirstlane(T.H * T.W * P.cout_stride * 4), 0x00020000);
            const bool has_res = (P.flags & K4_EPI_RES) != 0;
            const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ra_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ra_)),
                0, __builtin_amdgcn_readfirstlane(has_res ? T.H * T.W * P.res_stride * 4 : 0), 0x00020000);
            const float sl = (P.flags & K4_EPI_LRELU) ? P.slope : 1.f;
#pragma unroll
            for (int j = 0; j < NBK; ++j) {
                const int co = (T.nb * NBK + j) * 32 + l31;
                const bool co_ok = co < P.cout;
                // K4_EPI_LRELU_BWD: the LAST 32 output channels are the gradient slice of a LeakyReLU output that this accumulation completes --
                // multiplied by (activation > 0 ? 1 : slope) here instead of by a k4_lrelu_bwd launch behind this one (workgroup-uniform)
                const bool lbwd = (P.flags & K4_EPI_LRELU_BWD) && (T.nb * NBK + j) == nb_count - 1 && co_ok;
                const float bias = P.bias[co_ok ? co : 0];
                float unscale = 1.f;
                if constexpr (F16) unscale = ldexpf(wtail[co_ok ? co : 0], -tcur);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int gy = T.y0 + wv * RPW + r;                                  // wave-uniform
                    if (gy >= T.H) continue;
                    const int pix0 = gy * T.W + T.x0 + 4 * half;
                    const unsigned yoff = co_ok ? (unsigned)((pix0 * P.cout_stride + co) * 4) : 0x80000000u;      // lanes past cout: out of range, dropped
                    const unsigned roff = co_ok ? (unsigned)((pix0 * P.res_stride + co) * 4) : 0x80000000u;
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dp = (e & 3) + 8 * (e >> 2);
                            float v = F16 ? fmaf(acc[j][r][e], unscale, bias) : acc[j][r][e] + bias;
                            v = fmaxf(v, v * sl);
                            v = k4s_mul_add(v, P.res_scale, __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, (int)roff, dp * P.res_stride * 4, 0)));
                            if (lbwd) v = T.modx[(size_t)(pix0 + dp) * P.mod_stride + co] > 0.f ? v : v * P.slope;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc, (int)yoff, dp * P.cout_stride * 4, 0);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int dp = (e & 3) + 8 * (e >> 2);
                            float v = F16 ? fmaf(acc[j][r][e], unscale, bias) : acc[j][r][e] + bias;
                            v = fmaxf(v, v * sl);
                            if (lbwd) v = T.modx[(size_t)(pix0 + dp) * P.mod_stride + co] > 0.f ? v : v * P.slope;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc, (int)yoff, dp * P.cout_stride * 4, 0);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NBK; ++j) {
                const int co = (T.nb * NBK + j) * 32 + l31;
                if (co >= P.cout) continue;
                const float bias = P.bias[co];
                // F16: the accumulators are in units of 2^tcur (activation chunk scale x the chunk's weight factor) x 2^a[co]; all powers of two
                float unscale = 1.f;
                if constexpr (F16) unscale = ldexpf(wtail[co], -tcur);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int gy = T.y0 + wv * RPW + r;
                    if (gy >= T.H) continue;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int gx = T.x0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                        if (gx >= T.W) continue;
                        const size_t pix = (size_t)gy * T.W + gx;
                        float v = F16 ? fmaf(acc[j][r][e], unscale, bias) : acc[j][r][e] + bias;
                        if (P.flags & K4_EPI_LRELU) v = v > 0.f ? v : v * P.slope;
                        if (P.flags & K4_EPI_RES) v = k4s_mul_add(v, P.res_scale, T.res[pix * P.res_stride + co]);
                        if ((P.flags & K4_EPI_LRELU_BWD) && (T.nb * NBK + j) == nb_count - 1) v = T.modx[pix * P.mod_stride + co] > 0.f ? v : v * P.slope;
                        T.y[pix * P.cout_stride + co] = v;
                    }
                }
            }
        }
    }
    K4_SR_TSTAMP(7);                                     // epilogue
    }
#ifdef K4_SR_TIMING
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&k4_sr_timing[i], tacc[i]);
        atomicAdd(&k4_sr_timing[8], 1ull);
    }
#endif
#undef K4_V2_LOADB
#undef K4_V2_LOADW
#undef K4_V2_LOADW_ONE
#undef K4_V2_STOREW_ONE
#undef K4_V2_WR_LIST
#undef K4_V2_READA
#undef K4_V2_LOADRAW
#undef K4_V2_CHUNKMAX
#undef K4_V2_OOB
#undef K4_V2_SETUP
}

// ------------------------------------------------------------------------------------------------------------------
// K-split form of the 3x3 bf16x6 convolution for SMALL images (K4_CONV_SMALL: the 64x64 patch of the joint training step, run_sr.py:829-835).
// The kernels above give a wave whole rows of a tile and walk ALL input-channel chunks: on a 64x64 image that is 32-128 workgroups whose launch
// lasts as long as one wave's chain of cin/16 chunks x 9 taps x 6 products plus their staging (10 us at cin = 64, 25 us at cin = 192 -- for
// 0.5-2.7 GFLOP), and a training iteration issues ~160 of them back to back.  Here a workgroup is ONE row x 32 pixels x 32 output channels and
// its four waves split the input-channel chunks (wave w: chunks w, w+4, ...): 4x the workgroups, a quarter of the chain each.  No LDS staging:
// a lane fetches its A fragment (pixel l31 + dx, 8 channels) of each tap straight from L1 / L2, splits it in registers, and the four partial
// accumulators meet in LDS (16 KB), summed in wave order.  Same products and split as above; the order of the fp32 additions differs from the
// row kernels' (chunks interleaved over the waves), so results agree to fp32 summation order, not bit for bit -- training only: the caller
// asks for it with K4_CONV_SMALL, inference (tile == frame bit-identity) never does.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k4_conv_ks_kernel(const ConvMulti M) {
    const ConvParams& P = M.base;
    const int H = M.H[0], W = M.W[0];
    const float* __restrict__ X = M.x[0];
    float* __restrict__ Y = M.y[0];
    const int nbc = (P.cout + 31) >> 5, NOUT = nbc * 32;
    const int tiles_x = M.tiles_x[0];
    const int b = (int)blockIdx.x;
    const int nb = b % nbc, t_ = b / nbc;
    const int x0 = (t_ % tiles_x) * TILE_W, gy = t_ / tiles_x;
    const int lane = k4_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int nchunks = (P.cin + KC2 - 1) / KC2;
    const bool vec = (P.cin_stride & 3) == 0 && (((size_t)X) & 15) == 0;
    __shared__ float red[4][16][64];
    f32x16 acc = (f32x16)(0.f);
    const uint4* const wbase = reinterpret_cast<const uint4*>(P.w) + (size_t)half * NOUT + nb * 32 + l31;
    const size_t wterm = (size_t)9 * 2 * NOUT;                  // 16-byte units between the split terms of one chunk
    for (int ch = wv; ch < nchunks; ch += 4) {
        const int c0 = ch * KC2 + half * 8;
        const bool full = vec && ch * KC2 + KC2 <= P.cin;         // wave-uniform
        // every fetch of the chunk is issued before the first wait: 18 x 16 bytes of activations (9 taps) and the 27 weight fragments -- ONE round trip
        // to L2 / the fabric per chunk (as a per-tap `if (full)` the compiler emitted nine load-wait pairs in a row)
        float4 ra[9][2];
        if (full) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sy = gy - 1 + t / 3, sx = x0 - 1 + t % 3 + l31;
                const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
                const float* src = X + ((size_t)(in ? sy : 0) * W + (in ? sx : 0)) * P.cin_stride + c0;
                ra[t][0] = *reinterpret_cast<const float4*>(src);
                ra[t][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sy = gy - 1 + t / 3, sx = x0 - 1 + t % 3 + l31;
                const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
                const float* src = X + ((size_t)(in ? sy : 0) * W + (in ? sx : 0)) * P.cin_stride + c0;
                float v8[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool ok = c0 + c < P.cin;
                    v8[c] = ok ? src[c] : 0.f;                    // (a pixel outside the image reads pixel (0, 0): zeroed below)
                }
                ra[t][0] = make_float4(v8[0], v8[1], v8[2], v8[3]); ra[t][1] = make_float4(v8[4], v8[5], v8[6], v8[7]);
            }
        }
        const uint4* const wp = wbase + (size_t)ch * 3 * wterm;
        uint4 bw[9][3];
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int q = 0; q < 3; ++q) bw[t][q] = wp[q * wterm + (size_t)t * 2 * NOUT];
        __builtin_amdgcn_sched_barrier(0);                       // (the scheduler sank the fetches to their uses, 3-6 in flight: keep these 36 in front;
#pragma unroll                                                   //  the last three taps' fragments follow behind the third tap: 240 registers, two waves per SIMD)
        for (int t = 0; t < 9; ++t) {
            if (t == 3) {
#pragma unroll
                for (int u = 6; u < 9; ++u)
#pragma unroll
                    for (int q = 0; q < 3; ++q) bw[u][q] = wp[q * wterm + (size_t)u * 2 * NOUT];
                __builtin_amdgcn_sched_barrier(0);
            }
            const int sy = gy - 1 + t / 3, sx = x0 - 1 + t % 3 + l31;
            const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
            const float v8[8] = {in ? ra[t][0].x : 0.f, in ? ra[t][0].y : 0.f, in ? ra[t][0].z : 0.f, in ? ra[t][0].w : 0.f,
                                 in ? ra[t][1].x : 0.f, in ? ra[t][1].y : 0.f, in ? ra[t][1].z : 0.f, in ? ra[t][1].w : 0.f};
            uint4 s0, s1, s2;
            k4s_split3(v8, s0, s1, s2);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, s0), a1 = __builtin_bit_cast(bf16x8, s1), a2 = __builtin_bit_cast(bf16x8, s2);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bw[t][0]), b1 = __builtin_bit_cast(bf16x8, bw[t][1]), b2 = __builtin_bit_cast(bf16x8, bw[t][2]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);          // smallest products first, as the row kernels
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wv][e][lane] = acc[e];
    __syncthreads();
    // wave w finishes accumulator elements 4w .. 4w+3 (pixels x0 + (e & 3) + 8 (e >> 2) + 4 half, output channel nb*32 + l31): partial sums in wave order
    const int co = nb * 32 + l31;
    if (co >= P.cout) return;
    const float bias = P.bias[co];
    const bool lbwd = (P.flags & K4_EPI_LRELU_BWD) && nb == nbc - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = 4 * wv + k;
        const int gx = x0 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (gx >= W) continue;
        const size_t pix = (size_t)gy * W + gx;
        float v = ((red[0][e][lane] + red[1][e][lane]) + red[2][e][lane]) + red[3][e][lane];
        v += bias;
        if (P.flags & K4_EPI_LRELU) v = v > 0.f ? v : v * P.slope;
        if (P.flags & K4_EPI_RES) v = k4s_mul_add(v, P.res_scale, M.res[0][pix * P.res_stride + co]);
        if (lbwd) v = M.modx[0][pix * P.mod_stride + co] > 0.f ? v : v * P.slope;
        Y[pix * P.cout_stride + co] = v;
    }
}

static int launch_conv_b6v2(ConvMulti& M, hipStream_t st) {
    const int nbc = (M.base.cout + 31) / 32;
    // K4_CONV_SMALL: one window, plain 3-term arithmetic, an image small enough that one-row workgroups do not exceed four per CU
    if ((M.base.flags & K4_CONV_SMALL) && M.n == 1 && !(M.base.flags & (K4_ARITH_2TERM | K4_ARITH_F16X3 | K4_PRE_UPSAMPLE2X)) && !(k4_env().sr_debug & 2048)) {
        const long long wgs = (long long)((M.W[0] + TILE_W - 1) / TILE_W) * M.H[0] * nbc;
        if (wgs <= 4ll * k4_num_cus() && (long long)M.H[0] * M.W[0] * M.base.cin_stride <= 0x7fffffffLL) {
            M.tiles_x[0] = (M.W[0] + TILE_W - 1) / TILE_W;
            hipLaunchKernelGGL(k4_conv_ks_kernel, dim3((unsigned)wgs), dim3(256), 0, st, M);
            return k4_check_launch();
        }
    }
    const int slots = 2 * k4_num_cus();
    auto count = [&](int trows) {
        int total = 0;
        for (int g = 0; g < M.n; ++g) {
            M.tiles_x[g] = (M.W[g] + TILE_W - 1) / TILE_W;
            total += M.tiles_x[g] * ((M.H[g] + trows - 1) / trows) * nbc;
            M.blk_end[g] = total;
        }
        return total;
    };
    for (int g = 0; g < M.n; ++g)                       // the kernel addresses the image with 32-bit element offsets
        if ((int64_t)M.H[g] * M.W[g] * M.base.cin_stride > 0x7fffffffLL) return K4_ERR_UNSUPPORTED;
    int total = count(16);
    // Launches of at most two "rounds" of 16-row tiles (the 8-GPU job's windows; layers of small images): pick the tile height
    // (4 / 8 / 12 / 16 rows) that minimises rounds x serial work per workgroup (rows + halo / staging overhead).  4-row tiles: the 64x64
    // training patch is 16 (32) tiles of 8 rows per 32 output channels on 256 CUs -- the launch lasts as long as ONE workgroup's chain
    // of chunks x 9 taps x rows x 6 products; halving the rows per wave halves it (same accumulation order per pixel: bit-identical).
    int rpw = 4;
    // fp16 form: ALWAYS 8-row tiles.  Its activation scale is found per haloed workgroup tile, and elements 2^-16 below a chunk's maximum
    // round their low term as fp16 subnormals -- with a tile height picked from the launch size (below), the bits of a window would
    // depend on how many other windows share its launch and on the CU count.  One tile geometry => a window's result is a function of the
    // window alone (tile_parallel's frame == the single-GPU frame, whatever the grouping).
    if (!(M.base.flags & K4_ARITH_F16X3) && total <= 2 * slots) {
        float best = 1e30f;
        for (int cand = 4; cand >= 1; --cand) {
            const int c = count(4 * cand);
            const float cost = (float)((c + slots - 1) / slots) * ((float)cand + 0.6f);
            if (cost < best - 1e-3f) { best = cost; rpw = cand; }
        }
        total = count(4 * rpw);
    }
    const bool two = (M.base.flags & (K4_ARITH_2TERM | K4_ARITH_F16X3)) != 0;
    if (rpw == 4 && two) {                                   // beyond the small-launch rule: 8-row tiles for the 2-term kernel (3 workgroups per CU), 16-row for 3-term
        rpw = 2;
        total = count(4 * rpw);
    }
    M.total = total;
    if (k4_env().sr_debug & 1) M.base.cin_stride = 0;        // profiling only (WRONG results): every pixel reads the same 64 bytes -- the kernel without memory traffic
    M.base.debug = k4_env().sr_debug;
    const dim3 grid((unsigned)total), block(256);
#define K4_V2_LAUNCH(...) do { \
        if (rpw == 1) hipLaunchKernelGGL((k4_conv_b6v2_kernel<1, __VA_ARGS__>), grid, block, 0, st, M); \
        else if (rpw == 2) hipLaunchKernelGGL((k4_conv_b6v2_kernel<2, __VA_ARGS__>), grid, block, 0, st, M); \
        else if (rpw == 3) hipLaunchKernelGGL((k4_conv_b6v2_kernel<3, __VA_ARGS__>), grid, block, 0, st, M); \
        else hipLaunchKernelGGL((k4_conv_b6v2_kernel<4, __VA_ARGS__>), grid, block, 0, st, M); } while (0)
    if (M.base.flags & K4_ARITH_F16X3) hipLaunchKernelGGL((k4_conv_b6v2_kernel<2, 2, true, true>), grid, block, 0, st, M);      // fp16 form: always 8-row tiles (above), weight fragments through LDS
    else if (M.base.flags & K4_ARITH_2TERM) K4_V2_LAUNCH(2, false);
    else K4_V2_LAUNCH(3, false);
#undef K4_V2_LAUNCH
    return k4_check_launch();
}

// fills blk_end / tiles_x of the jobs for a kernel whose workgroup covers tile_rows x TILE_W output pixels -> total workgroups
static int k4_multi_grid(ConvMulti& M, int tile_rows) {
    int total = 0;
    for (int g = 0; g < M.n; ++g) {
        M.tiles_x[g] = (M.W[g] + TILE_W - 1) / TILE_W;
        total += M.tiles_x[g] * ((M.H[g] + tile_rows - 1) / tile_rows);
        M.blk_end[g] = total;
    }
    return total;
}

template <int KS, int NT, int NW>
static int launch_conv_b6(ConvMulti& M, hipStream_t st) {
    constexpr int TAPS = KS * KS, PADW = KS / 2;
    constexpr size_t lds = ((size_t)3 * 2 * (2 * NW + 2 * PADW) * (TILE_W + 2 * PADW) + (size_t)3 * TAPS * 2 * NT * 32) * sizeof(uint4);
    static_assert(lds <= 160 * 1024, "split chunk must fit the CU's LDS");
    K4_ENSURE_DYN_LDS((k4_conv_b6_kernel<KS, NT, NW>), lds);
    const dim3 grid((unsigned)k4_multi_grid(M, 2 * NW)), block(64 * NW);
    hipLaunchKernelGGL((k4_conv_b6_kernel<KS, NT, NW>), grid, block, lds, st, M);
    return k4_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 convolution with <= 3 output channels (SFTNet.conv_last, 64 -> 3 at 4x resolution).  As an implicit GEMM with
// N = C_out it would waste 29 of every 32 matrix columns.  Here the nine taps become the N dimension instead:
//   Y[p][t*3 + co] = sum_c X[p][c] * W[co][c][t]          one 1x1 GEMM, N = 27 (padded to 32), over the HALOED tile,
//   out[p][co]     = bias[co] + sum_t Y[p + offset(t)][t*3 + co]      a 9-point gather-sum from LDS,
// 9x fewer matrix instructions than the padded 3x3 form.  Same 3-term split arithmetic as k4_conv_b6_kernel.
// Workgroup = 8 waves, 16 x 32 output pixels; the haloed tile is 18 x 34 = 612 pixels = 20 row blocks of 32 (the last
// one partly empty), wave w accumulates row blocks w, w+8, w+16.  Weights: the host packs [27->32][C_in] as a 1x1 layer
// in the bf16x6 layout (flag K4_W_TAPS_AS_COUT).
// ------------------------------------------------------------------------------------------------------------------
#define K4_TAPS_ROWS (TILE_HB + 2)
#define K4_TAPS_COLS (TILE_W + 2)
#define K4_TAPS_NPIX (K4_TAPS_ROWS * K4_TAPS_COLS)        /* 612 */
#define K4_TAPS_PPAD 640                                   /* 20 row blocks */
#define K4_TAPS_YS 29                                      /* Y row stride in floats (27 used; odd -> conflict-free gathers) */

__global__ __launch_bounds__(512) void k4_conv_taps_b6_kernel(const ConvMulti M) {
    int tile;
    const ConvParams P = k4_select_job(M, tile);
    extern __shared__ uint4 k4_taps_smem[];
    uint4* const in_s = k4_taps_smem;                                    // [term][channel group][pixel] x 8 bf16
    uint4* const w_s = k4_taps_smem + 3 * 2 * K4_TAPS_PPAD;              // [term][channel group][n] x 8 bf16
    float* const y_s = reinterpret_cast<float*>(k4_taps_smem);           // [pixel][29], aliases in_s/w_s after the GEMM
    const int tid = (int)threadIdx.x;
    const int lane = k4_lane();
    const int wv = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int x0 = tx * TILE_W, y0 = ty * TILE_HB;
    const bool vec_ok = (P.cin_stride & 3) == 0 && (((size_t)P.x) & 15) == 0;

    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x16)(0.f);

    const int nchunks = (P.cin + KC2 - 1) / KC2;
    const uint4* const wsrc_all = reinterpret_cast<const uint4*>(P.w);
    // the 16-channel chunks of the haloed tile: chunk c + 1 is requested before the matrix instructions of chunk c (16-byte rows, whole
    // chunks); each thread owns up to three (pixel, 8-channel group) items, their addresses are computed once
    const bool piped = vec_ok && (P.cin % KC2) == 0;
    const float* isrc[3];
    bool iin[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int it = tid + 512 * j;
        const int kg = it & 1, pp = it >> 1;
        const int py = pp / K4_TAPS_COLS, px = pp - py * K4_TAPS_COLS;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        iin[j] = it < K4_TAPS_PPAD * 2 && pp < K4_TAPS_NPIX && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        isrc[j] = iin[j] ? P.x + ((size_t)gy * P.W + gx) * P.cin_stride + kg * 8 : P.x;
    }
    float4 pa[3], pb[3];
    uint4 wpre = make_uint4(0, 0, 0, 0);
    const int wtid = tid < 3 * 2 * 32 ? tid : 0;
    if (piped) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pa[j] = *reinterpret_cast<const float4*>(isrc[j]);
            pb[j] = *reinterpret_cast<const float4*>(isrc[j] + 4);
        }
        wpre = wsrc_all[wtid];
    }
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * KC2;
        if (piped) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int it = tid + 512 * j;
                const float v[8] = {iin[j] ? pa[j].x : 0.f, iin[j] ? pa[j].y : 0.f, iin[j] ? pa[j].z : 0.f, iin[j] ? pa[j].w : 0.f,
                                    iin[j] ? pb[j].x : 0.f, iin[j] ? pb[j].y : 0.f, iin[j] ? pb[j].z : 0.f, iin[j] ? pb[j].w : 0.f};
                uint4 t0, t1, t2;
                k4s_split3(v, t0, t1, t2);
                if (it < K4_TAPS_PPAD * 2) {
                    const int kg = it & 1, pp = it >> 1;
                    in_s[(0 * 2 + kg) * K4_TAPS_PPAD + pp] = t0;
                    in_s[(1 * 2 + kg) * K4_TAPS_PPAD + pp] = t1;
                    in_s[(2 * 2 + kg) * K4_TAPS_PPAD + pp] = t2;
                }
            }
            if (tid < 3 * 2 * 32) w_s[tid] = wpre;
            __syncthreads();
            const int chn = ch + 1 < nchunks ? ch + 1 : ch;                 // the last round re-reads its own chunk (unused)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float* q = iin[j] ? isrc[j] + chn * KC2 : P.x;
                pa[j] = *reinterpret_cast<const float4*>(q);
                pb[j] = *reinterpret_cast<const float4*>(q + 4);
            }
            wpre = wsrc_all[(size_t)chn * 3 * 2 * 32 + wtid];
        } else {
        for (int it = tid; it < K4_TAPS_PPAD * 2; it += 512) {
            const int kg = it & 1, pp = it >> 1;
            const int py = pp / K4_TAPS_COLS, px = pp - py * K4_TAPS_COLS;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool inside = pp < K4_TAPS_NPIX && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            const int cb = c0 + kg * 8;
            float v[8];
            const float* src = inside ? P.x + ((size_t)gy * P.W + gx) * P.cin_stride + cb : P.x;
            if (vec_ok && c0 + KC2 <= P.cin) {
                const float4 a = *reinterpret_cast<const float4*>(src);
                const float4 b = *reinterpret_cast<const float4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = inside ? v[c] : 0.f;
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool ok_ = inside && cb + c < P.cin;
                    const float q_ = *(ok_ ? src + c : P.x);
                    v[c] = ok_ ? q_ : 0.f;
                }
            }
            uint4 t0, t1, t2;
            k4s_split3(v, t0, t1, t2);
            in_s[(0 * 2 + kg) * K4_TAPS_PPAD + pp] = t0;
            in_s[(1 * 2 + kg) * K4_TAPS_PPAD + pp] = t1;
            in_s[(2 * 2 + kg) * K4_TAPS_PPAD + pp] = t2;
        }
        if (tid < 3 * 2 * 32) w_s[tid] = wsrc_all[(size_t)ch * 3 * 2 * 32 + tid];
        __syncthreads();
        }
        bf16x8 b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, w_s[(q * 2 + half) * 32 + l31]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rb = wv + 8 * i;
            if (rb < K4_TAPS_PPAD / 32) {
                bf16x8 a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(bf16x8, in_s[(q * 2 + half) * K4_TAPS_PPAD + rb * 32 + l31]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // Y tile to LDS (aliases the staging buffers: every wave is past its last operand read)
    if (l31 < 9 * P.cout) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rb = wv + 8 * i;
            if (rb < K4_TAPS_PPAD / 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pp = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pp < K4_TAPS_NPIX) y_s[pp * K4_TAPS_YS + l31] = acc[i][r];
                }
            }
        }
    }
    __syncthreads();
    // 9-point gather-sum, one output pixel per thread
    const int oy = tid >> 5, ox = tid & 31;
    const int gy = y0 + oy, gx = x0 + ox;
    if (gy < P.H && gx < P.W) {
        const size_t pix = (size_t)gy * P.W + gx;
        for (int co = 0; co < P.cout; ++co) {
            float v = P.bias[co];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t - dy * 3;
                v += y_s[((oy + dy) * K4_TAPS_COLS + ox + dx) * K4_TAPS_YS + t * P.cout + co];
            }
            if (P.flags & K4_EPI_LRELU) v = v > 0.f ? v : v * P.slope;
            if (P.flags & K4_EPI_RES) v = k4s_mul_add(v, P.res_scale, P.res[pix * P.res_stride + co]);
            P.y[pix * P.cout_stride + co] = v;
        }
    }
}

static int launch_conv_taps_b6(ConvMulti& M, hipStream_t st) {
    constexpr size_t lds_gemm = ((size_t)3 * 2 * K4_TAPS_PPAD + 3 * 2 * 32) * sizeof(uint4);
    constexpr size_t lds_y = (size_t)K4_TAPS_NPIX * K4_TAPS_YS * sizeof(float);
    constexpr size_t lds = lds_gemm > lds_y ? lds_gemm : lds_y;
    K4_ENSURE_DYN_LDS(k4_conv_taps_b6_kernel, lds);
    const dim3 grid((unsigned)k4_multi_grid(M, TILE_HB)), block(512);
    hipLaunchKernelGGL(k4_conv_taps_b6_kernel, grid, block, lds, st, M);
    return k4_check_launch();
}

extern "C" int64_t k4_conv_weight_bf16x6_bytes(int32_t cout, int32_t cin, int32_t ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return -1;
    const int64_t nt = (cout + 31) / 32;
    if (ksize == 1 ? (nt != 1 && nt != 2 && nt != 4) : nt > 8) return -1;      // 3x3 (v2 kernel): any number of 32-channel blocks up to 256 channels
    return (int64_t)((cin + KC2 - 1) / KC2) * 3 * ksize * ksize * 2 * nt * 32 * 16;
}

static int conv_b6_multi(const k4_conv_job* jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                         const void* w_split, const float* bias, int32_t ksize, int32_t cout, int32_t cout_stride,
                         uint32_t flags, float slope, int32_t res_stride, float res_scale, int32_t mod_stride,
                         void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > K4_MAX_JOBS || !w_split || !bias || cin <= 0 || cout <= 0) return K4_ERR_BAD_ARG;
    if (cin_stride < cin || (ksize != 1 && ksize != 3) || cout_stride < cout) return K4_ERR_BAD_ARG;
    const bool modulate = (flags & K4_EPI_MODULATE) != 0;
    if (modulate && (mod_stride <= 0 || cout % 32 != 0)) return K4_ERR_BAD_ARG;
    if ((flags & K4_EPI_RES) && res_stride <= 0) return K4_ERR_BAD_ARG;
    const bool lbwd = (flags & K4_EPI_LRELU_BWD) != 0;
    if (lbwd && (ksize != 3 || cout % 32 != 0 || mod_stride < cout || modulate || (flags & K4_W_TAPS_AS_COUT))) return K4_ERR_BAD_ARG;
    const int gemm_n = modulate ? 2 * cout : cout;
    const int nt = (gemm_n + 31) / 32;
    ConvMulti M{};
    ConvParams& P = M.base;
    P.cin = cin; P.cin_stride = cin_stride; P.w = (const float*)w_split; P.bias = bias;
    P.cout = cout; P.cout_stride = cout_stride;
    P.flags = flags; P.slope = slope; P.res_stride = res_stride; P.res_scale = res_scale; P.mod_stride = mod_stride;
    M.n = n_jobs;
    for (int g = 0; g < n_jobs; ++g) {
        const k4_conv_job& j = jobs[g];
        if (!j.x || !j.y || j.H <= 0 || j.W <= 0) return K4_ERR_BAD_ARG;
        if ((flags & K4_EPI_RES) && !j.res) return K4_ERR_BAD_ARG;
        if ((modulate || lbwd) && !j.mod_x) return K4_ERR_BAD_ARG;
        if ((flags & K4_PRE_UPSAMPLE2X) && ((j.H & 1) || (j.W & 1))) return K4_ERR_BAD_ARG;
        M.x[g] = j.x; M.y[g] = j.y; M.res[g] = j.res; M.modx[g] = j.mod_x; M.H[g] = j.H; M.W[g] = j.W;
    }
    hipStream_t st = (hipStream_t)stream;
    if (flags & K4_W_TAPS_AS_COUT) {
        if (ksize != 3 || cout > 3 || modulate || (flags & K4_PRE_UPSAMPLE2X)) return K4_ERR_BAD_ARG;
        return launch_conv_taps_b6(M, st);
    }
    if ((flags & K4_ARITH_2TERM) && (flags & K4_ARITH_F16X3)) return K4_ERR_BAD_ARG;
    if (ksize == 3) {
        if (modulate) return K4_ERR_UNSUPPORTED;               // no 3x3 layer of the decoder modulates (SFTLayer's convolutions are 1x1)
        return launch_conv_b6v2(M, st);
    }
    if (flags & (K4_ARITH_2TERM | K4_ARITH_F16X3)) return K4_ERR_BAD_ARG;      // 1x1 layers keep the 6-product form
    if (nt == 1) return launch_conv_b6<1, 1, 8>(M, st);
    if (nt == 2) return launch_conv_b6<1, 2, 8>(M, st);
    if (nt == 4) return launch_conv_b6<1, 4, 8>(M, st);
    return K4_ERR_UNSUPPORTED;
}

#ifdef K4_SR_TIMING
extern "C" int k4_debug_sr_timing(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(k4_sr_timing), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) { unsigned long long z[16] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(k4_sr_timing), z, sizeof(z)); }
    return (int)e;
}
#endif

extern "C" int64_t k4_conv_weight_f16x3_bytes(int32_t cout, int32_t cin, int32_t ksize) {
    if (cout <= 3 || cin <= 0 || ksize != 3 || (cout + 31) / 32 > 8) return -1;
    const int64_t nch = (cin + KC2 - 1) / KC2, nout = (int64_t)((cout + 31) / 32) * 32;
    return nch * 2 * 9 * 2 * nout * 16 + nout * 4 + (nch + 3) / 4 * 16;            // split terms | [NOUT] fp32 2^-a[co] | [nch] int32 b[chunk] (padded to 16 B)
}

extern "C" int k4_conv2d_nhwc_bf16x6_multi(const k4_conv_job* jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                                           const void* w_split, const float* bias, int32_t ksize, int32_t cout, int32_t cout_stride,
                                           uint32_t flags, float slope, int32_t res_stride, float res_scale, int32_t mod_stride,
                                           void* stream) {
    return conv_b6_multi(jobs, n_jobs, cin, cin_stride, w_split, bias, ksize, cout, cout_stride, flags, slope, res_stride, res_scale,
                         mod_stride, stream);
}

extern "C" int k4_conv2d_nhwc_bf16x6(const float* x, int32_t cin, int32_t cin_stride,
                                     const void* w_split, const float* bias, int32_t ksize,
                                     float* y, int32_t cout, int32_t cout_stride,
                                     int32_t H, int32_t W, uint32_t flags, float slope,
                                     const float* res, int32_t res_stride, float res_scale,
                                     const float* mod_x, int32_t mod_stride, void* stream) {
    return k4_taped(stream, [=](void* stream) -> int {                    // recordable (k4_tape.hip): the training graph's convolutions
        k4_conv_job j{};
        j.x = x; j.y = y; j.res = res; j.mod_x = mod_x; j.H = H; j.W = W;
        return conv_b6_multi(&j, 1, cin, cin_stride, w_split, bias, ksize, cout, cout_stride, flags, slope, res_stride, res_scale, mod_stride,
                             stream);
    });
}

// ------------------------------------------------------------------------------------------------------------------
// Fused SFTLayer (lib/sr_esrnet.py:112-123): y = x * (scale(cond) + 1) + shift(cond) [* res_scale + res] in ONE launch,
// scale/shift = 1x1 conv -> LeakyReLU(0.2) -> 1x1 conv of the 32-channel condition map.  As two separate convolutions
// the pair was bound by launch/sync overhead and by the round trip of the 64-channel hidden map through HBM (it took
// 18 % of the decoder for 6 % of its FLOPs).  Here a wave owns 64 pixels and chains both GEMMs on the matrix cores in
// the transposed form H^T[channel][pixel] = W . X^T (as the marcher's MLP does): the accumulator layout of GEMM 1 is
// the B-operand layout of GEMM 2, so the hidden activations never leave registers; only cond, x and y touch memory.
// Packed weights (host: SFTNet._pack_sft): WA [2][17][64] | WS [C/32][17][64] | WH [C/32][17][64], k-step 16 = bias.
// ------------------------------------------------------------------------------------------------------------------
struct SftParams {
    const float* cond; int cond_stride;       // [n_pix][cond_stride], 32 channels read
    const float* w;
    const float* x; int x_stride;
    float* y; int y_stride;
    const float* res; int res_stride; float res_scale;
    int n_pix; float slope;
    int vec4;                                  // all of x / y / res rows are 16-byte aligned (strides % 4 == 0, bases % 16 == 0)
    float out_scale; uint32_t* overflow;       // p16 output (k4_sft_nhwc_p16_multi): 2^E of the produced tensor, the windows' overflow words
};
struct SftMulti {                              // grouped launch over the windows of a frame (see ConvMulti)
    SftParams base;
    int n;
    int blk_end[K4_MAX_JOBS];
    const float* cond[K4_MAX_JOBS]; const float* x[K4_MAX_JOBS]; float* y[K4_MAX_JOBS]; const float* res[K4_MAX_JOBS];
    int n_pix[K4_MAX_JOBS];
};

template <int CB>
__global__ __launch_bounds__(256) void k4_sft_kernel(const SftMulti M) {
    SftParams P = M.base;
    int blk = (int)blockIdx.x;
    {
        int g = 0;
        while (g + 1 < M.n && blk >= M.blk_end[g]) ++g;
        blk -= g ? M.blk_end[g - 1] : 0;
        P.cond = M.cond[g]; P.x = M.x[g]; P.y = M.y[g]; P.res = M.res[g]; P.n_pix = M.n_pix[g];
    }
    constexpr int NW = (2 + 2 * CB) * 17 * 64;
    __shared__ float wl[NW];
    __shared__ float ct[4][32][64];               // per wave: cond^T [channel][pixel]
    const int lane = k4_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    for (int i = (int)threadIdx.x; i < NW; i += 256) wl[i] = P.w[i];
    const int base = (blk * 4 + wv) * 64;
    {
        const int pix = base + lane;
        float4 c4[8];
        if (pix < P.n_pix) {
            const float4* src = reinterpret_cast<const float4*>(P.cond + (size_t)pix * P.cond_stride);
#pragma unroll
            for (int q = 0; q < 8; ++q) c4[q] = src[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) c4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ct[wv][4 * q + 0][lane] = c4[q].x; ct[wv][4 * q + 1][lane] = c4[q].y;
            ct[wv][4 * q + 2][lane] = c4[q].z; ct[wv][4 * q + 3][lane] = c4[q].w;
        }
    }
    __syncthreads();
    // ---- GEMM 1: hidden^T[64][pix] = lrelu([scale0; shift0] . cond^T + b) ----
    f32x16 h[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) { h[mb][0] = (f32x16)(0.f); h[mb][1] = (f32x16)(0.f); }
    const float* wa = wl;
#pragma unroll
    for (int kk = 0; kk < 17; ++kk) {
        const float b0 = kk < 16 ? ct[wv][2 * kk + half][l31] : 1.f;
        const float b1 = kk < 16 ? ct[wv][2 * kk + half][32 + l31] : 1.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float a = wa[(mb * 17 + kk) * 64 + lane];
            h[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, h[mb][0], 0, 0, 0);
            h[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, h[mb][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = h[mb][t][r]; h[mb][t][r] = v > 0.f ? v : v * P.slope; }
    // ---- GEMM 2 + modulation, 32 output channels at a time ----
    const float* ws = wl + 2 * 17 * 64;
    const float* wh = ws + CB * 17 * 64;
#pragma unroll 1
    for (int mb2 = 0; mb2 < CB; ++mb2) {
        f32x16 cs[2] = {(f32x16)(0.f), (f32x16)(0.f)}, ch[2] = {(f32x16)(0.f), (f32x16)(0.f)};
#pragma unroll
        for (int r = 0; r < 17; ++r) {
            const float as = ws[(mb2 * 17 + r) * 64 + lane];
            const float ah = wh[(mb2 * 17 + r) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                cs[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, r < 16 ? h[0][t][r < 16 ? r : 0] : 1.f, cs[t], 0, 0, 0);
                ch[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ah, r < 16 ? h[1][t][r < 16 ? r : 0] : 1.f, ch[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int pix = base + t * 32 + l31;
            if (pix >= P.n_pix) continue;
            // accumulator registers 4q..4q+3 are 4 CONSECUTIVE channels (co = 8q + 4*half + 0..3): 16-byte accesses
            const bool vec = P.vec4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = mb2 * 32 + 8 * q + 4 * half;
                float xv[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (vec) {
                    const float4 x4 = *reinterpret_cast<const float4*>(P.x + (size_t)pix * P.x_stride + co);
                    xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
                    if (P.res) {
                        const float4 r4 = *reinterpret_cast<const float4*>(P.res + (size_t)pix * P.res_stride + co);
                        rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xv[e] = P.x[(size_t)pix * P.x_stride + co + e];
                        if (P.res) rv[e] = P.res[(size_t)pix * P.res_stride + co + e];
                    }
                }
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = k4s_mul_add(xv[e], cs[t][4 * q + e] + 1.f, ch[t][4 * q + e]);               // x*(scale+1)+shift
                    if (P.res) v = k4s_mul_add(v, P.res_scale, rv[e]);
                    ov[e] = v;
                }
                if (vec) *reinterpret_cast<float4*>(P.y + (size_t)pix * P.y_stride + co) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) P.y[(size_t)pix * P.y_stride + co + e] = ov[e];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same fused SFTLayer on the bf16 matrix pipe with the exact 3-term split (the decoder's default arithmetic): the fp32-input
// MFMA form above runs the layer's 6 % of the decoder FLOPs at 1/16 of the bf16 rate (12 % of the frame time).  Per 32-pixel
// tile: GEMM 1 hidden^T[64][pix] = lrelu(WA . cond^T + ba) -- B operand = 8 consecutive condition channels of the lane's pixel,
// split in registers; GEMM 2 scale / shift = WS . h[0..31] + bs, WH . h[32..63] + bh -- the accumulator layout of GEMM 1 is the
// B-operand layout of GEMM 2 when K is walked in accumulator-register order (as in the marcher's rgbnet), so the hidden
// activations never leave registers; the biases are the accumulators' initial values.
// Split section of the packed buffer (after the fp32 section; host: sr_esrnet.SFTNet._pack_sft), 16-byte units = 8 bf16, lane l:
//   WA6 [2][2][3][64]   WA[mb*32 + (l&31)][kb*16 + 8*(l>>5) + e]
//   WS6 [CB][2][3][64]  WS[mb2*32 + (l&31)][n(kb, l>>5, e)],  n = (e&3) + 8*(2*kb + (e>>2)) + 4*(l>>5);   WH6 likewise
//   BA [2][2][16], BS [CB][2][16], BH [CB][2][16]  fp32: bias[blk*32 + (r&3) + 8*(r>>2) + 4*half]
// ------------------------------------------------------------------------------------------------------------------
#define K4_SFT6_FLOATS(CB) ((2 + 2 * (CB)) * 2 * 3 * 64 * 4 + (2 + 2 * (CB)) * 2 * 16)
#define K4_B6(ACC, A0, A1, A2, B0, B1, B2) do { \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2), __builtin_bit_cast(bf16x8, B0), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), __builtin_bit_cast(bf16x8, B2), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1), __builtin_bit_cast(bf16x8, B1), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1), __builtin_bit_cast(bf16x8, B0), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), __builtin_bit_cast(bf16x8, B1), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A0), __builtin_bit_cast(bf16x8, B0), ACC, 0, 0, 0); } while (0)

template <int CB>
__global__ __launch_bounds__(256) void k4_sft_b6_kernel(const SftMulti M) {
    constexpr int NW6 = K4_SFT6_FLOATS(CB);
    constexpr int NW32 = (2 + 2 * CB) * 17 * 64;
    __shared__ __attribute__((aligned(16))) float wl[NW6];
    SftParams P = M.base;
    int blk = (int)blockIdx.x;
    {
        int g = 0;
        while (g + 1 < M.n && blk >= M.blk_end[g]) ++g;
        blk -= g ? M.blk_end[g - 1] : 0;
        P.cond = M.cond[g]; P.x = M.x[g]; P.y = M.y[g]; P.res = M.res[g]; P.n_pix = M.n_pix[g];
    }
    const int lane = k4_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    {
        const float4* src = reinterpret_cast<const float4*>(P.w + NW32);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = (int)threadIdx.x; i < NW6 / 4; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const uint4* const wa6 = reinterpret_cast<const uint4*>(wl);
    const uint4* const ws6 = wa6 + 2 * 2 * 3 * 64;
    const uint4* const wh6 = ws6 + CB * 2 * 3 * 64;
    const float* const ba = wl + (2 + 2 * CB) * 2 * 3 * 64 * 4;
    const float* const bs = ba + 2 * 2 * 16;
    const float* const bh = bs + CB * 2 * 16;
    const int base = (blk * 4 + wv) * 64;
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        const int pix = base + t * 32 + l31;
        const bool pok = pix < P.n_pix;
        // ---- GEMM 1: hidden^T = lrelu(WA . cond^T + ba) ----
        f32x16 h[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[mb][r] = ba[(mb * 2 + half) * 16 + r];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float v[8];
            if (pok) {
                const float4* src = reinterpret_cast<const float4*>(P.cond + (size_t)pix * P.cond_stride + kb * 16 + 8 * half);
                const float4 c0 = src[0], c1 = src[1];
                v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w; v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            uint4 x0, x1, x2;
            k4s_split3(v, x0, x1, x2);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const uint4* wp = wa6 + ((mb * 2 + kb) * 3) * 64 + lane;
                const uint4 a0 = wp[0], a1 = wp[64], a2 = wp[128];
                K4_B6(h[mb], a0, a1, a2, x0, x1, x2);
            }
        }
        // lrelu + split of the hidden activations: K chunk kb = accumulator registers 8*kb .. 8*kb+7
        uint4 hs[2][2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float q = h[mb][8 * kb + e]; v[e] = q > 0.f ? q : q * P.slope; }
                k4s_split3(v, hs[mb][kb][0], hs[mb][kb][1], hs[mb][kb][2]);
            }
        // ---- GEMM 2 + modulation, 32 output channels at a time ----
#pragma unroll 1
        for (int mb2 = 0; mb2 < CB; ++mb2) {
            f32x16 cs, ch;
#pragma unroll
            for (int r = 0; r < 16; ++r) { cs[r] = bs[(mb2 * 2 + half) * 16 + r]; ch[r] = bh[(mb2 * 2 + half) * 16 + r]; }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint4* wps = ws6 + ((mb2 * 2 + kb) * 3) * 64 + lane;
                const uint4* wph = wh6 + ((mb2 * 2 + kb) * 3) * 64 + lane;
                const uint4 s0 = wps[0], s1 = wps[64], s2 = wps[128];
                const uint4 g0 = wph[0], g1 = wph[64], g2 = wph[128];
                K4_B6(cs, s0, s1, s2, hs[0][kb][0], hs[0][kb][1], hs[0][kb][2]);
                K4_B6(ch, g0, g1, g2, hs[1][kb][0], hs[1][kb][1], hs[1][kb][2]);
            }
            if (!pok) continue;
            // accumulator registers 4q..4q+3 are 4 CONSECUTIVE channels (co = 8q + 4*half + 0..3): 16-byte accesses
            const bool vec = P.vec4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = mb2 * 32 + 8 * q + 4 * half;
                float xv[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (vec) {
                    const float4 x4 = *reinterpret_cast<const float4*>(P.x + (size_t)pix * P.x_stride + co);
                    xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
                    if (P.res) {
                        const float4 r4 = *reinterpret_cast<const float4*>(P.res + (size_t)pix * P.res_stride + co);
                        rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xv[e] = P.x[(size_t)pix * P.x_stride + co + e];
                        if (P.res) rv[e] = P.res[(size_t)pix * P.res_stride + co + e];
                    }
                }
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = k4s_mul_add(xv[e], cs[4 * q + e] + 1.f, ch[4 * q + e]);               // x*(scale+1)+shift
                    if (P.res) v = k4s_mul_add(v, P.res_scale, rv[e]);
                    ov[e] = v;
                }
                if (vec) *reinterpret_cast<float4*>(P.y + (size_t)pix * P.y_stride + co) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) P.y[(size_t)pix * P.y_stride + co + e] = ov[e];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same layer, same arithmetic, same bits, with the memory round trips taken off the critical path.  The kernel above issues
// cond -> (wait) -> GEMM 1 / 2 -> x [, res] -> (wait) -> modulation -> store three times per 32-pixel tile at two waves per SIMD:
// 194 us per 64-channel layer of a 4K frame against an HBM floor of ~66 us.  Here one tile is ONE basic block (the second output
// block unrolled, the residual a template parameter, bounds by buffer descriptors: a pixel past the end loads zeros and its store is
// dropped, so there is no branch and hipcc counts its waits instead of draining the queue): the x (and residual) rows of the tile are
// requested before GEMM 1, the next tile's condition rows right after GEMM 1 consumed the current ones.  Needs 16-byte aligned rows and images below 2 GB (buffer offsets); anything else runs
// on the kernel above.
// ------------------------------------------------------------------------------------------------------------------
#define K4_SFT_OOB 0x80000000u
// Channel offsets go into the LANE offset (hipcc folds the constant into the instruction's immediate field), never into the scalar-offset
// operand: with an SGPR there, hipcc's hazard recogniser assumes a 16-byte store has read its data registers when the next instruction
// issues and lets a vector instruction overwrite them right behind the store -- on gfx950 lanes 12..15 of every 16 then stored the NEW
// values (second output block wrong in exactly those pixels; with the offset in the immediate field the recogniser inserts the wait state).
#define K4_SFT_CH(MB2, Q) ((unsigned)(((MB2) * 32 + 8 * (Q)) * 4))
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k4s_rsrc(const void* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a)),
        0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// OUT16: y is written PRE-SPLIT (k4_p16.h: fp16 hi / lo units under the tensor's scale 2^E = P.out_scale) for the 3x3 convolutions of
// k4_sr_p16.hip -- the lane arrangement (4 consecutive channels of one pixel, lanes l / l + 32 = the two halves of an 8-channel group) is the
// one p16_unit expects; a value beyond fp16 raises the window's overflow word.
template <int CB, bool RES, bool OUT16 = false>
__global__ __launch_bounds__(256) void k4_sft_b6p_kernel(const SftMulti M) {
    constexpr int NW6 = K4_SFT6_FLOATS(CB);
    constexpr int NW32 = (2 + 2 * CB) * 17 * 64;
    constexpr int NIT = (NW6 / 4 + 255) / 256;           // rounds of 256 x 16 bytes that fill the weight image
    __shared__ __attribute__((aligned(16))) float wl[NIT * 256 * 4];
    SftParams P = M.base;
    int blk = (int)blockIdx.x;
    int gjob = 0;
    {
        int g = 0;
        while (g + 1 < M.n && blk >= M.blk_end[g]) ++g;
        blk -= g ? M.blk_end[g - 1] : 0;
        P.cond = M.cond[g]; P.x = M.x[g]; P.y = M.y[g]; P.res = M.res[g]; P.n_pix = M.n_pix[g];
        gjob = g;
    }
    float amax = 0.f;
    const int lane = k4_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const __amdgpu_buffer_rsrc_t crs = k4s_rsrc(P.cond, P.n_pix * P.cond_stride * 4);
    const __amdgpu_buffer_rsrc_t xrs = k4s_rsrc(P.x, P.n_pix * P.x_stride * 4);
    const __amdgpu_buffer_rsrc_t yrs = k4s_rsrc(P.y, P.n_pix * P.y_stride * 4);
    const __amdgpu_buffer_rsrc_t rrs = k4s_rsrc(RES ? P.res : P.x, RES ? P.n_pix * P.res_stride * 4 : 0);
    const int base = (blk * 4 + wv) * 64;
    k4s_u32x4 cn[4];                                   // condition channels kb*16 + 8*half + 0..7 of the lane's pixel, kb = 0, 1
    {
        const int pix = base + l31;
        const unsigned coff = pix < P.n_pix ? (unsigned)(pix * P.cond_stride * 4 + half * 32) : K4_SFT_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) cn[i] = __builtin_amdgcn_raw_buffer_load_b128(crs, (int)(coff + (unsigned)((i >> 1) * 64 + (i & 1) * 16)), 0, 0);
    }
    {
        // the weight image goes global -> LDS directly (global_load_lds_dwordx4: lane l of a wave writes base + 16 l), every request of
        // the workgroup in flight at once and no staging registers: as a load / ds_write loop the fill was ~10 dependent L2 round trips,
        // as long as the two tiles that follow it.  The last round re-reads the image's first rows into the padding.
        const float4* src = reinterpret_cast<const float4*>(P.w + NW32);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = i * 256 + (int)threadIdx.x;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (idx < NW6 / 4 ? idx : idx - NW6 / 4)),
                                             (__attribute__((address_space(3))) void*)(wl + (i * 256 + wv * 64) * 4), 16, 0, 0);
        }
    }
    __syncthreads();
    const uint4* const wa6 = reinterpret_cast<const uint4*>(wl);
    const uint4* const ws6 = wa6 + 2 * 2 * 3 * 64;
    const uint4* const wh6 = ws6 + CB * 2 * 3 * 64;
    const float* const ba = wl + (2 + 2 * CB) * 2 * 3 * 64 * 4;
    const float* const bs = ba + 2 * 2 * 16;
    const float* const bh = bs + CB * 2 * 16;
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        asm volatile("" ::: "memory");                 // keeps the loop-invariant weight fragments in LDS (hoisted, they take 190 registers and a wave slot)
        const int pix = base + t * 32 + l31;
        const bool pok = pix < P.n_pix;
        const unsigned xoff = pok ? (unsigned)(pix * P.x_stride * 4 + half * 16) : K4_SFT_OOB;
        const unsigned yoff = pok ? (unsigned)(pix * P.y_stride * 4 + half * 16) : K4_SFT_OOB;
        const unsigned roff = (RES && pok) ? (unsigned)(pix * P.res_stride * 4 + half * 16) : K4_SFT_OOB;
        const unsigned cnext = (t == 0 && pix + 32 < P.n_pix) ? (unsigned)((pix + 32) * P.cond_stride * 4 + half * 32) : K4_SFT_OOB;
        k4s_u32x4 xq[4 * CB];                          // x channels mb2*32 + 8q + 4*half + 0..3
#pragma unroll
        for (int i = 0; i < 4 * CB; ++i) xq[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(xoff + K4_SFT_CH(i >> 2, i & 3)), 0, 0);
        k4s_u32x4 rq[RES ? 4 * CB : 1];
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < 4 * CB; ++i) rq[i] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(roff + K4_SFT_CH(i >> 2, i & 3)), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);             // the scheduler otherwise sinks the requests to their first use, behind the matrix work
        // ---- GEMM 1: hidden^T = lrelu(WA . cond^T + ba) ----
        f32x16 h[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[mb][r] = ba[(mb * 2 + half) * 16 + r];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const k4s_u32x4 c0 = cn[2 * kb], c1 = cn[2 * kb + 1];
            const float v[8] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w),
                                __uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w)};
            uint4 x0, x1, x2;
            k4s_split3(v, x0, x1, x2);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const uint4* wp = wa6 + ((mb * 2 + kb) * 3) * 64 + lane;
                const uint4 a0 = wp[0], a1 = wp[64], a2 = wp[128];
                K4_B6(h[mb], a0, a1, a2, x0, x1, x2);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cn[i] = __builtin_amdgcn_raw_buffer_load_b128(crs, (int)(cnext + (unsigned)((i >> 1) * 64 + (i & 1) * 16)), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // lrelu + split of the hidden activations: K chunk kb = accumulator registers 8*kb .. 8*kb+7
        uint4 hs[2][2][3];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float q = h[mb][8 * kb + e]; v[e] = q > 0.f ? q : q * P.slope; }
                k4s_split3(v, hs[mb][kb][0], hs[mb][kb][1], hs[mb][kb][2]);
            }
        // ---- GEMM 2 + modulation, 32 output channels at a time ----
#pragma unroll
        for (int mb2 = 0; mb2 < CB; ++mb2) {
            f32x16 cs, ch;
#pragma unroll
            for (int r = 0; r < 16; ++r) { cs[r] = bs[(mb2 * 2 + half) * 16 + r]; ch[r] = bh[(mb2 * 2 + half) * 16 + r]; }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint4* wps = ws6 + ((mb2 * 2 + kb) * 3) * 64 + lane;
                const uint4* wph = wh6 + ((mb2 * 2 + kb) * 3) * 64 + lane;
                const uint4 s0 = wps[0], s1 = wps[64], s2 = wps[128];
                const uint4 g0 = wph[0], g1 = wph[64], g2 = wph[128];
                K4_B6(cs, s0, s1, s2, hs[0][kb][0], hs[0][kb][1], hs[0][kb][2]);
                K4_B6(ch, g0, g1, g2, hs[1][kb][0], hs[1][kb][1], hs[1][kb][2]);
            }
            // accumulator registers 4q..4q+3 are 4 CONSECUTIVE channels (co = 8q + 4*half + 0..3): 16-byte accesses
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const k4s_u32x4 x4 = xq[mb2 * 4 + q];
                const float xv[4] = {__uint_as_float(x4.x), __uint_as_float(x4.y), __uint_as_float(x4.z), __uint_as_float(x4.w)};
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = k4s_mul_add(xv[e], cs[4 * q + e] + 1.f, ch[4 * q + e]);               // x*(scale+1)+shift
                if constexpr (RES) {
                    const k4s_u32x4 r4 = rq[mb2 * 4 + q];
                    const float rv[4] = {__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z), __uint_as_float(r4.w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = k4s_mul_add(ov[e], P.res_scale, rv[e]);
                }
                if constexpr (OUT16) {
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(ov[0]), fabsf(ov[1]))), fmaxf(fabsf(ov[2]), fabsf(ov[3])));
                    unsigned Hh[2], Ll[2];
                    p16_split4(ov, P.out_scale, Hh, Ll);
                    const p16_u32x4 unit = p16_unit(Hh, Ll);
                    const unsigned ypix = pok ? (unsigned)(pix * P.y_stride * 4) : K4_SFT_OOB;          // the unit offset is relative to the pixel's 32-channel block
                    __builtin_amdgcn_raw_buffer_store_b128(unit, yrs, (int)(ypix + (unsigned)(mb2 * 128) + P16_UNIT_OFF(q, half)), 0, 0);
                } else {
                    const k4s_u32x4 o4 = {__float_as_uint(ov[0]), __float_as_uint(ov[1]), __float_as_uint(ov[2]), __float_as_uint(ov[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(o4, yrs, (int)(yoff + K4_SFT_CH(mb2, q)), 0, 0);
                }
            }
        }
    }
    if constexpr (OUT16) {
        if (__builtin_amdgcn_ballot_w64(!(amax * P.out_scale <= 65504.f)) != 0ull && lane == 0) atomicOr(P.overflow + gjob, 1u);
    }
}
#undef K4_SFT_OOB
#undef K4_SFT_CH

extern "C" int64_t k4_sft_weight_floats(int32_t channels) {
    if (channels != 32 && channels != 64) return -1;
    const int cb = channels / 32;
    return (int64_t)(2 + 2 * cb) * 17 * 64 + (cb == 2 ? K4_SFT6_FLOATS(2) : K4_SFT6_FLOATS(1));
}

static int sft_multi(const k4_sft_job* jobs, int32_t n_jobs, int32_t cond_stride, const float* w_packed, int32_t x_stride,
                     int32_t y_stride, int32_t channels, float slope, int32_t res_stride, float res_scale, int32_t arith, void* stream,
                     float out_scale = 0.f, uint32_t* overflow = nullptr) {
    if (arith != K4_SFT_ARITH_FP32 && arith != K4_SFT_ARITH_BF16X6) return K4_ERR_BAD_ARG;
    if (!jobs || n_jobs <= 0 || n_jobs > K4_MAX_JOBS || !w_packed || cond_stride < 32 || (cond_stride & 3)) return K4_ERR_BAD_ARG;
    if ((channels != 32 && channels != 64) || x_stride < channels || y_stride < channels) return K4_ERR_BAD_ARG;
    SftMulti M{};
    SftParams& P = M.base;
    P.cond_stride = cond_stride; P.w = w_packed; P.x_stride = x_stride; P.y_stride = y_stride;
    P.res_stride = res_stride; P.res_scale = res_scale; P.slope = slope;
    P.out_scale = out_scale; P.overflow = overflow;
    M.n = n_jobs;
    int total = 0;
    bool any_res = false, all_res = true;
    size_t align = 0;
    long long max_bytes = 0;
    int max_stride = cond_stride > x_stride ? cond_stride : x_stride;
    if (y_stride > max_stride) max_stride = y_stride;
    if (res_stride > max_stride) max_stride = res_stride;
    for (int g = 0; g < n_jobs; ++g) {
        const k4_sft_job& j = jobs[g];
        if ((long long)j.n_pix * max_stride * 4 > max_bytes) max_bytes = (long long)j.n_pix * max_stride * 4;
        if (!j.cond || !j.x || !j.y || j.n_pix <= 0 || j.n_pix > 0x7fffffff || (((size_t)j.cond) & 15) != 0) return K4_ERR_BAD_ARG;
        any_res |= j.res != nullptr; all_res &= j.res != nullptr;
        align |= (size_t)j.x | (size_t)j.y | (size_t)(j.res ? j.res : j.x);
        M.cond[g] = j.cond; M.x[g] = j.x; M.y[g] = j.y; M.res[g] = j.res; M.n_pix[g] = (int)j.n_pix;
        total += (int)((j.n_pix + 255) / 256);
        M.blk_end[g] = total;
    }
    if (any_res != all_res || (any_res && res_stride < channels)) return K4_ERR_BAD_ARG;
    P.res = any_res ? jobs[0].res : nullptr;                         // the kernel tests P.res for "has residual"; the job's pointer is used
    P.vec4 = ((x_stride | y_stride | (any_res ? res_stride : 0)) & 3) == 0 && (align & 15) == 0;
    const dim3 grid((unsigned)total), block(256);
    if (arith == K4_SFT_ARITH_BF16X6) {
        // the pipelined form addresses through 32-bit buffer offsets: 16-byte rows, every image below 2 GB (K4_SR_DEBUG bit 5: A/B)
        const bool piped = P.vec4 && max_bytes < (1ll << 31) && !(k4_env().sr_debug & 32);
        if (out_scale != 0.f) {                                      // p16 output: the pipelined kernel only (aligned rows, images below 2 GB), no residual
            if (!piped || any_res || (y_stride & 15)) return K4_ERR_UNSUPPORTED;
            for (int g = 0; g < n_jobs; ++g) if (((size_t)jobs[g].y) & 63) return K4_ERR_BAD_ARG;       // the slice starts on a 16-channel chunk
            if (channels == 64) hipLaunchKernelGGL((k4_sft_b6p_kernel<2, false, true>), grid, block, 0, (hipStream_t)stream, M);
            else hipLaunchKernelGGL((k4_sft_b6p_kernel<1, false, true>), grid, block, 0, (hipStream_t)stream, M);
        } else if (piped && channels == 64) {
            if (any_res) hipLaunchKernelGGL((k4_sft_b6p_kernel<2, true>), grid, block, 0, (hipStream_t)stream, M);
            else hipLaunchKernelGGL((k4_sft_b6p_kernel<2, false>), grid, block, 0, (hipStream_t)stream, M);
        } else if (piped) {
            if (any_res) hipLaunchKernelGGL((k4_sft_b6p_kernel<1, true>), grid, block, 0, (hipStream_t)stream, M);
            else hipLaunchKernelGGL((k4_sft_b6p_kernel<1, false>), grid, block, 0, (hipStream_t)stream, M);
        } else if (channels == 64) hipLaunchKernelGGL((k4_sft_b6_kernel<2>), grid, block, 0, (hipStream_t)stream, M);
        else hipLaunchKernelGGL((k4_sft_b6_kernel<1>), grid, block, 0, (hipStream_t)stream, M);
    } else {
        if (channels == 64) hipLaunchKernelGGL((k4_sft_kernel<2>), grid, block, 0, (hipStream_t)stream, M);
        else hipLaunchKernelGGL((k4_sft_kernel<1>), grid, block, 0, (hipStream_t)stream, M);
    }
    return k4_check_launch();
}

// SFTLayer with PRE-SPLIT output (K4_SFT_ARITH_BF16X6 arithmetic; no residual): the producer side of k4_conv3x3_p16_multi.
extern "C" int k4_sft_nhwc_p16_multi(const k4_sft_job* jobs, int32_t n_jobs, int32_t cond_stride, const float* w_packed, int32_t x_stride,
                                     int32_t y_stride, int32_t channels, float slope, float out_scale, uint32_t* overflow, void* stream) {
    if (!(out_scale > 0.f) || !overflow) return K4_ERR_BAD_ARG;
    return sft_multi(jobs, n_jobs, cond_stride, w_packed, x_stride, y_stride, channels, slope, 0, 0.f, K4_SFT_ARITH_BF16X6, stream, out_scale, overflow);
}

extern "C" int k4_sft_nhwc_multi(const k4_sft_job* jobs, int32_t n_jobs, int32_t cond_stride, const float* w_packed, int32_t x_stride,
                                 int32_t y_stride, int32_t channels, float slope, int32_t res_stride, float res_scale, int32_t arith,
                                 void* stream) {
    return sft_multi(jobs, n_jobs, cond_stride, w_packed, x_stride, y_stride, channels, slope, res_stride, res_scale, arith, stream);
}

extern "C" int k4_sft_nhwc(const float* cond, int32_t cond_stride, const float* w_packed,
                           const float* x, int32_t x_stride, float* y, int32_t y_stride, int32_t channels,
                           int64_t n_pix, float slope, const float* res, int32_t res_stride, float res_scale, void* stream) {
    k4_sft_job j{};
    j.cond = cond; j.x = x; j.y = y; j.res = res; j.n_pix = n_pix;
    return sft_multi(&j, 1, cond_stride, w_packed, x_stride, y_stride, channels, slope, res ? res_stride : 0, res_scale, K4_SFT_ARITH_FP32, stream);
}
