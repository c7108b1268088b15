// VC-Decoder 3x3 convolutions on PRE-SPLIT activations ("p16", the decoder's 'f16x3p' arithmetic; lib/sr_esrnet.py:126-182,446-465).
//
// The f16x3 kernel of k4_sr.hip reads fp32 activations and splits every value into fp16 hi + lo while it stages a chunk (load -> chunk
// maximum -> barrier -> scale / split -> ds_write): a dense-block activation is split again by each of up to five consuming layers, and the
// staging's vector work sits between two barriers where it cannot hide under matrix work (profiles/r04_mfma_valu_overlap.md: a wave hides
// <= 5 vector instructions per MFMA only when they are spread through the MFMA stream).  Here the PRODUCER of a tensor (this kernel's
// epilogue, the SFT kernel) writes it split, once, under one power-of-two scale per tensor:
//
//   p16 tensor = a channel slice (offset and width multiples of 16) of an NHWC image of 4-byte elements; per pixel and 16-channel chunk
//   (64 bytes) four 16-byte units: [hi ch 0-7][hi ch 8-15][lo ch 0-7][lo ch 8-15], hi = RNE_fp16(v 2^E), lo = RNE_fp16(v 2^E - hi)
//   (22 significant bits while |v 2^E| >= 2^-3; an absolute error <= 2^-25 2^-E below).  E is fixed per tensor BEFORE it is written
//   (calibration, SFTNet._k4_calibrate): the epilogue raises the window's overflow word when |v| 2^E > 65504 and the host re-runs that window
//   on the per-tile kernel.  A unit IS the MFMA operand fragment of (term, 8-channel group): the consumer moves chunks global -> LDS with
//   buffer_load ... lds (no register, no vector instruction, no ds_write), double buffered, one barrier per chunk.
//
// Consumer (this kernel): workgroup = 4 waves = 8 rows x 32 columns x 32 output channels; the GEMM is TRANSPOSED against k4_sr.hip
// (A = weights, B = activations: D[co][pixel]), so a lane ends with ONE pixel and 4 x 4 consecutive output channels: 16-byte stores
// / residual loads (the b32 epilogue of the other orientation was a third of a trunk layer's time) and the exact lane arrangement of the
// SFT kernel, so both producers share the p16 store.  The weights carry every scale: w 2^a[co] 2^-E[chunk's tensor] (host packing), so
// the accumulators are 2^a[co] x the convolution whatever the scales of the input slices are: no exponent arithmetic in the kernel.
// LDS image of a chunk's haloed tile: [pixel][4 slots of 16 bytes], slot = unit ^ ((column >> 2) & 3) -- the swizzle is applied on the
// SOURCE address of the DMA (its LDS side is lane-linear), and makes the fragment reads (32 pixels at a 64-byte stride) conflict-free.
#include "k4_p16.h"

#define P16_COLS 34
#define P16_ROWS 10
#define P16_NPIX (P16_ROWS * P16_COLS)              /* 340 */
#define P16_ACT_BYTES (P16_NPIX * 64)               /* 21760 */
#define P16_ACT_ITEMS (P16_NPIX * 4)                /* 1360 units of 16 bytes */
#define P16_ACT_INSTR ((P16_ACT_ITEMS + 63) / 64)   /* 22 wave-instructions of 1 KB (the last one 16 lanes) */
#define P16_W_BYTES (2 * 9 * 2 * 32 * 16)           /* 18432: [term][tap][channel group][32 co] x 8 fp16 */
#define P16_W_INSTR (P16_W_BYTES / 1024)            /* 18 */
#define P16_W_BYTES_UP (2 * 4 * 2 * 32 * 16)        /* 8192: [term][2 x 2 tap][channel group][32 co] x 8 fp16, per phase */

struct P16Conv {
    int n;
    int blk_end[K4_MAX_JOBS];
    const void* x[K4_MAX_JOBS]; void* y[K4_MAX_JOBS]; const float* res[K4_MAX_JOBS];
    int H[K4_MAX_JOBS], W[K4_MAX_JOBS], tiles_x[K4_MAX_JOBS];
    int total;
    int cin, cin_stride;
    const void* w; const float* bias;
    int cout, cout_stride; unsigned flags; float slope; int res_stride; float res_scale;
    float out_scale;
    uint32_t* overflow;
    int debug;               // K4_SR_DEBUG ablation bits (profiling only, WRONG results; 0 in production): 64 = activation DMA for chunk 0 only, 128 = weight DMA for
                             // chunk 0 only, 256 = no MFMAs, 512 = every pixel reads pixel 0 (no HBM traffic)
};

// UP (K4_PRE_UPSAMPLE2X): the layer reads its input through a nearest x2 upsampling (lib/sr_esrnet.py:461-463).  Output pixel (2Y + py, 2X + px)
// then sees only a 2 x 2 neighbourhood of LR pixels -- the 3 x 3 taps that land on the same LR pixel add up: per PHASE (py, px) the layer is a
// 2 x 2 convolution with the weights W'[a][b] = sum of the taps (dy, dx) with (py + dy - 1) >> 1 == a + py - 1 (rows; same for columns), 16
// tap matrices instead of 36 per 2 x 2 output pixels: 2.25x fewer matrix instructions, exact algebra (the tap sums are formed in fp32 by the
// packer: the products differ from the 9-tap form by one rounding of a weight sum).  A workgroup = 8 x 32 LR positions of ONE phase; the four
// phase workgroups of a tile are adjacent in launch order (the tile's activations come from L2 three times out of four).
template <bool OUT16, bool UP>
__global__ __launch_bounds__(256, 2) void k4_conv_p16_kernel(const P16Conv M) {
    constexpr int W_CH = UP ? P16_W_BYTES_UP : P16_W_BYTES;           // weight bytes of one (chunk, output block[, phase])
    constexpr int W_NI = W_CH / 1024;                                  // DMA instructions
    constexpr int NSUB = UP ? 8 : 18;                                  // sub-stages (tap x row) per chunk
    constexpr int NTAP = UP ? 4 : 9;
    __shared__ __attribute__((aligned(16))) unsigned char wbuf0[P16_W_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf1[P16_W_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char abuf0[P16_ACT_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char abuf1[P16_ACT_BYTES];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform for the compiler too: DMA instruction indices, M0, scalar offsets
    const int b = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (b >= M.total) return;
    int g = 0;
    while (g + 1 < M.n && b >= M.blk_end[g]) ++g;
    const int local = b - (g ? M.blk_end[g - 1] : 0);
    const int nb_count = M.cout >> 5;
    const int phase = UP ? (local & 3) : 0, py = phase >> 1, px = phase & 1;
    const int lt = UP ? (local >> 2) : local;
    const int tile = lt / nb_count, nb = lt - tile * nb_count;
    const int H = M.H[g], W = M.W[g];                                  // OUTPUT size; the tile grid and the DMA plan live on the input (LR) image
    const int srcH = UP ? H >> 1 : H, srcW = UP ? W >> 1 : W;
    const int tiles_x = M.tiles_x[g];
    const int x0 = (tile % tiles_x) * 32, y0 = (tile / tiles_x) * 8;   // input-image coordinates of the tile
    const int nchunks = M.cin >> 4;

    // ---- DMA plan of this thread (chunk independent): activation instructions k = wv + 4 i, weight instructions k = wv + 4 i ----
    const __amdgpu_buffer_rsrc_t xrs = p16_rsrc(M.x[g], (unsigned)(((long long)(srcH * srcW - 1) * M.cin_stride + M.cin) * 4));
    const __amdgpu_buffer_rsrc_t wrs = p16_rsrc(M.w, (unsigned)(nchunks * nb_count * (UP ? 4 : 1) * W_CH));
    unsigned aoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int item = (wv + 4 * i) * 64 + lane;
        const int p = item >> 2, j = item & 3;
        const int row = p / P16_COLS, col = p - row * P16_COLS;
        const int gy = y0 - 1 + row, gx = x0 - 1 + col;
        const bool inside = item < P16_ACT_ITEMS && gy >= 0 && gy < srcH && gx >= 0 && gx < srcW;
        aoff[i] = inside ? (unsigned)(((M.debug & 512) ? 0 : (gy * srcW + gx) * M.cin_stride * 4) + ((j ^ ((col >> 2) & 3)) << 4)) : P16_OOB;
    }
    const unsigned woff = (unsigned)(lane * 16);
#define P16_ISSUE(CH, WB, AB) do { \
        const int wso_ = (((CH) * nb_count + nb) * (UP ? 4 : 1) + phase) * W_CH; \
        const int aso_ = (CH) * 64; \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) { \
            const int k_ = wv + 4 * i_; \
            if (k_ < W_NI && (!(M.debug & 128) || (CH) == 0)) \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(WB + k_ * 1024), 16, (int)woff, wso_ + k_ * 1024, 0, 0); \
        } \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) { \
            const int k_ = wv + 4 * i_; \
            if ((k_ < P16_ACT_INSTR - 1 || (k_ == P16_ACT_INSTR - 1 && lane < P16_ACT_ITEMS - (P16_ACT_INSTR - 1) * 64)) && (!(M.debug & 64) || (CH) == 0)) \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(AB + k_ * 1024), 16, (int)aoff[i_], aso_, 0, 0); \
        } } while (0)

    // ---- fragment addresses: weights (A operand) lane = (channel group, co); activations (B operand) lane = (channel group, column) ----
    const unsigned wrd = (unsigned)(lane * 16);
    unsigned ard[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = l31 + dx + (UP && dx < 2 ? px : 0);           // UP: column taps b = 0, 1 read haloed columns l31 + b + px (ard[2] unused)
            ard[dx][t] = (unsigned)(((wv * 2 + (UP ? py : 0)) * P16_COLS + col) * 64 + (((2 * t + half) ^ ((col >> 2) & 3)) << 4));
        }

    // per-lane epilogue tables: output channels co(q, e) = nb*32 + 8q + 4 half + e
    const int cob = nb * 32 + 4 * half;
    p16_f32x4 us[4], bs[4];
    {
        const float* const wtail = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(M.w) + (size_t)nchunks * nb_count * (UP ? 4 : 1) * W_CH);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            us[q] = *reinterpret_cast<const p16_f32x4*>(wtail + cob + 8 * q);
            bs[q] = *reinterpret_cast<const p16_f32x4*>(M.bias + cob + 8 * q);
        }
    }

    p16_f32x16 acc[2];
    acc[0] = (p16_f32x16)(0.f); acc[1] = (p16_f32x16)(0.f);

    P16_ISSUE(0, wbuf0, abuf0);

    // tap t -> (row, column) offset in the haloed tile: 3 x 3 taps (dy, dx) = (t / 3, t % 3); UP: 2 x 2 taps (a, b) = (t / 2, t % 2), the phase's
    // offset already sits in ard
#define P16_TROW(T) (UP ? (T) / 2 : (T) / 3)
#define P16_TCOL(T) (UP ? (T) % 2 : (T) % 3)
#define P16_RDW(DST, WB, T) do { \
        DST[0] = *reinterpret_cast<const p16_u32x4*>(WB + wrd + (0 * NTAP + (T)) * 1024); \
        DST[1] = *reinterpret_cast<const p16_u32x4*>(WB + wrd + (1 * NTAP + (T)) * 1024); } while (0)
#define P16_RDX(DST, AB, U) do { \
        const int t_ = (U) >> 1, r_ = (U) & 1; \
        DST[0] = *reinterpret_cast<const p16_u32x4*>(AB + ard[P16_TCOL(t_)][0] + (r_ + P16_TROW(t_)) * (P16_COLS * 64)); \
        DST[1] = *reinterpret_cast<const p16_u32x4*>(AB + ard[P16_TCOL(t_)][1] + (r_ + P16_TROW(t_)) * (P16_COLS * 64)); } while (0)
    // one chunk: its DMA has been issued an iteration ago; wait, barrier, issue the next chunk's DMA into the other buffers, 18 sub-stages
    // (tap x row) of 3 MFMAs with the fragments of sub-stage u + 2 / tap t + 1 read under the MFMAs of u
#define P16_CHUNK(CH, WB, AB, WBN, ABN) do { \
        __syncthreads(); \
        if ((CH) + 1 < nchunks) P16_ISSUE((CH) + 1, WBN, ABN); \
        p16_u32x4 wa[2][2], xb[3][2]; \
        P16_RDW(wa[0], WB, 0); \
        P16_RDX(xb[0], AB, 0); \
        P16_RDX(xb[1], AB, 1); \
        if (!(M.debug & 256)) \
        _Pragma("unroll") for (int u = 0; u < NSUB; ++u) { \
            const int t = u >> 1, r = u & 1; \
            if (r == 0 && t + 1 < NTAP) P16_RDW(wa[(t + 1) & 1], WB, t + 1); \
            if (u + 2 < NSUB) P16_RDX(xb[(u + 2) % 3], AB, u + 2); \
            __builtin_amdgcn_sched_barrier(0); \
            const p16_f16x8 wh = __builtin_bit_cast(p16_f16x8, wa[t & 1][0]), wl = __builtin_bit_cast(p16_f16x8, wa[t & 1][1]); \
            const p16_f16x8 xh = __builtin_bit_cast(p16_f16x8, xb[u % 3][0]), xl = __builtin_bit_cast(p16_f16x8, xb[u % 3][1]); \
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[r], 0, 0, 0); \
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[r], 0, 0, 0); \
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[r], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); \
        } } while (0)

    for (int ch = 0; ch < nchunks; ch += 2) {
        P16_CHUNK(ch, wbuf0, abuf0, wbuf1, abuf1);
        if (ch + 1 < nchunks) P16_CHUNK(ch + 1, wbuf1, abuf1, wbuf0, abuf0);
    }
#undef P16_CHUNK
#undef P16_RDX
#undef P16_RDW
#undef P16_TROW
#undef P16_TCOL
#undef P16_ISSUE

    // ---- epilogue: lane = pixel (x0 + l31, row y0 + 2 wv + r), registers 4q .. 4q+3 = channels cob + 8q + 0..3 ----
    const int gx = UP ? 2 * (x0 + l31) + px : x0 + l31;               // output column of this lane
    const float sl = (M.flags & K4_EPI_LRELU) ? M.slope : 1.f;
    const bool has_res = (M.flags & K4_EPI_RES) != 0;
    const __amdgpu_buffer_rsrc_t yrs = p16_rsrc(M.y[g], (unsigned)(((long long)(H * W - 1) * M.cout_stride + M.cout) * 4));
    const __amdgpu_buffer_rsrc_t rrs = p16_rsrc(has_res ? (const void*)M.res[g] : M.y[g], has_res ? (unsigned)(((long long)(H * W - 1) * M.res_stride + M.cout) * 4) : 0u);
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int gy = UP ? 2 * (y0 + wv * 2 + r) + py : y0 + wv * 2 + r;
        if (gy >= H) continue;                                               // wave-uniform
        const bool ok = gx < W;
        const unsigned pix = (unsigned)(gy * W + gx);
        const unsigned roff = ok ? (pix * (unsigned)M.res_stride + (unsigned)cob) * 4u : P16_OOB;
        p16_u32x4 rq[4];
        if (has_res) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rq[q] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(roff + (unsigned)(q * 32)), 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = fmaf(acc[r][4 * q + e], us[q][e], bs[q][e]);
                t = fmaxf(t, t * sl);                                        // LeakyReLU for 0 <= slope <= 1 (host check); identity with sl = 1
                if (has_res) t = p16_mul_add(t, M.res_scale, __uint_as_float(rq[q][e]));
                v[e] = t;
            }
            if constexpr (OUT16) {
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                unsigned Hh[2], Ll[2];
                p16_split4(v, M.out_scale, Hh, Ll);
                const p16_u32x4 unit = p16_unit(Hh, Ll);
                const unsigned yoff = ok ? (pix * (unsigned)M.cout_stride + (unsigned)(nb * 32)) * 4u + P16_UNIT_OFF(q, half) : P16_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(unit, yrs, (int)yoff, 0, 0);
            } else {
                // (constant channel offsets ride in the lane offset -> immediate field, never in the scalar offset: k4_sr.hip, K4_SFT_CH)
                const unsigned yoff = ok ? (pix * (unsigned)M.cout_stride + (unsigned)cob) * 4u + (unsigned)(q * 32) : P16_OOB;
                const p16_u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)yoff, 0, 0);
            }
        }
    }
    if constexpr (OUT16) {
        // |v| 2^E beyond fp16 (or non-finite): this window's frame is redone on the per-tile kernel (host).  Values of masked lanes (columns past
        // the image) come from zero-padded inputs like any other pixel's: harmless to include.
        if (__builtin_amdgcn_ballot_w64(!(amax * M.out_scale <= 65504.f)) != 0ull && lane == 0) atomicOr(M.overflow + g, 1u);
    }
}

extern "C" int64_t k4_conv_weight_p16_bytes(int32_t cout, int32_t cin) {
    if (cout <= 0 || cin <= 0 || (cout & 31) || (cin & 15)) return -1;
    return (int64_t)(cin / 16) * (cout / 32) * P16_W_BYTES + (int64_t)cout * 4;
}
extern "C" int64_t k4_conv_weight_p16_up2x_bytes(int32_t cout, int32_t cin) {       // the K4_PRE_UPSAMPLE2X operand: four phases of 2 x 2 taps
    if (cout <= 0 || cin <= 0 || (cout & 31) || (cin & 15)) return -1;
    return (int64_t)(cin / 16) * (cout / 32) * 4 * P16_W_BYTES_UP + (int64_t)cout * 4;
}

extern "C" int k4_conv3x3_p16_multi(const k4_conv_job* jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                                    const void* w_p16, const float* bias, int32_t cout, int32_t cout_stride,
                                    uint32_t flags, float slope, int32_t res_stride, float res_scale,
                                    float out_scale, uint32_t* overflow, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > K4_MAX_JOBS || !w_p16 || !bias) return K4_ERR_BAD_ARG;
    if (cin <= 0 || (cin & 15) || cin_stride < cin || (cin_stride & 3) || cout <= 0 || (cout & 31) || cout_stride < cout || (cout_stride & 3)) return K4_ERR_BAD_ARG;
    if (flags & ~(K4_EPI_LRELU | K4_EPI_RES | K4_PRE_UPSAMPLE2X)) return K4_ERR_BAD_ARG;
    if ((flags & K4_EPI_LRELU) && !(slope >= 0.f && slope <= 1.f)) return K4_ERR_BAD_ARG;
    if ((flags & K4_EPI_RES) && (res_stride < cout || (res_stride & 3))) return K4_ERR_BAD_ARG;
    if (out_scale != 0.f && (!overflow || !(out_scale > 0.f))) return K4_ERR_BAD_ARG;
    if ((((size_t)w_p16) & 15) || (((size_t)bias) & 15)) return K4_ERR_BAD_ARG;
    P16Conv M{};
    M.n = n_jobs; M.cin = cin; M.cin_stride = cin_stride; M.w = w_p16; M.bias = bias; M.cout = cout; M.cout_stride = cout_stride;
    M.flags = flags; M.slope = slope; M.res_stride = res_stride; M.res_scale = res_scale; M.out_scale = out_scale; M.overflow = overflow; M.debug = k4_env().sr_debug;
    const int nbc = cout / 32;
    int total = 0;
    for (int g = 0; g < n_jobs; ++g) {
        const k4_conv_job& j = jobs[g];
        if (!j.x || !j.y || j.H <= 0 || j.W <= 0 || (((size_t)j.x | (size_t)j.y) & 15)) return K4_ERR_BAD_ARG;
        if ((flags & K4_EPI_RES) && (!j.res || (((size_t)j.res) & 15))) return K4_ERR_BAD_ARG;
        if ((flags & K4_PRE_UPSAMPLE2X) && ((j.H & 1) || (j.W & 1))) return K4_ERR_BAD_ARG;
        const long long strd = cin_stride > cout_stride ? cin_stride : cout_stride;
        if ((long long)j.H * j.W * (strd > res_stride ? strd : res_stride) * 4 >= 0x80000000LL) return K4_ERR_UNSUPPORTED;      // 32-bit buffer offsets
        M.x[g] = j.x; M.y[g] = j.y; M.res[g] = j.res; M.H[g] = j.H; M.W[g] = j.W;
        const bool up = (flags & K4_PRE_UPSAMPLE2X) != 0;              // the tile grid lives on the INPUT image; four phase workgroups per tile
        const int gw = up ? j.W / 2 : j.W, gh = up ? j.H / 2 : j.H;
        M.tiles_x[g] = (gw + 31) / 32;
        total += M.tiles_x[g] * ((gh + 7) / 8) * nbc * (up ? 4 : 1);
        M.blk_end[g] = total;
    }
    M.total = total;
    const dim3 grid((unsigned)total), block(256);
    if (flags & K4_PRE_UPSAMPLE2X) {
        if (out_scale != 0.f) hipLaunchKernelGGL((k4_conv_p16_kernel<true, true>), grid, block, 0, (hipStream_t)stream, M);
        else hipLaunchKernelGGL((k4_conv_p16_kernel<false, true>), grid, block, 0, (hipStream_t)stream, M);
    } else if (out_scale != 0.f) hipLaunchKernelGGL((k4_conv_p16_kernel<true, false>), grid, block, 0, (hipStream_t)stream, M);
    else hipLaunchKernelGGL((k4_conv_p16_kernel<false, false>), grid, block, 0, (hipStream_t)stream, M);
    return k4_check_launch();
}

// ---- largest |x| of a channel slice, as bits, into *out_bits with atomicMax (calibration of the p16 exponents; non-finite values count) ----
__global__ __launch_bounds__(256) void k4_absmax_slice_kernel(const float* x, long long n_pix, int stride, int channels, uint32_t* out_bits) {
    unsigned m = 0u;
    const long long total = n_pix * channels;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / channels;
        const int c = (int)(i - p * channels);
        const unsigned bits = __float_as_uint(x[p * stride + c]) & 0x7fffffffu;
        m = bits > m ? bits : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_bits, m);
}

extern "C" int k4_absmax_slice(const float* x, int64_t n_pix, int32_t stride, int32_t channels, uint32_t* out_bits, void* stream) {
    if (!x || !out_bits || n_pix <= 0 || channels <= 0 || stride < channels) return K4_ERR_BAD_ARG;
    const long long total = (long long)n_pix * channels;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k4_absmax_slice_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n_pix, stride, channels, out_bits);
    return k4_check_launch();
}
