// rgbnet device functions of the shading kernel (k4_shade_kernel, csrc/k4_march.hip): operand layouts, the bf16 splits, the MLP on the matrix cores in its
// three arithmetics (fp32-input MFMA, exact 3-term bf16 splits, 2-term splits = the round-6 default) and in its two schedules (one 32-sample tile at a
// time; both tiles of a batch in one software pipeline), the trilinear corner setup.  Included by k4_march.hip AFTER MarchParams, f32x16 and the K4_TSTAMP
// macros are defined -- a textual part of that translation unit, split off for readability only (lib/dmpigo.py:336-379, lib/dvgo.py:372-412).
// Packed MLP buffer (host: 4k-nerf_amd/lib/dvgo.py::_pack_mlp_mfma), all in MFMA operand order; NB = W/32,
// K1P = dim0+1 rounded up to even (the extra input is the constant 1 that carries the bias):
//   W1A [NB][K1P/2][64]      lane l: W1ext[j = mb*32+(l&31)][k = 2*kk+(l>>5)],  W1ext = [W1 | b1 | 0]
//   if NHID: W2A [NB][NB][16][64]  lane l: W2[j2 = mb2*32+(l&31)][k = mb*32 + row(r,l>>5)]
//            B2A [NB][64]          lane l: l<32 ? b2[mb2*32+l] : 0
//   WOT [NB][16][2][4]       Wout[c][mb*32 + row(r,half)], c padded to 4
//   BO  [4]
// with row(r,half) = (r&3) + 8*(r>>2) + 4*half: the C/D register->row map of v_mfma_f32_32x32x2_f32.
template <int WIDTH, int NHID>
struct MlpLayout {
    static constexpr int NB = WIDTH / 32;
    __device__ static int w1a(int k1p) { (void)k1p; return 0; }
    __device__ static int w2a(int k1p) { return NB * (k1p / 2) * 64; }
    __device__ static int b2a(int k1p) { return w2a(k1p) + (NHID ? NB * NB * 16 * 64 : 0); }
    __device__ static int wot(int k1p) { return b2a(k1p) + (NHID ? NB * 64 : 0); }
    __device__ static int bo(int k1p) { return wot(k1p) + NB * 16 * 2 * 4; }
    __device__ static int total(int k1p) { return bo(k1p) + 4; }
};

// rgbnet on the matrix cores: logits[3] of the sample owned by this lane, features read from LDS feat[K1P][64].
template <int W, int NHID>
__device__ __forceinline__ void mlp_mfma(const float* wl, const float* feat, int k1p_, int lane, int half, int debug_,
                                         float& out0, float& out1, float& out2) {
    constexpr int NB = W / 32;
    typedef MlpLayout<W, NHID> ML;
    const struct { int k1p; int debug; } P = {k1p_, debug_};
    // ---------------- layer 1: H1^T[j][s] = sum_k W1ext[j][k] * X[k][s]  ----------------
    f32x16 h1[NB][2];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) { h1[mb][0] = (f32x16)(0.f); h1[mb][1] = (f32x16)(0.f); }
    const float* const w1a = wl + ML::w1a(P.k1p);
    const int ksteps = (P.debug & 2) ? 0 : (P.k1p >> 1);
    for (int kk = 0; kk < ksteps; ++kk) {
        const float b0 = feat[(2 * kk + half) * 64 + (lane & 31)];
        const float b1 = feat[(2 * kk + half) * 64 + 32 + (lane & 31)];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            const float a = w1a[(mb * ksteps + kk) * 64 + lane];
            h1[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, h1[mb][0], 0, 0, 0);
            h1[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, h1[mb][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { h1[mb][0][r] = fmaxf(h1[mb][0][r], 0.f); h1[mb][1][r] = fmaxf(h1[mb][1][r], 0.f); }

    // partial output sums of the two 32-sample tiles (this lane: neurons row(r,half) of each block)
    float pt[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const float* const wot = wl + ML::wot(P.k1p);
    if (NHID == 1 && !(P.debug & 2)) {
        const float* const w2a = wl + ML::w2a(P.k1p);
        const float* const b2a = wl + ML::b2a(P.k1p);
#pragma unroll 1
        for (int mb2 = 0; mb2 < NB; ++mb2) {
            f32x16 c0 = (f32x16)(0.f), c1 = (f32x16)(0.f);
            {   // bias: k-step with A = [b2 | 0], B = 1
                const float a = b2a[mb2 * 64 + lane];
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.f, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.f, c1, 0, 0, 0);
            }
#pragma unroll
            for (int mb = 0; mb < NB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = w2a[((mb2 * NB + mb) * 16 + r) * 64 + lane];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h1[mb][0][r], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h1[mb][1][r], c1, 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 wo = *reinterpret_cast<const float4*>(wot + ((mb2 * 16 + r) * 2 + half) * 4);
                const float a0 = fmaxf(c0[r], 0.f), a1 = fmaxf(c1[r], 0.f);
                pt[0][0] = fmaf(wo.x, a0, pt[0][0]); pt[0][1] = fmaf(wo.y, a0, pt[0][1]); pt[0][2] = fmaf(wo.z, a0, pt[0][2]);
                pt[1][0] = fmaf(wo.x, a1, pt[1][0]); pt[1][1] = fmaf(wo.y, a1, pt[1][1]); pt[1][2] = fmaf(wo.z, a1, pt[1][2]);
            }
        }
    } else {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 wo = *reinterpret_cast<const float4*>(wot + ((mb * 16 + r) * 2 + half) * 4);
                const float a0 = h1[mb][0][r], a1 = h1[mb][1][r];
                pt[0][0] = fmaf(wo.x, a0, pt[0][0]); pt[0][1] = fmaf(wo.y, a0, pt[0][1]); pt[0][2] = fmaf(wo.z, a0, pt[0][2]);
                pt[1][0] = fmaf(wo.x, a1, pt[1][0]); pt[1][1] = fmaf(wo.y, a1, pt[1][1]); pt[1][2] = fmaf(wo.z, a1, pt[1][2]);
            }
    }
    // lanes l and l^32 hold the two halves of the neurons of sample (l&31) of each tile
    const float* const bo = wl + ML::bo(P.k1p);
    float q[2][3];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int c = 0; c < 3; ++c) q[tt][c] = pt[tt][c] + __shfl_xor(pt[tt][c], 32) + bo[c];
    out0 = half ? q[1][0] : q[0][0];
    out1 = half ? q[1][1] : q[0][1];
    out2 = half ? q[1][2] : q[0][2];
}

// ---------------------------------------------------------------------------------------------------------------------
// Same MLP on the bf16 matrix pipe with fp32-equivalent accuracy (default for width <= 64).  gfx950 has no fast fp32
// MFMA (v_mfma_f32_32x32x2_f32: 256 flop/clk/CU-SIMD; v_mfma_f32_32x32x16_bf16: 1024), and the fp32 form made this kernel
// matrix-pipe bound.  Every fp32 value v is split EXACTLY into three bf16 terms v = v0 + v1 + v2 (v0 = RNE_bf16(v),
// v1 = RNE_bf16(v - v0), v2 = v - v0 - v1: 8+8+8 significant bits); a product x*w is accumulated as the 6 partial
// products x0w0 + x0w1 + x1w0 + x1w1 + x0w2 + x2w0 (smallest first) in the fp32 accumulator.  The 3 dropped terms are
// <= 2^-23 |x w|, i.e. the size of one fp32 rounding of the product: the result differs from the fp32-MFMA path like
// one summation order differs from another (tests/test_march_gpu.py holds both to the same oracle tolerance).
// Split-weight section of the packed buffer (after the fp32 section; host: dvgo.py::pack_mlp_mfma), units of 16 B:
//   W1S [NB][KB1][3][64]   lane l, 8 bf16: W1ext[j = mb*32+(l&31)][k = kb*16 + 8*(l>>5) + e]     KB1 = ceil(K1P/16)
//   W2S [NB][W/16][3][64]  lane l, 8 bf16: W2[j2 = mb2*32+(l&31)][n(kb,l>>5,e)],  n = (kb>>1)*32 + (e&3) + 8*(2*(kb&1)+(e>>2)) + 4*(l>>5)
//   B2S [NB][2][16] fp32   b2[mb2*32 + row(r,half)]    (accumulator initial value)
//   WOT [NB][16][2][4] fp32, BO [4] fp32 as in the fp32 section
// n(kb,h,e) is the neuron whose layer-1 accumulator lane-half h holds in register 8*(kb&1)+e of block kb>>1: the C layout of
// one layer is already the B-operand layout of the next, no data moves between lanes.
typedef __bf16 k4_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 k4_bf16x2 __attribute__((ext_vector_type(2)));
typedef float k4_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned k4_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned k4_u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// Corner voxel indices and trilinear weights of a shaded sample, The geometry kernel only records samples inside the bounding box (mask_outbbox,
// lib/dvgo.py:306-316), so the lower corner is a voxel and an upper corner leaves the grid only at the far faces: its axis
// factor is zeroed there (grid_sample's zero padding; the same weights as testing the 8 corners one by one: zl*yl*0 == 0)
// and its address offset dropped.  ~15 VALU instead of ~100 for the eight bounds tests.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k4_corner_setup(const MarchParams& P, float nx, float ny, float nz, unsigned (&cidx)[8], float (&cw)[8]) {
    const float ux = k4_unnorm(nx, P.X), uy = k4_unnorm(ny, P.Y), uz = k4_unnorm(nz, P.Z);
    const float fx = floorf(ux), fy = floorf(uy), fz = floorf(uz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const bool hx = x0 + 1 < P.X, hy = y0 + 1 < P.Y, hz = z0 + 1 < P.Z;
    const float xl = (fx + 1.f) - ux, xh = hx ? ux - fx : 0.f;
    const float yl = (fy + 1.f) - uy, yh = hy ? uy - fy : 0.f;
    const float zl = (fz + 1.f) - uz, zh = hz ? uz - fz : 0.f;
    cw[0] = zl * yl * xl; cw[1] = zh * yl * xl; cw[2] = zl * yh * xl; cw[3] = zh * yh * xl;
    cw[4] = zl * yl * xh; cw[5] = zh * yl * xh; cw[6] = zl * yh * xh; cw[7] = zh * yh * xh;
    const int xc = min(max(x0, 0), P.X - 1), yc = min(max(y0, 0), P.Y - 1), zc = min(max(z0, 0), P.Z - 1);   // never an out-of-range address
    const unsigned base = (unsigned)(xc * P.Y + yc) * (unsigned)P.Z + (unsigned)zc;
    const unsigned ox = hx ? (unsigned)(P.Y * P.Z) : 0u, oy = hy ? (unsigned)P.Z : 0u, oz = hz ? 1u : 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) cidx[c] = base + (K4_CX(c) ? ox : 0u) + (K4_CY(c) ? oy : 0u) + (K4_CZ(c) ? oz : 0u);
}
__device__ __forceinline__ unsigned k4_pk_bf16(float lo, float hi) {               // v_cvt_pk_bf16_f32 (RNE)
    const k4_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, k4_bf16x2));
}
// 8 floats -> three 8 x bf16 operands (exact 3-term split)
__device__ __forceinline__ void k4_split3(const float (&v)[8], uint4& t0, uint4& t1, uint4& t2) {
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // 2-wide fp32 vectors: the two subtractions of a pair are one v_pk_add_f32 (same IEEE results as the scalar form)
        const k4_f32x2 x = {v[2 * i], v[2 * i + 1]};
        p0[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, k4_bf16x2));
        const k4_f32x2 h0 = {__uint_as_float(p0[i] << 16), __uint_as_float(p0[i] & 0xffff0000u)};
        const k4_f32x2 r = x - h0;
        p1[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, k4_bf16x2));
        const k4_f32x2 h1 = {__uint_as_float(p1[i] << 16), __uint_as_float(p1[i] & 0xffff0000u)};
        const k4_f32x2 q = r - h1;
        p2[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, k4_bf16x2));
    }
    t0 = make_uint4(p0[0], p0[1], p0[2], p0[3]); t1 = make_uint4(p1[0], p1[1], p1[2], p1[3]); t2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
}
#define K4_MFMA_B3(ACC, A0, A1, A2, B0, B1, B2) do { \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A2), __builtin_bit_cast(k4_bf16x8, B0), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B2), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A1), __builtin_bit_cast(k4_bf16x8, B1), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A1), __builtin_bit_cast(k4_bf16x8, B0), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B1), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B0), ACC, 0, 0, 0); } while (0)

#ifndef K4_MLP_X2
#define K4_MLP_X2 1
#endif
// two accumulators side by side (same B operand): consecutive MFMAs never depend on each other; each accumulator still receives its
// six products in the order of K4_MFMA_B3 (bit-identical results)
#define K4_MFMA_B3_X2(ACCA, ACCB, A0, A1, A2, C0, C1, C2, B0, B1, B2) do { \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A2), __builtin_bit_cast(k4_bf16x8, B0), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C2), __builtin_bit_cast(k4_bf16x8, B0), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B2), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C0), __builtin_bit_cast(k4_bf16x8, B2), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A1), __builtin_bit_cast(k4_bf16x8, B1), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C1), __builtin_bit_cast(k4_bf16x8, B1), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A1), __builtin_bit_cast(k4_bf16x8, B0), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C1), __builtin_bit_cast(k4_bf16x8, B0), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B1), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C0), __builtin_bit_cast(k4_bf16x8, B1), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B0), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C0), __builtin_bit_cast(k4_bf16x8, B0), ACCB, 0, 0, 0); } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// Round 6, the DEFAULT arithmetic ("b2"): layer 1 keeps the exact 3-term form; the HIDDEN activations and the layer-2 weights are split
// into TWO bf16 terms (a ~ a0 + a1, 16 significant bits) and a product is a1 w0 + a0 w1 + a0 w0: half the matrix instructions of layer 2
// and 24 instead of 44 vector instructions per 8 activations.  Dropped: a1 w1, a0 w2, a2 w0 -- each <= 2^-16 |a w|; measured on the
// LLFF frame against the CPU oracle: tests/test_march_gpu.py (>= 100 dB), bench.py parity_vs_oracle.  K4_MLP_ARITH_B3 keeps rounds 2-5's
// exact form.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k4_split2(const float (&v)[8], uint4& t0, uint4& t1) {
    unsigned p0[4], p1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const k4_f32x2 x = {v[2 * i], v[2 * i + 1]};
        p0[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, k4_bf16x2));
        const k4_f32x2 h0 = {__uint_as_float(p0[i] << 16), __uint_as_float(p0[i] & 0xffff0000u)};
        const k4_f32x2 r = x - h0;
        p1[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, k4_bf16x2));
    }
    t0 = make_uint4(p0[0], p0[1], p0[2], p0[3]); t1 = make_uint4(p1[0], p1[1], p1[2], p1[3]);
}
#define K4_MFMA_B2(ACC, A0, A1, B0, B1) do { \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A1), __builtin_bit_cast(k4_bf16x8, B0), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B1), ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B0), ACC, 0, 0, 0); } while (0)
#define K4_MFMA_B2_X2(ACCA, ACCB, A0, A1, C0, C1, B0, B1) do { \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A1), __builtin_bit_cast(k4_bf16x8, B0), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C1), __builtin_bit_cast(k4_bf16x8, B0), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B1), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C0), __builtin_bit_cast(k4_bf16x8, B1), ACCB, 0, 0, 0); \
    ACCA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, A0), __builtin_bit_cast(k4_bf16x8, B0), ACCA, 0, 0, 0); \
    ACCB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k4_bf16x8, C0), __builtin_bit_cast(k4_bf16x8, B0), ACCB, 0, 0, 0); } while (0)

// NT1 / NT2 = bf16 terms of the layer-1 / layer-2 operands (weights AND activations of that layer): 3 (exact, "b3") or 2 ("b2")
#ifndef K4_B2_L1_TERMS
#define K4_B2_L1_TERMS 2                  // layer-1 terms of the default arithmetic (3: layer 1 exact, only layer 2 on 2-term splits -- A/B builds)
#endif
template <int W, int NHID, int NT1 = 3, int NT2 = 3>
struct MlpLayoutB3 {                      // offsets in floats from the start of the split section
    static constexpr int NB = W / 32;
    __host__ __device__ static int kb1(int k1p) { return (k1p + 15) >> 4; }
    __host__ __device__ static int w1s(int k1p) { (void)k1p; return 0; }
    __host__ __device__ static int w2s(int k1p) { return NB * kb1(k1p) * NT1 * 64 * 4; }
    __host__ __device__ static int b2s(int k1p) { return w2s(k1p) + (NHID ? NB * (W / 16) * NT2 * 64 * 4 : 0); }
    __host__ __device__ static int wot(int k1p) { return b2s(k1p) + (NHID ? NB * 2 * 16 : 0); }
    __host__ __device__ static int bo(int k1p) { return wot(k1p) + NB * 16 * 2 * 4; }
    __host__ __device__ static int total(int k1p) { return bo(k1p) + 4; }
};

// K4_SHADE_TIMING (profiling builds only, tools/r03_shade_timing.sh): s_memtime stamps at the phase boundaries of a shading batch,
// summed per wave and added to out_counters[8..15] -- where a batch's ~23k cycles go.  Not compiled into the product library.
#ifdef K4_SHADE_TIMING
#define K4_TSTAMP(SLOT) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                             tacc[SLOT] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define K4_TARGS , unsigned long long (&tacc)[8], unsigned long long& tlast
#define K4_TPASS , tacc, tlast
#else
#define K4_TSTAMP(SLOT) do { } while (0)
#define K4_TARGS
#define K4_TPASS
#endif
// max(x, 0) in ONE instruction, on the bit pattern: v_max_i32(bits, 0) -- a negative float is a negative integer, a non-negative one keeps
// its bits.  fmaxf(x, 0.f) compiles to a canonicalising v_max_f32 plus the max (and v_med3_f32 is folded back into that pair), and the
// kernel is bound by its vector work (a wave hides at most ~5 vector instructions per MFMA, profiles/r04_mfma_valu_overlap.md; this
// stream carries ~17): 128 of the ~1400 vector instructions of a 64-record batch were the second half of a ReLU.  Same values for every non-NaN input (-0 -> +0 either way); a NaN
// with a clear sign bit stays NaN as in torch.relu (fmaxf turned it into 0).  Not inline asm: the compiler must see the instruction
// to insert the wait states between an MFMA writing a register and a vector instruction reading it.
__device__ __forceinline__ float k4_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// A tile's layer-1 input: the shading kernel's feat[K1P][64] image in LDS.
struct LdsFeat {
    static constexpr bool kCompileTimeTile = false;
    const float* feat; int k1p, l31, half;
    __device__ __forceinline__ void load(int kb, int t, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kb * 16 + 8 * half + e;
            v[e] = k < k1p ? feat[k * 64 + t * 32 + l31] : 0.f;
        }
    }
};
// FAST shading path (round 6): the 16 layer-1 inputs of a record are built in ITS lane's registers and reach the B-operand layout
// (lane l: sample l & 31 of tile t, inputs 8 (l >> 5) .. + 7) by 8 v_permlane32_swap -- no LDS image, no ds_write / ds_read round trip.
struct RegFeat {
    static constexpr bool kCompileTimeTile = true;
    float x[8], y[8];                                     // tile 0's / tile 1's operand of this lane
    template <int T>
    __device__ __forceinline__ void load(int kb, std::integral_constant<int, T>, float (&v)[8]) {
        (void)kb;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = T ? y[e] : x[e];
    }
};
// The general form (any covered width / depth / input size), one 32-sample tile at a time.  NT2 = 3: the exact form of rounds 2-5 (every
// accumulator receives its products in the order of K4_MFMA_B3); NT2 = 2: the round-6 default (see k4_split2).  The 32-neuron output
// blocks of a layer are taken two at a time, side by side (independent accumulators: consecutive MFMAs never depend on each other);
// width 128 = two such passes per layer, so that at most 2 x 16 accumulators are live beside the split hidden activations.
template <int W, int NHID, int NT1, int NT2, class FS, bool TILE0_ONLY = false>      // TILE0_ONLY: the caller guarantees nproc <= 32
__device__ __forceinline__ void mlp_mfma_bx(const float* ws, FS& fs, int k1p, int lane, int half, int debug, int nproc,
                                            float& out0, float& out1, float& out2 K4_TARGS) {
    constexpr int NB = W / 32;
    constexpr int KB2 = W / 16;
    constexpr int NP = NB >= 2 ? NB / 2 : 1;              // passes of (up to) two output blocks
    constexpr int PB = NB >= 2 ? 2 : 1;                   // blocks per pass
    typedef MlpLayoutB3<W, NHID, NT1, NT2> ML;
    const int kb1n = ML::kb1(k1p);
    const int kb1 = (debug & 2) ? 0 : kb1n;
    const uint4* const w1s = reinterpret_cast<const uint4*>(ws + ML::w1s(k1p));
    const uint4* const w2s = reinterpret_cast<const uint4*>(ws + ML::w2s(k1p));
    const float* const b2s = ws + ML::b2s(k1p);
    const float* const wot = ws + ML::wot(k1p);
    const float* const bo = ws + ML::bo(k1p);
    out0 = out1 = out2 = 0.f;
    // one 32-sample tile at a time.  LdsFeat: NOT unrolled, the second tile reuses the code and the registers.  RegFeat: the tile index must be a
    // compile-time constant -- a run-time choice between the two register arrays made hipcc keep them in SCRATCH (64 bytes per lane written and read
    // back per batch: 350 MB of HBM writes per frame in the first FAST build, profiles/r06_marcher_pmc_raw.md) -- so the two tiles are two instances.
    auto tile_body = [&](auto tt) {
        const int t = tt;
        // ---------------- layer 1: H1^T[j][s] = sum_k W1ext[j][k] * X[k][s], exact 3-term products ----------------
        f32x16 h1[NB];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) h1[mb] = (f32x16)(0.f);
        for (int kb = 0; kb < kb1; ++kb) {
            float v[8];
            fs.load(kb, tt, v);
            uint4 x[NT1];
            if constexpr (NT1 == 3) k4_split3(v, x[0], x[1], x[2]);
            else k4_split2(v, x[0], x[1]);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const uint4* const wp = w1s + (((pp * PB) * kb1n + kb) * NT1) * 64 + lane;
                const uint4* const wq = w1s + (((pp * PB + PB - 1) * kb1n + kb) * NT1) * 64 + lane;
                if constexpr (NT1 == 3) {
                    const uint4 a0 = wp[0], a1 = wp[64], a2 = wp[128];
                    if constexpr (PB == 2) {
                        const uint4 c0 = wq[0], c1 = wq[64], c2 = wq[128];
                        K4_MFMA_B3_X2(h1[pp * 2], h1[pp * 2 + 1], a0, a1, a2, c0, c1, c2, x[0], x[1], x[NT1 - 1]);
                    } else {
                        K4_MFMA_B3(h1[0], a0, a1, a2, x[0], x[1], x[NT1 - 1]);
                    }
                } else {
                    const uint4 a0 = wp[0], a1 = wp[64];
                    if constexpr (PB == 2) {
                        const uint4 c0 = wq[0], c1 = wq[64];
                        K4_MFMA_B2_X2(h1[pp * 2], h1[pp * 2 + 1], a0, a1, c0, c1, x[0], x[1]);
                    } else {
                        K4_MFMA_B2(h1[0], a0, a1, x[0], x[1]);
                    }
                }
            }
        }
        K4_TSTAMP(3);                                    // layer 1
        k4_f32x2 pt01 = {0.f, 0.f};
        float pt2 = 0.f;
        auto out_block = [&](const f32x16& cc, int mb) {   // this lane's share of the output layer for one 32-neuron block
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 wo = *reinterpret_cast<const float4*>(wot + ((mb * 16 + r) * 2 + half) * 4);
                const float a0 = k4_relu(cc[r]);
                const k4_f32x2 w01 = {wo.x, wo.y}, aa = {a0, a0};
                pt01 = __builtin_elementwise_fma(w01, aa, pt01);                  // v_pk_fma_f32: channels 0 and 1 in one instruction
                pt2 = fmaf(wo.z, a0, pt2);
            }
        };
        if (NHID == 1 && !(debug & 2)) {
            // relu + split of this tile's hidden activations once
            uint4 hs[KB2][NT2];
#pragma unroll
            for (int kb = 0; kb < KB2; ++kb) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = k4_relu(h1[kb >> 1][8 * (kb & 1) + e]);
                if constexpr (NT2 == 3) k4_split3(v, hs[kb][0], hs[kb][1], hs[kb][NT2 - 1]);
                else k4_split2(v, hs[kb][0], hs[kb][1]);
            }
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                f32x16 c[PB];
#pragma unroll
                for (int q = 0; q < PB; ++q)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float4 bv = *reinterpret_cast<const float4*>(b2s + ((pp * PB + q) * 2 + half) * 16 + r4 * 4);
                        c[q][r4 * 4 + 0] = bv.x; c[q][r4 * 4 + 1] = bv.y; c[q][r4 * 4 + 2] = bv.z; c[q][r4 * 4 + 3] = bv.w;
                    }
#pragma unroll
                for (int kb = 0; kb < KB2; ++kb) {
                    const uint4* const wp = w2s + (((pp * PB) * KB2 + kb) * NT2) * 64 + lane;
                    if constexpr (PB == 2) {
                        const uint4* const wq = w2s + (((pp * PB + 1) * KB2 + kb) * NT2) * 64 + lane;
                        if constexpr (NT2 == 3) {
                            const uint4 a0 = wp[0], a1 = wp[64], a2 = wp[128], c0 = wq[0], c1 = wq[64], c2 = wq[128];
                            K4_MFMA_B3_X2(c[0], c[1], a0, a1, a2, c0, c1, c2, hs[kb][0], hs[kb][1], hs[kb][NT2 - 1]);
                        } else {
                            const uint4 a0 = wp[0], a1 = wp[64], c0 = wq[0], c1 = wq[64];
                            K4_MFMA_B2_X2(c[0], c[1], a0, a1, c0, c1, hs[kb][0], hs[kb][1]);
                        }
                    } else {
                        if constexpr (NT2 == 3) {
                            const uint4 a0 = wp[0], a1 = wp[64], a2 = wp[128];
                            K4_MFMA_B3(c[0], a0, a1, a2, hs[kb][0], hs[kb][1], hs[kb][NT2 - 1]);
                        } else {
                            const uint4 a0 = wp[0], a1 = wp[64];
                            K4_MFMA_B2(c[0], a0, a1, hs[kb][0], hs[kb][1]);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < PB; ++q) out_block(c[q], pp * PB + q);
            }
            K4_TSTAMP(4);                                // split of the hidden activations + layer 2 + output layer
        } else {
#pragma unroll
            for (int mb = 0; mb < NB; ++mb) out_block(h1[mb], mb);
        }
        // lanes l and l^32 hold the two halves of the neurons of sample (l&31) of this tile; sample 32*t + (l&31) belongs to
        // lane 32*t + (l&31)
        const float q0 = pt01.x + __shfl_xor(pt01.x, 32) + bo[0];
        const float q1 = pt01.y + __shfl_xor(pt01.y, 32) + bo[1];
        const float q2 = pt2 + __shfl_xor(pt2, 32) + bo[2];
        if (half == t) { out0 = q0; out1 = q1; out2 = q2; }
        K4_TSTAMP(5);                                    // output layer
    };
    if constexpr (FS::kCompileTimeTile) {
        tile_body(std::integral_constant<int, 0>{});
        if constexpr (!TILE0_ONLY) {
            if (32 < nproc) tile_body(std::integral_constant<int, 1>{});  // a bundle's last batch may hold no record in the second tile (wave-uniform)
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            if (t * 32 >= nproc) break;
            tile_body(t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: BOTH 32-sample tiles of a batch in one software pipeline (rgbnet 16 -> 64 -> 64 -> 3, the LLFF shape).
// mlp_mfma_b3 runs a tile as split -> layer-1 MFMAs -> ReLU + split -> layer-2 MFMAs -> output layer, every phase waiting for the one
// before it: inside ONE wave the matrix pipe and the vector pipe never work at the same time, and the measured per-SIMD cost of a batch is
// the SUM of the two (profiles/r05_marcher_split_path.md: ~1,240 vector instructions x 2.8 cycles + 120 MFMAs x 32 cycles).  Here the
// vector work of one stage runs under the matrix work of an INDEPENDENT stage: tile B's input split under tile A's layer 1, the ReLU +
// split of hidden block k+1 under the layer-2 MFMAs of block k, tile A's output layer under tile B's layer 2; sched_group_barrier
// prescribes the interleave (one MFMA, then a few vector instructions: a wave hides <= 5 per MFMA, profiles/r04_mfma_valu_overlap.md),
// sched_barrier keeps the stages apart.  Every accumulator and every output sum receives its terms in mlp_mfma_b3's order (layer 1 and
// layer 2 with the two 32-neuron blocks side by side, K4_MFMA_B3_X2): the same bits.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef K4_MLP_PAIR
#define K4_MLP_PAIR 1
#endif
#define K4_SGB_MFMA 0x008
#define K4_SGB_VALU 0x002
#define K4_SGB_DSRD 0x100
// one stage: NM MFMAs, each followed by NV vector instructions (the scheduler takes them from the stage's region in dependency order)
#define K4_STAGE_SCHED(NM, NV) do { _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) { \
        __builtin_amdgcn_sched_group_barrier(K4_SGB_MFMA, 1, 0); __builtin_amdgcn_sched_group_barrier(K4_SGB_VALU, (NV), 0); } } while (0)
template <int NT2>
__device__ __forceinline__ void k4_relu_split8(const f32x16& h, int hi, uint4 (&t)[NT2]) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = k4_relu(h[8 * hi + e]);
    if constexpr (NT2 == 3) k4_split3(v, t[0], t[1], t[2]);
    else k4_split2(v, t[0], t[1]);
}
// NT2 = 3: rounds 2-5's exact arithmetic (same bits as mlp_mfma_bx<64, 1, 3>); NT2 = 2: the round-6 default -- a layer-2 stage is 6
// MFMAs and ~32 vector instructions (8 ReLUs + a 2-term split) instead of 12 and ~52.
template <int NT1, int NT2, class FS>
__device__ __forceinline__ void mlp_pair64(const float* ws, FS& fs, int lane, int half,
                                           float& out0, float& out1, float& out2 K4_TARGS) {
    typedef MlpLayoutB3<64, 1, NT1, NT2> ML;
    constexpr int NM1 = 2 * (NT1 == 3 ? 6 : 3);          // MFMAs of a layer-1 stage
    constexpr int NM2 = 2 * (NT2 == 3 ? 6 : 3);          // ... of a layer-2 stage
    const uint4* const w1s = reinterpret_cast<const uint4*>(ws + ML::w1s(16));
    const uint4* const w2s = reinterpret_cast<const uint4*>(ws + ML::w2s(16));
    const float* const b2s = ws + ML::b2s(16);
    const float* const wot = ws + ML::wot(16);
    const float* const bo = ws + ML::bo(16);
    float vA[8], vB[8];
    if constexpr (FS::kCompileTimeTile) { fs.load(0, std::integral_constant<int, 0>{}, vA); fs.load(0, std::integral_constant<int, 1>{}, vB); }
    else { fs.load(0, 0, vA); fs.load(0, 1, vB); }
    // layer-1 weight fragments (both 32-neuron blocks), shared by the two tiles
    uint4 a[NT1], c[NT1], xA[NT1], xB[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) { a[t] = w1s[t * 64 + lane]; c[t] = w1s[(NT1 + t) * 64 + lane]; }
    if constexpr (NT1 == 3) k4_split3(vA, xA[0], xA[1], xA[NT1 - 1]); else k4_split2(vA, xA[0], xA[1]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S1: layer 1 of tile A  ||  input split of tile B ----
    f32x16 hA0 = (f32x16)(0.f), hA1 = (f32x16)(0.f), hB0 = (f32x16)(0.f), hB1 = (f32x16)(0.f);
    if constexpr (NT1 == 3) { K4_MFMA_B3_X2(hA0, hA1, a[0], a[1], a[NT1 - 1], c[0], c[1], c[NT1 - 1], xA[0], xA[1], xA[NT1 - 1]); }
    else { K4_MFMA_B2_X2(hA0, hA1, a[0], a[1], c[0], c[1], xA[0], xA[1]); }
    if constexpr (NT1 == 3) k4_split3(vB, xB[0], xB[1], xB[NT1 - 1]); else k4_split2(vB, xB[0], xB[1]);
    K4_STAGE_SCHED(NM1, NT1 == 3 ? 3 : 4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- S2: layer 1 of tile B  ||  ReLU + split of tile A's hidden block 0 ----
    uint4 hsA[4][NT2], hsB[4][NT2];
    if constexpr (NT1 == 3) { K4_MFMA_B3_X2(hB0, hB1, a[0], a[1], a[NT1 - 1], c[0], c[1], c[NT1 - 1], xB[0], xB[1], xB[NT1 - 1]); }
    else { K4_MFMA_B2_X2(hB0, hB1, a[0], a[1], c[0], c[1], xB[0], xB[1]); }
    k4_relu_split8<NT2>(hA0, 0, hsA[0]);
    K4_STAGE_SCHED(NM1, (NT2 == 3 ? 52 : 32) / NM1 + 1);
    __builtin_amdgcn_sched_barrier(0);
    // layer-2 accumulators start from the bias
    f32x16 cA0, cA1, cB0, cB1;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const float4 b0v = *reinterpret_cast<const float4*>(b2s + (0 * 2 + half) * 16 + r4 * 4);
        const float4 b1v = *reinterpret_cast<const float4*>(b2s + (1 * 2 + half) * 16 + r4 * 4);
        cA0[r4 * 4 + 0] = b0v.x; cA0[r4 * 4 + 1] = b0v.y; cA0[r4 * 4 + 2] = b0v.z; cA0[r4 * 4 + 3] = b0v.w;
        cA1[r4 * 4 + 0] = b1v.x; cA1[r4 * 4 + 1] = b1v.y; cA1[r4 * 4 + 2] = b1v.z; cA1[r4 * 4 + 3] = b1v.w;
    }
    cB0 = cA0; cB1 = cA1;
    // ---- S3..S6: layer 2 of tile A, hidden block kb  ||  ReLU + split of the next hidden block (of A, then B's first) ----
#define K4_L2_STAGE(C0, C1, HS, KB, NEXT_STMT, NV) do { \
        const uint4* const wp_ = w2s + ((KB) * NT2) * 64 + lane; \
        const uint4* const wq_ = w2s + ((4 + (KB)) * NT2) * 64 + lane; \
        if constexpr (NT2 == 3) { \
            const uint4 p0_ = wp_[0], p1_ = wp_[64], p2_ = wp_[128], q0_ = wq_[0], q1_ = wq_[64], q2_ = wq_[128]; \
            K4_MFMA_B3_X2(C0, C1, p0_, p1_, p2_, q0_, q1_, q2_, HS[KB][0], HS[KB][1], HS[KB][NT2 - 1]); \
        } else { \
            const uint4 p0_ = wp_[0], p1_ = wp_[64], q0_ = wq_[0], q1_ = wq_[64]; \
            K4_MFMA_B2_X2(C0, C1, p0_, p1_, q0_, q1_, HS[KB][0], HS[KB][1]); \
        } \
        NEXT_STMT; \
        K4_STAGE_SCHED(NM2, NV); \
        __builtin_amdgcn_sched_barrier(0); } while (0)
    K4_TSTAMP(3);                                        // (timing builds) input splits + layer 1 of both tiles + first hidden block's split
#ifndef K4_PAIR_NVS
#define K4_PAIR_NVS 5
#endif
#ifndef K4_PAIR_NVO
#define K4_PAIR_NVO 8
#endif
#ifndef K4_PAIR_NVL
#define K4_PAIR_NVL 4
#endif
    constexpr int NVS = NT2 == 3 ? 4 : K4_PAIR_NVS;                // vector instructions per MFMA of a ReLU + split stage (~52 / 12, ~32 / 6)
    K4_L2_STAGE(cA0, cA1, hsA, 0, k4_relu_split8<NT2>(hA0, 1, hsA[1]), NVS);
    K4_L2_STAGE(cA0, cA1, hsA, 1, k4_relu_split8<NT2>(hA1, 0, hsA[2]), NVS);
    K4_L2_STAGE(cA0, cA1, hsA, 2, k4_relu_split8<NT2>(hA1, 1, hsA[3]), NVS);
    K4_L2_STAGE(cA0, cA1, hsA, 3, k4_relu_split8<NT2>(hB0, 0, hsB[0]), NVS);
    // ---- S7..S10: layer 2 of tile B  ||  the next hidden block of B + tile A's output layer (its accumulators are complete) ----
    k4_f32x2 ptA01 = {0.f, 0.f}, ptB01 = {0.f, 0.f};
    float ptA2 = 0.f, ptB2 = 0.f;
#ifndef K4_OUT_SCALAR_FMA
#define K4_OUT_SCALAR_FMA 1      // the output layer's channel pair as two v_fma_f32 instead of one v_pk_fma_f32: packed fp32 beside MFMAs is an anti-lever (K2 377 -> 372 us, same bits; 0 = the packed form)
#endif
#define K4_OUT_HALF(C, MB2, R0, PT01, PT2) do { \
        _Pragma("unroll") for (int r_ = (R0); r_ < (R0) + 8; ++r_) { \
            const float4 wo_ = *reinterpret_cast<const float4*>(wot + (((MB2) * 16 + r_) * 2 + half) * 4); \
            const float a_ = k4_relu(C[r_]); \
            if (K4_OUT_SCALAR_FMA) { PT01.x = fmaf(wo_.x, a_, PT01.x); PT01.y = fmaf(wo_.y, a_, PT01.y); } \
            else { const k4_f32x2 w01_ = {wo_.x, wo_.y}, aa_ = {a_, a_}; PT01 = __builtin_elementwise_fma(w01_, aa_, PT01); } \
            PT2 = fmaf(wo_.z, a_, PT2); } } while (0)
    K4_TSTAMP(4);                                        // layer 2 of tile A (+ splits)
    constexpr int NVO = NT2 == 3 ? 5 : K4_PAIR_NVO;                // ... of a ReLU + split + half-block output stage (~76 / 12, ~56 / 6: what does not fit runs behind the stage)
    K4_L2_STAGE(cB0, cB1, hsB, 0, k4_relu_split8<NT2>(hB0, 1, hsB[1]); K4_OUT_HALF(cA0, 0, 0, ptA01, ptA2), NVO);
    K4_L2_STAGE(cB0, cB1, hsB, 1, k4_relu_split8<NT2>(hB1, 0, hsB[2]); K4_OUT_HALF(cA0, 0, 8, ptA01, ptA2), NVO);
    K4_L2_STAGE(cB0, cB1, hsB, 2, k4_relu_split8<NT2>(hB1, 1, hsB[3]); K4_OUT_HALF(cA1, 1, 0, ptA01, ptA2), NVO);
    K4_L2_STAGE(cB0, cB1, hsB, 3, K4_OUT_HALF(cA1, 1, 8, ptA01, ptA2), NT2 == 3 ? 2 : K4_PAIR_NVL);
#undef K4_L2_STAGE
    K4_TSTAMP(5);                                        // layer 2 of tile B (+ splits, tile A's output layer)
    // ---- S11: tile B's output layer ----
    K4_OUT_HALF(cB0, 0, 0, ptB01, ptB2); K4_OUT_HALF(cB0, 0, 8, ptB01, ptB2);
    K4_OUT_HALF(cB1, 1, 0, ptB01, ptB2); K4_OUT_HALF(cB1, 1, 8, ptB01, ptB2);
#undef K4_OUT_HALF
    // lanes l and l^32 hold the two halves of the neurons of sample (l&31) of a tile; sample 32 t + (l&31) belongs to lane 32 t + (l&31)
    const float qA0 = ptA01.x + __shfl_xor(ptA01.x, 32) + bo[0], qA1 = ptA01.y + __shfl_xor(ptA01.y, 32) + bo[1], qA2 = ptA2 + __shfl_xor(ptA2, 32) + bo[2];
    const float qB0 = ptB01.x + __shfl_xor(ptB01.x, 32) + bo[0], qB1 = ptB01.y + __shfl_xor(ptB01.y, 32) + bo[1], qB2 = ptB2 + __shfl_xor(ptB2, 32) + bo[2];
    out0 = half ? qB0 : qA0; out1 = half ? qB1 : qA1; out2 = half ? qB2 : qA2;
    K4_TSTAMP(2);                                        // tile B's output layer + shuffles
}

