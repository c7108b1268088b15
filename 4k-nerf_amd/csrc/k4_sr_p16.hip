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
    // SFT epilogue (k4_conv3x3_p16_sft_multi): the layer's result v goes through the NEXT SFTLayer before it is stored pre-split into y2
    const float* cond[K4_MAX_JOBS]; void* y2[K4_MAX_JOBS];
    const void* w_sfe; int cond_stride, y2_stride; float cond_scale, sft_slope;
    int debug;               // K4_SR_DEBUG ablation bits (profiling only, WRONG results; 0 in production): 64 = activation DMA for chunk 0 only, 128 = weight DMA for
                             // chunk 0 only, 256 = no MFMAs, 512 = every pixel reads pixel 0 (no HBM traffic)
};

// UP (K4_PRE_UPSAMPLE2X): the layer reads its input through a nearest x2 upsampling (lib/sr_esrnet.py:461-463).  Output pixel (2Y + py, 2X + px)
// then sees only a 2 x 2 neighbourhood of LR pixels -- the 3 x 3 taps that land on the same LR pixel add up: per PHASE (py, px) the layer is a
// 2 x 2 convolution with the weights W'[a][b] = sum of the taps (dy, dx) with (py + dy - 1) >> 1 == a + py - 1 (rows; same for columns), 16
// tap matrices instead of 36 per 2 x 2 output pixels: 2.25x fewer matrix instructions, exact algebra (the tap sums are formed in fp32 by the
// packer: the products differ from the 9-tap form by one rounding of a weight sum).  A workgroup = 8 x 32 LR positions of the two column phases
// (py, 0), (py, 1) of one row phase: four accumulator blocks per wave, the haloed tile fetched once for both (P16_CHUNK_UP); the two row-phase
// workgroups of a tile are adjacent in launch order.
//
// SFT (k4_conv3x3_p16_sft_multi): the SFTLayer that consumes this layer's result (lib/sr_esrnet.py:112-123,149-156: sft1 after conv4, the next
// dense block's sft0 after conv5) runs in the epilogue instead of as a launch of its own (36 of a frame's 111 launches, HBM-bound at 75-125 us
// each).  scale / shift depend on the condition pixel only: two 32 -> 32 -> 32 1x1 stacks per 32-channel output block, on the matrix pipe in
// this kernel's arithmetic (three fp16 products, operands split in two fp16 terms): 24 MFMAs per 32 pixels against the layer's 270-324.  The
// SFT operand of the (layer, output block) -- 17 KB, P16_SFE_* below -- arrives by DMA in the weight buffer the LAST chunk does not use, issued
// where a next chunk's DMA would be; the lane's condition rows are requested before the chunk loop.  Hidden activations never leave registers:
// GEMM 1's C layout is GEMM 2's B layout once the host walks GEMM 2's k in accumulator-register order.  Every scale rides in the operands
// (condition 2^Ec, per-neuron 2^Eh[j] from the bound of |hidden j| over the calibrated condition range): no exponent arithmetic here.
#define P16_SFE_A2 8192               /* A1 [path][kb][term][64 lanes] x 16 B | A2 [path][kb][term][64] x 16 B | tables [us1|b1|us2|b2][path][half][16] fp32 */
#define P16_SFE_TAB 16384
#define P16_SFE_BYTES 17408
// K4_P16_TIMING (profiling builds only, tools/p16_phase_timing.py): s_memtime stamps at the phase boundaries, summed over all waves into
// k4_p16_timing[] (read and reset through k4_debug_p16_timing).  Not compiled into the product library.
#ifdef K4_P16_TIMING
#define P16_TIMING_WAVES 65536
__device__ unsigned long long k4_p16_timing[P16_TIMING_WAVES][8];      // per wave (index 4 blockIdx + wave): no atomics -- 17680 waves adding to nine words took longer than the layer
#define P16_TSTAMP(SLOT) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                              tacc[SLOT] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define P16_TSTAMP(SLOT) do { } while (0)
#endif
template <bool OUT16, bool UP, bool SFT = false>
__global__ __launch_bounds__(256, 2) void k4_conv_p16_kernel(const P16Conv M) {
#ifdef K4_P16_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long mt0_ = tlast, rt0_ = __builtin_amdgcn_s_memrealtime();     // (non-SFT: slots 5 / 6 = the wave's life in s_memtime / s_memrealtime (100 MHz) ticks)
#define P16_TFLUSH() do { const unsigned wi_ = blockIdx.x * 4 + (threadIdx.x >> 6); \
                          if ((threadIdx.x & 63) == 0 && wi_ < P16_TIMING_WAVES) { tacc[7] = 1; _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) k4_p16_timing[wi_][i_] += tacc[i_]; } } while (0)
#else
#define P16_TFLUSH() do { } while (0)
#endif
    constexpr int W_CH = UP ? 2 * P16_W_BYTES_UP : P16_W_BYTES;       // weight bytes of one (chunk, output block[, phase PAIR (py, 0), (py, 1)])
    constexpr int W_NI = W_CH / 1024;                                  // DMA instructions
    constexpr int NSUB = 18;                                           // sub-stages (tap x row) per chunk of the 3 x 3 form
    constexpr int NTAP = 9;
    __shared__ __attribute__((aligned(16))) unsigned char wbuf0[P16_W_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char wbuf1[P16_W_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char abuf0[P16_ACT_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char abuf1[P16_ACT_BYTES];
    // LDS-typed views (the kernel selects between the buffers at run time; generic pointers that are selected and then cast back to LDS trip
    // hipcc's backend: "Illegal instruction detected: V_CMP_NE_U32 0, src_shared_base")
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef const __attribute__((address_space(3))) p16_u32x4 lds_u32x4;
    typedef const __attribute__((address_space(3))) p16_f32x4 lds_f32x4;
    lds_u8* const W0 = (lds_u8*)wbuf0; lds_u8* const W1 = (lds_u8*)wbuf1; lds_u8* const A0 = (lds_u8*)abuf0; lds_u8* const A1 = (lds_u8*)abuf1;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform for the compiler too: DMA instruction indices, M0, scalar offsets
    const int b = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (b >= M.total) return;
    int g = 0;
    while (g + 1 < M.n && b >= M.blk_end[g]) ++g;
    const int local = b - (g ? M.blk_end[g - 1] : 0);
    const int nb_count = M.cout >> 5;
    const int py = UP ? (local & 1) : 0;                               // UP: a workgroup = both column phases (py, 0), (py, 1) of its tile
    const int lt = UP ? (local >> 1) : local;
    const int tile = lt / nb_count, nb = lt - tile * nb_count;
    const int H = M.H[g], W = M.W[g];                                  // OUTPUT size; the tile grid and the DMA plan live on the input (LR) image
    const int srcH = UP ? H >> 1 : H, srcW = UP ? W >> 1 : W;
    const int tiles_x = M.tiles_x[g];
    const int x0 = (tile % tiles_x) * 32, y0 = (tile / tiles_x) * 8;   // input-image coordinates of the tile
    const int nchunks = M.cin >> 4;

    // ---- DMA plan of this thread (chunk independent): activation instructions k = wv + 4 i, weight instructions k = wv + 4 i ----
    const __amdgpu_buffer_rsrc_t xrs = p16_rsrc(M.x[g], (unsigned)(((long long)(srcH * srcW - 1) * M.cin_stride + M.cin) * 4));
    const int w_tail = nchunks * nb_count * (UP ? 2 : 1) * W_CH;      // byte offset of the [cout] floats 2^-a[co] behind the weights
    const __amdgpu_buffer_rsrc_t wrs = p16_rsrc(M.w, (unsigned)(w_tail + M.cout * 4));
    // DMA plan of this wave: instruction k of a plan belongs to wave k % 4.  (A fifth / sixth wave that only issues the DMA -- the matrix waves
    // then never sit on the vector-memory queue, ~20 % of their life by s_memtime stamps -- was built and measured: bit-identical, the chunk
    // loop then waits on the producers' issue + flight time instead, no layer faster: profiles/r04_p16_producer_waves_not_faster.md)
    constexpr int NISS = 4;
    constexpr int AW = (P16_ACT_INSTR + NISS - 1) / NISS, WW = (W_NI + NISS - 1) / NISS, SW = (P16_SFE_BYTES / 1024 + NISS - 1) / NISS;
    const int wi = wv;
    unsigned aoff[AW];
    {
#pragma unroll
        for (int i = 0; i < AW; ++i) {
            const int item = (wi + NISS * i) * 64 + lane;
            const int p = item >> 2, j = item & 3;
            const int row = p / P16_COLS, col = p - row * P16_COLS;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            const bool inside = item < P16_ACT_ITEMS && gy >= 0 && gy < srcH && gx >= 0 && gx < srcW;
            aoff[i] = inside ? (unsigned)(((M.debug & 512) ? 0 : (gy * srcW + gx) * M.cin_stride * 4) + ((j ^ ((col >> 2) & 3)) << 4)) : P16_OOB;
        }
    }
    const unsigned woff = (unsigned)(lane * 16);
#define P16_ISSUE(CH, WB, AB) do { { \
        const int wso_ = (((CH) * nb_count + nb) * (UP ? 2 : 1) + py) * W_CH; \
        const int aso_ = (CH) * 64; \
        _Pragma("unroll") for (int i_ = 0; i_ < WW; ++i_) { \
            const int k_ = wi + NISS * i_; \
            if (k_ < W_NI && (!(M.debug & 128) || (CH) == 0)) \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)((WB) + k_ * 1024), 16, (int)woff, wso_ + k_ * 1024, 0, 0); \
        } \
        _Pragma("unroll") for (int i_ = 0; i_ < AW; ++i_) { \
            const int k_ = wi + NISS * i_; \
            if ((k_ < P16_ACT_INSTR - 1 || (k_ == P16_ACT_INSTR - 1 && lane < P16_ACT_ITEMS - (P16_ACT_INSTR - 1) * 64)) && (!(M.debug & 64) || (CH) == 0)) \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)((AB) + k_ * 1024), 16, (int)aoff[i_], aso_, 0, 0); \
        } } } while (0)

    const __amdgpu_buffer_rsrc_t srs = p16_rsrc(SFT ? M.w_sfe : M.w, SFT ? (unsigned)(nb_count * P16_SFE_BYTES) : 0u);
    const __amdgpu_buffer_rsrc_t brs = p16_rsrc(M.bias, (unsigned)(M.cout * 4));
    // (+ this block's 32 + 32 epilogue constants 2^-a[co], bias[co] behind the operand: the SFT kernel keeps them out of its registers)
#define P16_ISSUE_SFE(WB) do { { \
        _Pragma("unroll") for (int i_ = 0; i_ < SW; ++i_) { \
            const int k_ = wi + NISS * i_; \
            if (k_ < P16_SFE_BYTES / 1024) \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (__attribute__((address_space(3))) void*)((WB) + k_ * 1024), 16, (int)woff, nb * P16_SFE_BYTES + k_ * 1024, 0, 0); \
        } \
        if (wi == 0 && lane < 8) { \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)((WB) + P16_SFE_BYTES), 16, (int)woff, w_tail + nb * 128, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)((WB) + P16_SFE_BYTES + 128), 16, (int)woff, nb * 128, 0, 0); \
        } } } while (0)
    // every DMA of a chunk must have LANDED when its barrier releases the readers: an explicit wait -- hipcc's own count before s_barrier is
    // not to be relied on (with the eight condition loads in flight it emitted vmcnt(8) at the loop's first barrier, which on the back edge
    // let eight DMA instructions of the next chunk stay in flight: run-to-run different bits once the issue moved closer to the barrier)
#define P16_BARRIER() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
    P16_ISSUE(0, W0, A0);

    // ---- fragment addresses: weights (A operand) lane = (channel group, co); activations (B operand) lane = (channel group, column) ----
    const unsigned wrd = (unsigned)(lane * 16);
    unsigned ard[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = l31 + dx;                                     // UP: column tap b of phase px reads haloed column l31 + b + px = ard[b + px]
            ard[dx][t] = (unsigned)(((wv * 2 + (UP ? py : 0)) * P16_COLS + col) * 64 + (((2 * t + half) ^ ((col >> 2) & 3)) << 4));
        }

    // per-lane epilogue tables: output channels co(q, e) = nb*32 + 8q + 4 half + e
    const int cob = nb * 32 + 4 * half;
    p16_f32x4 us[4], bs[4];
    if constexpr (!SFT) {
        const float* const wtail = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(M.w) + (size_t)w_tail);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            us[q] = *reinterpret_cast<const p16_f32x4*>(wtail + cob + 8 * q);
            bs[q] = *reinterpret_cast<const p16_f32x4*>(M.bias + cob + 8 * q);
        }
    }

    p16_f32x16 acc[UP ? 4 : 2];                                        // [row r] (UP: [column phase px][row r])
#pragma unroll
    for (int i = 0; i < (UP ? 4 : 2); ++i) acc[i] = (p16_f32x16)(0.f);

    // SFT: the lane's condition channels kb*16 + 8*half + 0..7 (kb = 0, 1) of its two pixels, requested now; the operand's DMA plan
    p16_u32x4 cq[SFT ? 2 : 1][4];
    if constexpr (SFT) {
        const __amdgpu_buffer_rsrc_t crs = p16_rsrc(M.cond[g], (unsigned)(((long long)(H * W - 1) * M.cond_stride + 32) * 4));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int gy = y0 + wv * 2 + r, gxc = x0 + l31;
            const unsigned coff = (gy < H && gxc < W) ? (unsigned)((gy * W + gxc) * M.cond_stride * 4 + half * 32) : P16_OOB;
#pragma unroll
            for (int i = 0; i < 4; ++i) cq[r][i] = __builtin_amdgcn_raw_buffer_load_b128(crs, (int)(coff + (unsigned)((i >> 1) * 64 + (i & 1) * 16)), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // tap t -> (row, column) offset in the haloed tile: 3 x 3 taps (dy, dx) = (t / 3, t % 3)
#define P16_TROW(T) ((T) / 3)
#define P16_TCOL(T) ((T) % 3)
#define P16_RDW(DST, WB, T) do { \
        DST[0] = *reinterpret_cast<lds_u32x4*>(WB + wrd + (0 * NTAP + (T)) * 1024); \
        DST[1] = *reinterpret_cast<lds_u32x4*>(WB + wrd + (1 * NTAP + (T)) * 1024); } while (0)
#define P16_RDX(DST, AB, U) do { \
        const int t_ = (U) >> 1, r_ = (U) & 1; \
        DST[0] = *reinterpret_cast<lds_u32x4*>(AB + ard[P16_TCOL(t_)][0] + (r_ + P16_TROW(t_)) * (P16_COLS * 64)); \
        DST[1] = *reinterpret_cast<lds_u32x4*>(AB + ard[P16_TCOL(t_)][1] + (r_ + P16_TROW(t_)) * (P16_COLS * 64)); } while (0)
    // one chunk: its DMA has been issued an iteration ago; wait, barrier, issue the next chunk's DMA into the other buffers, 18 sub-stages
    // (tap x row) of 3 MFMAs with the fragments of sub-stage u + 2 / tap t + 1 read under the MFMAs of u
#define P16_CHUNK(CH, WB, AB, WBN, ABN) do { \
        P16_TSTAMP(CH == 0 ? 0 : 3); \
        P16_BARRIER(); \
        P16_TSTAMP(1); \
        if ((CH) + 1 < nchunks) P16_ISSUE((CH) + 1, WBN, ABN); \
        else if (SFT) P16_ISSUE_SFE(WBN); \
        P16_TSTAMP(2); \
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

    // UP: one chunk of the phase pair = 4 stages (row tap a, row r); a stage reads the THREE haloed columns l31 + 0 / 1 / 2 of input row r + a + py
    // once -- column tap b of phase px multiplies column b + px, so the middle column serves both phases -- and issues 4 (px, b) x 3 MFMAs.
    // LDS weight image of the pair: [px][hi | lo][tap a*2+b][64 lanes] x 16 B.  Fragments of stage s + 1 (and of row tap 1 during stage 1) are read
    // under the MFMAs of stage s.  (Four single-phase workgroups per tile fetched the same activations four times for 24 MFMAs per chunk and
    // wave: the layer was bound by the DMA issue.)
#define P16_RDW_UP(DST, WB, A) do { \
        _Pragma("unroll") for (int px_ = 0; px_ < 2; ++px_) \
        _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) \
        _Pragma("unroll") for (int tm_ = 0; tm_ < 2; ++tm_) \
            DST[px_][b_][tm_] = *reinterpret_cast<lds_u32x4*>(WB + wrd + ((px_ * 2 + tm_) * 4 + (A) * 2 + b_) * 1024); } while (0)
#define P16_RDX_UP(DST, AB, S) do { \
        _Pragma("unroll") for (int c_ = 0; c_ < 3; ++c_) \
        _Pragma("unroll") for (int tm_ = 0; tm_ < 2; ++tm_) \
            DST[c_][tm_] = *reinterpret_cast<lds_u32x4*>(AB + ard[c_][tm_] + (((S) & 1) + ((S) >> 1)) * (P16_COLS * 64)); } while (0)
#define P16_CHUNK_UP(CH, WB, AB, WBN, ABN) do { \
        P16_TSTAMP(CH == 0 ? 0 : 3); \
        P16_BARRIER(); \
        P16_TSTAMP(1); \
        if ((CH) + 1 < nchunks) P16_ISSUE((CH) + 1, WBN, ABN); \
        P16_TSTAMP(2); \
        p16_u32x4 ww[2][2][2][2], xx[2][3][2];                   /* [buffer][px][b][term], [buffer][column][term] */ \
        P16_RDW_UP(ww[0], WB, 0); \
        P16_RDX_UP(xx[0], AB, 0); \
        if (!(M.debug & 256)) \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {       /* stage s = (a = s >> 1, r = s & 1) */ \
            const int a_ = s_ >> 1, r_ = s_ & 1; \
            if (s_ == 1) P16_RDW_UP(ww[1], WB, 1); \
            if (s_ + 1 < 4) P16_RDX_UP(xx[(s_ + 1) & 1], AB, s_ + 1); \
            __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_) \
            _Pragma("unroll") for (int px_ = 0; px_ < 2; ++px_) { \
                const p16_f16x8 wh = __builtin_bit_cast(p16_f16x8, ww[a_][px_][b_][0]), wl = __builtin_bit_cast(p16_f16x8, ww[a_][px_][b_][1]); \
                const p16_f16x8 xh = __builtin_bit_cast(p16_f16x8, xx[s_ & 1][b_ + px_][0]), xl = __builtin_bit_cast(p16_f16x8, xx[s_ & 1][b_ + px_][1]); \
                acc[px_ * 2 + r_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[px_ * 2 + r_], 0, 0, 0); \
                acc[px_ * 2 + r_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[px_ * 2 + r_], 0, 0, 0); \
                acc[px_ * 2 + r_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[px_ * 2 + r_], 0, 0, 0); \
            } \
            __builtin_amdgcn_sched_barrier(0); \
        } } while (0)

    for (int ch = 0; ch < nchunks; ch += 2) {
        if constexpr (UP) {
            P16_CHUNK_UP(ch, W0, A0, W1, A1);
            if (ch + 1 < nchunks) P16_CHUNK_UP(ch + 1, W1, A1, W0, A0);
        } else {
            P16_CHUNK(ch, W0, A0, W1, A1);
            if (ch + 1 < nchunks) P16_CHUNK(ch + 1, W1, A1, W0, A0);
        }
    }
#undef P16_CHUNK
#undef P16_CHUNK_UP
#undef P16_RDW_UP
#undef P16_RDX_UP
#undef P16_RDX
#undef P16_RDW
#undef P16_TROW
#undef P16_TCOL
#undef P16_ISSUE
#undef P16_ISSUE_SFE

    P16_TSTAMP(3);
    // ---- epilogue: lane = pixel (x0 + l31, row y0 + 2 wv + r), registers 4q .. 4q+3 = channels cob + 8q + 0..3 ----
    const int gx = x0 + l31;                                           // output column of this lane (UP: 2 gx + px, below)
    const float sl = (M.flags & K4_EPI_LRELU) ? M.slope : 1.f;
    const bool has_res = (M.flags & K4_EPI_RES) != 0;
    const __amdgpu_buffer_rsrc_t yrs = p16_rsrc(M.y[g], (unsigned)(((long long)(H * W - 1) * M.cout_stride + M.cout) * 4));
    const __amdgpu_buffer_rsrc_t rrs = p16_rsrc(has_res ? (const void*)M.res[g] : M.y[g], has_res ? (unsigned)(((long long)(H * W - 1) * M.res_stride + M.cout) * 4) : 0u);
    if constexpr (SFT) {
        P16_BARRIER();                                                       // the SFT operand has landed; every wave is done with the chunk buffers
        P16_TSTAMP(6);
        const lds_u8* const SB = (nchunks & 1) ? W1 : W0;                    // the weight buffer the last chunk did not use
        const lds_u8* const tab = SB + P16_SFE_TAB;                          // tables: floats
        const __amdgpu_buffer_rsrc_t y2rs = p16_rsrc(M.y2[g], (unsigned)(((long long)(H * W - 1) * M.y2_stride + M.cout) * 4));
        const bool dual = M.y[g] != nullptr;                                 // the layer's own result is kept too (fp32: the next block's residual)
        const float one = 1.f;
        float amax = 0.f, cmax = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int gy = y0 + wv * 2 + r;
            if (gy >= H) continue;                                           // wave-uniform
            const bool ok = gx < W;
            const unsigned pix = (unsigned)(gy * W + gx);
            const unsigned roff = ok ? (pix * (unsigned)M.res_stride + (unsigned)cob) * 4u : P16_OOB;
            p16_u32x4 rq[4];
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rq[q] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(roff + (unsigned)(q * 32)), 0, 0);
            }
            // ---- the layer's own result v (constants from the LDS image), kept fp32 when the caller wants it ----
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const p16_f32x4 usq = *reinterpret_cast<lds_f32x4*>(SB + P16_SFE_BYTES + (4 * half + 8 * q) * 4);
                const p16_f32x4 bsq = *reinterpret_cast<lds_f32x4*>(SB + P16_SFE_BYTES + 128 + (4 * half + 8 * q) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaf(acc[r][4 * q + e], usq[e], bsq[e]);
                    t = fmaxf(t, t * sl);
                    if (has_res) t = p16_mul_add(t, M.res_scale, __uint_as_float(rq[q][e]));
                    v[4 * q + e] = t;
                }
                if (dual) {
                    const unsigned yoff = ok ? (pix * (unsigned)M.cout_stride + (unsigned)cob) * 4u + (unsigned)(q * 32) : P16_OOB;
                    const p16_u32x4 o4 = {__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(o4, yrs, (int)yoff, 0, 0);
                }
            }
            // ---- B operand of GEMM 1: the lane's condition channels, split here ----
            p16_u32x4 bh[2], bl[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const p16_u32x4 c0 = cq[r][2 * kb], c1 = cq[r][2 * kb + 1];
                const float va[4] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w)};
                const float vb[4] = {__uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w)};
                cmax = fmaxf(fmaxf(cmax, fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(va[2]), fabsf(va[3])))),
                             fmaxf(fmaxf(fabsf(vb[0]), fabsf(vb[1])), fmaxf(fabsf(vb[2]), fabsf(vb[3]))));
                unsigned Ha[2], La[2], Hb[2], Lb[2];
                p16_split4(va, M.cond_scale, Ha, La);
                p16_split4(vb, M.cond_scale, Hb, Lb);
                bh[kb] = p16_u32x4{Ha[0], Ha[1], Hb[0], Hb[1]};
                bl[kb] = p16_u32x4{La[0], La[1], Lb[0], Lb[1]};
            }
            // ---- one path at a time (scale, then shift): GEMM 1 -> hidden (registers = GEMM 2's B operand) -> GEMM 2 -> fold into v.
            //      x (scale + 1) + shift with the reference's two roundings: v <- v * (scale + 1), then v <- v + shift ----
#pragma unroll
            for (int pth = 0; pth < 2; ++pth) {
                p16_f32x16 h = (p16_f32x16)(0.f);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const p16_f16x8 ah = __builtin_bit_cast(p16_f16x8, *reinterpret_cast<lds_u32x4*>(SB + ((pth * 2 + kb) * 2 + 0) * 1024 + lane * 16));
                    const p16_f16x8 al = __builtin_bit_cast(p16_f16x8, *reinterpret_cast<lds_u32x4*>(SB + ((pth * 2 + kb) * 2 + 1) * 1024 + lane * 16));
                    const p16_f16x8 xh = __builtin_bit_cast(p16_f16x8, bh[kb]), xl = __builtin_bit_cast(p16_f16x8, bl[kb]);
                    h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, h, 0, 0, 0);
                    h = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, h, 0, 0, 0);
                    h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, h, 0, 0, 0);
                }
                // hidden j 2^Eh[j] = lrelu(acc us1 + b1) (tables carry 2^Eh: lrelu is positively homogeneous); registers 8 kb .. 8 kb + 7 = K block kb
                lds_f32x4* const u1 = reinterpret_cast<lds_f32x4*>(tab + ((0 * 2 + pth) * 2 + half) * 16 * 4);
                lds_f32x4* const b1 = reinterpret_cast<lds_f32x4*>(tab + ((1 * 2 + pth) * 2 + half) * 16 * 4);
                p16_u32x4 hh[2], hl[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float t8[8];
#pragma unroll
                    for (int e4 = 0; e4 < 2; ++e4) {
                        const p16_f32x4 uu = u1[2 * kb + e4], bb = b1[2 * kb + e4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = fmaf(h[8 * kb + 4 * e4 + e], uu[e], bb[e]);
                            t8[4 * e4 + e] = fmaxf(t, t * M.sft_slope);                 // LeakyReLU for 0 <= slope <= 1 (host check)
                        }
                    }
                    const float ta[4] = {t8[0], t8[1], t8[2], t8[3]}, tb[4] = {t8[4], t8[5], t8[6], t8[7]};
                    unsigned Ha[2], La[2], Hb[2], Lb[2];
                    p16_split4(ta, one, Ha, La);
                    p16_split4(tb, one, Hb, Lb);
                    hh[kb] = p16_u32x4{Ha[0], Ha[1], Hb[0], Hb[1]};
                    hl[kb] = p16_u32x4{La[0], La[1], Lb[0], Lb[1]};
                }
                p16_f32x16 c = (p16_f32x16)(0.f);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const p16_f16x8 ah = __builtin_bit_cast(p16_f16x8, *reinterpret_cast<lds_u32x4*>(SB + P16_SFE_A2 + ((pth * 2 + kb) * 2 + 0) * 1024 + lane * 16));
                    const p16_f16x8 al = __builtin_bit_cast(p16_f16x8, *reinterpret_cast<lds_u32x4*>(SB + P16_SFE_A2 + ((pth * 2 + kb) * 2 + 1) * 1024 + lane * 16));
                    const p16_f16x8 xh = __builtin_bit_cast(p16_f16x8, hh[kb]), xl = __builtin_bit_cast(p16_f16x8, hl[kb]);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, c, 0, 0, 0);
                }
                lds_f32x4* const u2 = reinterpret_cast<lds_f32x4*>(tab + ((2 * 2 + pth) * 2 + half) * 16 * 4);
                lds_f32x4* const b2 = reinterpret_cast<lds_f32x4*>(tab + ((3 * 2 + pth) * 2 + half) * 16 * 4);
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const p16_f32x4 uu = u2[i4], bb = b2[i4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m = fmaf(c[4 * i4 + e], uu[e], bb[e]);
                        v[4 * i4 + e] = pth == 0 ? __fmul_rn(v[4 * i4 + e], m + 1.f) : __fadd_rn(v[4 * i4 + e], m);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- stored pre-split ----
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float o[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                unsigned Hh[2], Ll[2];
                p16_split4(o, M.out_scale, Hh, Ll);
                const p16_u32x4 unit = p16_unit(Hh, Ll);
                const unsigned y2off = ok ? (pix * (unsigned)M.y2_stride + (unsigned)(nb * 32)) * 4u + P16_UNIT_OFF(q, half) : P16_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(unit, y2rs, (int)y2off, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                               // one row at a time: interleaved, the two rows' operands double the register peak
        }
        // beyond fp16 (+-Inf included; a NaN slips through fmaxf and simply propagates to the pixels it reaches, as it does in the fp32 reference)
        // in the stored tensor or in the condition: the window is redone on the per-tile kernels (host).  The hidden
        // activations cannot overflow while the condition passes (their scales come from bounds over 2^-6 of fp16's range, see the packer)
        if (__builtin_amdgcn_ballot_w64(!(amax * M.out_scale <= 65504.f) || !(cmax * M.cond_scale <= 65504.f)) != 0ull && lane == 0) atomicOr(M.overflow + g, 1u);
        P16_TSTAMP(5);
        P16_TFLUSH();
        return;
    }
    float amax = 0.f;
#pragma unroll
    for (int ri = 0; ri < (UP ? 4 : 2); ++ri) {                              // accumulator ri = (column phase px, row r)
        const int r = ri & 1, pxe = ri >> 1;
        const int gy = UP ? 2 * (y0 + wv * 2 + r) + py : y0 + wv * 2 + r;
        if (gy >= H) continue;                                               // wave-uniform
        const int gxo = UP ? 2 * gx + pxe : gx;
        const bool ok = gxo < W;
        const unsigned pix = (unsigned)(gy * W + gxo);
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
                float t = fmaf(acc[ri][4 * q + e], us[q][e], bs[q][e]);
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
        // |v| 2^E beyond fp16 (+-Inf included; NaN is not detected -- fmaxf drops it -- and propagates to the pixels it reaches like in the fp32
        // reference): this window's frame is redone on the per-tile kernel (host).  Values of masked lanes (columns past
        // the image) come from zero-padded inputs like any other pixel's: harmless to include.
        if (__builtin_amdgcn_ballot_w64(!(amax * M.out_scale <= 65504.f)) != 0ull && lane == 0) atomicOr(M.overflow + g, 1u);
    }
    P16_TSTAMP(4);
#ifdef K4_P16_TIMING
    tacc[5] = ((__builtin_amdgcn_s_memtime() - mt0_) << 32) | (__builtin_amdgcn_s_memrealtime() - rt0_);       // the wave's life in both clocks
    tacc[6] = (rt0_ << 24) | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u) << 16) | (__builtin_amdgcn_s_getreg(4 | (15 << 11)) & 0xffffu);   // start (100 MHz) | XCC | HW_ID[15:0]
#endif
    P16_TFLUSH();
}
#ifdef K4_P16_TIMING
extern "C" int k4_debug_p16_timing(unsigned long long* out, int reset) {             // out: [P16_TIMING_WAVES][8]; slot 7 = number of passes of that wave index
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(k4_p16_timing), sizeof(unsigned long long) * 8 * P16_TIMING_WAVES);
    if (e == hipSuccess && reset) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(k4_p16_timing));
        if (e == hipSuccess) e = hipMemset(p, 0, sizeof(unsigned long long) * 8 * P16_TIMING_WAVES);
    }
    return (int)e;
}
#endif

extern "C" int64_t k4_conv_weight_p16_bytes(int32_t cout, int32_t cin) {
    if (cout <= 0 || cin <= 0 || (cout & 31) || (cin & 15)) return -1;
    return (int64_t)(cin / 16) * (cout / 32) * P16_W_BYTES + (int64_t)cout * 4;
}
extern "C" int64_t k4_conv_weight_p16_up2x_bytes(int32_t cout, int32_t cin) {       // the K4_PRE_UPSAMPLE2X operand: four phases of 2 x 2 taps
    if (cout <= 0 || cin <= 0 || (cout & 31) || (cin & 15)) return -1;
    return (int64_t)(cin / 16) * (cout / 32) * 4 * P16_W_BYTES_UP + (int64_t)cout * 4;
}

extern "C" int64_t k4_conv_sft_epilogue_bytes(int32_t channels) {                      // the epilogue-SFT operand of a layer with `channels` outputs
    if (channels <= 0 || (channels & 31)) return -1;
    return (int64_t)(channels / 32) * P16_SFE_BYTES;
}

static int conv3x3_p16(const k4_conv_job* jobs, const k4_conv_sft_job* sjobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                       const void* w_p16, const float* bias, int32_t cout, int32_t cout_stride,
                       uint32_t flags, float slope, int32_t res_stride, float res_scale,
                       int32_t cond_stride, float cond_scale, const void* w_sfe, float sft_slope, int32_t y2_stride,
                       float out_scale, uint32_t* overflow, void* stream) {
    const bool sft = sjobs != nullptr;
    if (!jobs || n_jobs <= 0 || n_jobs > K4_MAX_JOBS || !w_p16 || !bias) return K4_ERR_BAD_ARG;
    if (sft) {
        if (!w_sfe || (((size_t)w_sfe) & 15) || (flags & K4_PRE_UPSAMPLE2X) || cond_stride < 32 || (cond_stride & 3) || y2_stride < cout || (y2_stride & 15)) return K4_ERR_BAD_ARG;
        if (!(cond_scale > 0.f) || !(out_scale > 0.f) || !(sft_slope >= 0.f && sft_slope <= 1.f)) return K4_ERR_BAD_ARG;
    }
    if (cin <= 0 || (cin & 15) || cin_stride < cin || (cin_stride & 3) || cout <= 0 || (cout & 31) || cout_stride < cout || (cout_stride & 3)) return K4_ERR_BAD_ARG;
    if (flags & ~(K4_EPI_LRELU | K4_EPI_RES | K4_PRE_UPSAMPLE2X)) return K4_ERR_BAD_ARG;
    if ((flags & K4_EPI_LRELU) && !(slope >= 0.f && slope <= 1.f)) return K4_ERR_BAD_ARG;
    if ((flags & K4_EPI_RES) && (res_stride < cout || (res_stride & 3))) return K4_ERR_BAD_ARG;
    if (out_scale != 0.f && (!overflow || !(out_scale > 0.f))) return K4_ERR_BAD_ARG;
    if ((((size_t)w_p16) & 15) || (((size_t)bias) & 15)) return K4_ERR_BAD_ARG;
    P16Conv M{};
    M.n = n_jobs; M.cin = cin; M.cin_stride = cin_stride; M.w = w_p16; M.bias = bias; M.cout = cout; M.cout_stride = cout_stride;
    M.flags = flags; M.slope = slope; M.res_stride = res_stride; M.res_scale = res_scale; M.out_scale = out_scale; M.overflow = overflow; M.debug = k4_env().sr_debug;
    M.w_sfe = w_sfe; M.cond_stride = cond_stride; M.y2_stride = y2_stride; M.cond_scale = cond_scale; M.sft_slope = sft_slope;
    const int nbc = cout / 32;
    int total = 0;
    for (int g = 0; g < n_jobs; ++g) {
        const k4_conv_job& j = jobs[g];
        if (!j.x || (!j.y && !sft) || j.H <= 0 || j.W <= 0 || (((size_t)j.x | (size_t)j.y) & 15)) return K4_ERR_BAD_ARG;      // SFT: y may be NULL (only y2 is produced)
        if (sft) {
            if (!sjobs[g].cond || !sjobs[g].y2 || (((size_t)sjobs[g].cond) & 15) || (((size_t)sjobs[g].y2) & 63)) return K4_ERR_BAD_ARG;
            if ((long long)j.H * j.W * (cond_stride > y2_stride ? cond_stride : y2_stride) * 4 >= 0x80000000LL) return K4_ERR_UNSUPPORTED;
            M.cond[g] = sjobs[g].cond; M.y2[g] = sjobs[g].y2;
        }
        if ((flags & K4_EPI_RES) && (!j.res || (((size_t)j.res) & 15))) return K4_ERR_BAD_ARG;
        if ((flags & K4_PRE_UPSAMPLE2X) && ((j.H & 1) || (j.W & 1))) return K4_ERR_BAD_ARG;
        const long long strd = cin_stride > cout_stride ? cin_stride : cout_stride;
        if ((long long)j.H * j.W * (strd > res_stride ? strd : res_stride) * 4 >= 0x80000000LL) return K4_ERR_UNSUPPORTED;      // 32-bit buffer offsets
        M.x[g] = j.x; M.y[g] = j.y; M.res[g] = j.res; M.H[g] = j.H; M.W[g] = j.W;
        const bool up = (flags & K4_PRE_UPSAMPLE2X) != 0;              // the tile grid lives on the INPUT image; two workgroups (row phases) per tile
        const int gw = up ? j.W / 2 : j.W, gh = up ? j.H / 2 : j.H;
        M.tiles_x[g] = (gw + 31) / 32;
        total += M.tiles_x[g] * ((gh + 7) / 8) * nbc * (up ? 2 : 1);
        M.blk_end[g] = total;
    }
    M.total = total;
    const dim3 grid((unsigned)total), block(256);
    if (sft) {
        hipLaunchKernelGGL((k4_conv_p16_kernel<false, false, true>), grid, block, 0, (hipStream_t)stream, M);
        return k4_check_launch();
    }
    if (flags & K4_PRE_UPSAMPLE2X) {
        if (out_scale != 0.f) hipLaunchKernelGGL((k4_conv_p16_kernel<true, true>), grid, block, 0, (hipStream_t)stream, M);
        else hipLaunchKernelGGL((k4_conv_p16_kernel<false, true>), grid, block, 0, (hipStream_t)stream, M);
    } else if (out_scale != 0.f) hipLaunchKernelGGL((k4_conv_p16_kernel<true, false>), grid, block, 0, (hipStream_t)stream, M);
    else hipLaunchKernelGGL((k4_conv_p16_kernel<false, false>), grid, block, 0, (hipStream_t)stream, M);
    return k4_check_launch();
}

extern "C" int k4_conv3x3_p16_multi(const k4_conv_job* jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                                    const void* w_p16, const float* bias, int32_t cout, int32_t cout_stride,
                                    uint32_t flags, float slope, int32_t res_stride, float res_scale,
                                    float out_scale, uint32_t* overflow, void* stream) {
    return conv3x3_p16(jobs, nullptr, n_jobs, cin, cin_stride, w_p16, bias, cout, cout_stride, flags, slope, res_stride, res_scale,
                       0, 0.f, nullptr, 0.f, 0, out_scale, overflow, stream);
}

extern "C" int k4_conv3x3_p16_sft_multi(const k4_conv_job* jobs, const k4_conv_sft_job* sft_jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                                        const void* w_p16, const float* bias, int32_t cout, int32_t cout_stride,
                                        uint32_t flags, float slope, int32_t res_stride, float res_scale,
                                        int32_t cond_stride, float cond_scale, const void* w_sfe, float sft_slope, int32_t y2_stride,
                                        float out_scale, uint32_t* overflow, void* stream) {
    if (!sft_jobs || !overflow) return K4_ERR_BAD_ARG;
    return conv3x3_p16(jobs, sft_jobs, n_jobs, cin, cin_stride, w_p16, bias, cout, cout_stride, flags, slope, res_stride, res_scale,
                       cond_stride, cond_scale, w_sfe, sft_slope, y2_stride, out_scale, overflow, stream);
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
