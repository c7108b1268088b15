// Shared device helpers for the gfx950 kernels of lib4k_hip.so.  CDNA4 only: wave64, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <vector>
#include "k4nerf.h"

#define K4_WAVE 64

// Uniform (wave-invariant) read-only data -- MLP weights, ray tables read with a wave-uniform index --
// goes through the constant address space so that hipcc emits s_load_* (scalar cache, SGPR operands for
// v_fmac) instead of 64 identical vector loads.
typedef const float __attribute__((address_space(4)))* k4_cptr;
__device__ __forceinline__ k4_cptr k4_const(const float* p) {
    return (k4_cptr)(uintptr_t)p;
}

__device__ __forceinline__ int k4_lane() { return (int)(threadIdx.x & 63u); }

// number of set bits of `m` below this lane (wave64 prefix popcount)
__device__ __forceinline__ int k4_prefix(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// index of the run of equal `key` values this lane belongs to (runs = maximal groups of consecutive lanes)
__device__ __forceinline__ int k4_run_id(int key, int lane) {
    const int prev = __shfl_up(key, 1);
    const uint64_t heads = __ballot(lane == 0 || prev != key);
    return __popcll(heads & (~0ull >> (63 - lane)));
}

// LDS accumulate (ds_add_f32 / ds_add_f64 / ds_add_u64, no return value)
__device__ __forceinline__ void k4_lds_add(unsigned long long* p, unsigned long long v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void k4_lds_add(float* p, float v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void k4_lds_add(double* p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float k4_readlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// 64-bit fetch of two adjacent floats that are only 4-byte aligned (global_load_dwordx2 needs dword alignment only)
struct __attribute__((packed, aligned(4))) k4_f2u { float x, y; };
__device__ __forceinline__ float2 k4_ld2(const float* p) {
    const k4_f2u v = *reinterpret_cast<const k4_f2u*>(p);
    return make_float2(v.x, v.y);
}

// C round(): halves away from zero (the reference's maskcache_lookup, render_utils_kernel.cu:385-387)
__device__ __forceinline__ int k4_round_half_away(float x) { return (int)roundf(x); }

// bijective XCD-aware block remap (hardware places block b on XCD b%8; give each XCD a contiguous
// range of logical workgroups so neighbours share that XCD's L2).  Speed only, never correctness.
__device__ __forceinline__ int k4_xcd_remap(int b, int nwg) {
    const int xcd = b & 7, idx = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// F.grid_sample(bilinear, align_corners=True) coordinate, restating the reference's op sequence:
//   ind_norm = ((p - min) / (max - min)) * 2 - 1          (lib/grid.py:123)
//   u        = ((ind_norm + 1) / 2) * (size - 1)          (grid_sampler unnormalize, align_corners)
__device__ __forceinline__ float k4_norm_coord(float p, float lo, float hi) {
    return ((p - lo) / (hi - lo)) * 2.f - 1.f;
}
// Same value, 3 VALU instead of ~10: correctly rounded a/b from the correctly rounded reciprocal y = RN(1/b) of a
// wave-uniform divisor (Markstein: q0 = RN(a*y); r = a - b*q0 exactly (FMA); q = RN(q0 + r*y) is RN(a/b)).
__device__ __forceinline__ float k4_div_by(float a, float b, float y) {
    const float q0 = a * y;
    const float r = fmaf(-b, q0, a);
    return fmaf(r, y, q0);
}
__device__ __forceinline__ float k4_norm_coord_r(float p, float lo, float len, float rlen) {
    return k4_div_by(p - lo, len, rlen) * 2.f - 1.f;
}
__device__ __forceinline__ float k4_unnorm(float n, int size) {
    return ((n + 1.f) / 2.f) * (float)(size - 1);
}

// Same, with the logical range cut in bands of `band` blocks dealt round-robin to the XCDs (band r*8+x -> XCD x): every XCD gets a
// sample of the whole range instead of one contiguous eighth (load balance) while `band` consecutive blocks still share an L2.
__device__ __forceinline__ int k4_xcd_remap_banded(int b, int nwg, int band) {
    if (band <= 0) return k4_xcd_remap(b, nwg);
    const int full = nwg / (8 * band) * (8 * band);
    if (b >= full) return full + k4_xcd_remap(b - full, nwg - full);
    const int xcd = b & 7, idx = b >> 3;
    return ((idx / band) * 8 + xcd) * band + idx % band;
}

// The 8 trilinear corner weights in PyTorch's naming/order (tnw,tne,tsw,tse,bnw,bne,bsw,bse) with
// torch's ix<->our z (W axis), iy<->y (H), iz<->x (D):  t/b = x lo/hi, n/s = y lo/hi, w/e = z lo/hi.
struct K4Tri {
    int x0, y0, z0;
    float w[8];
};
__device__ __forceinline__ K4Tri k4_tri_setup(float ux, float uy, float uz) {
    K4Tri t;
    const float fx = floorf(ux), fy = floorf(uy), fz = floorf(uz);
    t.x0 = (int)fx; t.y0 = (int)fy; t.z0 = (int)fz;
    const float xl = (fx + 1.f) - ux, xh = ux - fx;
    const float yl = (fy + 1.f) - uy, yh = uy - fy;
    const float zl = (fz + 1.f) - uz, zh = uz - fz;
    t.w[0] = zl * yl * xl; t.w[1] = zh * yl * xl; t.w[2] = zl * yh * xl; t.w[3] = zh * yh * xl;
    t.w[4] = zl * yl * xh; t.w[5] = zh * yl * xh; t.w[6] = zl * yh * xh; t.w[7] = zh * yh * xh;
    return t;
}
// corner c -> (dx,dy,dz)
#define K4_CX(c) (((c) >> 2) & 1)
#define K4_CY(c) (((c) >> 1) & 1)
#define K4_CZ(c) ((c) & 1)

// Launch tapes (k4_tape.hip; include/k4nerf.h k4_tape_*).  A recordable entry point is `return k4_taped(stream, [=](void* stream) -> int { body });`:
// the body runs as always; while this thread records, a copy of the closure (the call's arguments BY VALUE -- a body must not keep a pointer
// to a caller's host struct or array) is appended to the tape first.  Calls made from inside a recordable body are not recorded again.
// A call whose stream is the recording's main stream (k4_tape_begin) follows the replaying stream; a call placed on any other stream (a side
// stream's weight gradients) stays on the stream it was recorded with.
struct k4_tape_op { std::function<int(void*)> fn; void* pinned; bool follow; };
struct k4_tape { std::vector<k4_tape_op> ops; void* main_stream; };
extern thread_local k4_tape* k4_tape_rec;
extern thread_local int k4_tape_depth;
template <class F> static inline int k4_taped(void* stream, F&& f) {
    if (k4_tape_rec && k4_tape_depth == 0) k4_tape_rec->ops.push_back(k4_tape_op{std::function<int(void*)>(f), stream, stream == k4_tape_rec->main_stream});
    ++k4_tape_depth;
    const int rc = f(stream);
    --k4_tape_depth;
    return rc;
}

static inline int k4_check_launch() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

// ---- host-side process state: none that can change after load --------------------------------------------------------
// Experiment / profiling knobs are environment variables read ONCE while the library is being loaded (a namespace-scope
// constant in k4_march.hip); launches never call getenv.  Per-device facts (CU count, "dynamic LDS attribute already raised
// for kernel F") are cached in fixed arrays indexed by the HIP device ordinal; concurrent first calls write the same values.
struct K4Env {
    int geom_skip;       // K4_GEOM_SKIP    (1) 0: do not use the coarse occupancy summary (A/B of the empty-space skipping)
    int debug;           // K4_DEBUG        (0) ablation bits of the marcher kernels, profiling only
    int sr_debug;        // K4_SR_DEBUG     (0) profiling bits of the decoder kernels (1: input channel stride 0 = no memory traffic, WRONG results; 2..16: phases of the 3x3 kernel off; 32: SFT layers on the unpipelined kernel; exact ones: 2048: row kernel instead of the K-split 3x3 kernel on small images, 4096: per-tap weight-gradient kernel instead of the nine-tap one)
    bool no_fast_shade;  // K4_DEBUG & 1024: the shading kernel's general path on shapes the FAST path covers (A/B, tests: identical outputs)
};
// Settled by measurement and no longer switchable (the evidence is in profiles/ and DESIGN.md): geometry kernel bounded for 5 waves per
// SIMD (6 spilled), serpentine ray order inside an 8x8 tile, one row of workgroup tiles per XCD band, persistent shading grid of 2 workgroups
// per CU (K4_SHADE_WG_PER_CU, a compile-time constant), small-launch tile rule of the 3x3 convolution on, 16-row tiles for the 3-term kernel and 8-row tiles for the
// 2-term (f16x3) kernel beyond it (8-row 37.1 ms per 4K frame, 12-row 39.3, 16-row 38.5-38.8).
const K4Env& k4_env();
#define K4_MAX_DEVICES 64
static inline int k4_device_ordinal() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= K4_MAX_DEVICES) dev = 0;
    return dev;
}
int k4_num_cus();
// raise hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device)
#define K4_ENSURE_DYN_LDS(KERN, BYTES) do { \
        static bool k4_attr_done_[K4_MAX_DEVICES]; \
        const int k4_dev_ = k4_device_ordinal(); \
        if ((BYTES) > 64 * 1024 && !k4_attr_done_[k4_dev_]) { \
            hipError_t k4_e_ = hipFuncSetAttribute((const void*)(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)); \
            if (k4_e_ != hipSuccess) return (int)k4_e_; \
            k4_attr_done_[k4_dev_] = true; } } while (0)
