// Launch tapes (include/k4nerf.h, k4_tape_*): the decoder's training pass as ONE native call.
//
// The joint training iteration (run_sr.py:869-1014) evaluates SFTNet on a 64x64 patch: ~450 launches of 4-30 us per iteration whose issue
// cost -- one Python-to-C transition, argument marshalling and, under autograd, a node per fused block -- paced the iteration (10.5 ms of
// host time against 9.3 ms of kernels on the main stream).  hipGraphs were measured slower than issuing the launches (a replayed
// kernel node costs more than a stream launch on this runtime, and the capture gives up the direct gradient hand-over).  A tape is the same
// idea without the graph runtime: while a thread records, every recordable entry point of this library appends a copy of its arguments
// (plain pointers and sizes: the C ABI has nothing else) and runs as usual; k4_tape_replay issues the recorded calls again, in order, from C++
// -- the same launches on the same buffers with ~2 us of host time each.  The caller keeps every buffer a tape touches alive and at the same
// address (lib/sr_tape.py: one set of activation / gradient buffers per patch shape).
//
// Recordable entry points wrap their body in k4_taped (k4_common.h).  An entry point that calls other recordable entry points (k4_rdb_train_*)
// is recorded once, as itself (k4_tape_depth).
#include "k4_common.h"

thread_local k4_tape* k4_tape_rec = nullptr;
thread_local int k4_tape_depth = 0;

extern "C" k4_tape* k4_tape_begin(void* main_stream) {
    if (k4_tape_rec) return nullptr;                       // one recording per thread at a time
    k4_tape_rec = new k4_tape();
    k4_tape_rec->main_stream = main_stream;
    return k4_tape_rec;
}
extern "C" int k4_tape_end(k4_tape* t) {
    if (!t || t != k4_tape_rec) return K4_ERR_BAD_ARG;
    k4_tape_rec = nullptr;
    return K4_OK;
}
extern "C" int64_t k4_tape_length(const k4_tape* t) { return t ? (int64_t)t->ops.size() : -1; }
extern "C" int k4_tape_replay(const k4_tape* t, void* stream) {
    if (!t || t == k4_tape_rec) return K4_ERR_BAD_ARG;      // (a tape still being recorded is not replayable)
    for (const auto& op : t->ops) {
        const int rc = op.fn(op.follow ? stream : op.pinned);
        if (rc != 0) return rc;
    }
    return K4_OK;
}
extern "C" void k4_tape_free(k4_tape* t) {
    if (t && t == k4_tape_rec) k4_tape_rec = nullptr;
    delete t;
}

// ---------------------------------------------------------------------------------------------------------------------
// The elementwise glue of SFTNet's training graph that PyTorch ops (and the autograd engine's gradient sums) supplied between the fused
// Functions of lib/sr_train.py: nearest x2 upsampling (lib/sr_esrnet.py:461-463) with its backward, and the sum of two gradients.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_f32(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, int64_t n4,
                                                 const float* __restrict__ as, const float* __restrict__ bs, float* __restrict__ os, int tail) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        const float4 x = a[i], y = b[i];
        out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    } else if (i - n4 < tail) {
        const int64_t j = n4 * 4 + (i - n4);
        os[j] = as[j] + bs[j];
    }
}
extern "C" int k4_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    return k4_taped(stream, [=](void* stream) -> int {
        if (n < 0 || (n > 0 && (!a || !b || !out)) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15u)) return K4_ERR_BAD_ARG;
        if (n == 0) return K4_OK;
        const int64_t n4 = n / 4;
        const int tail = (int)(n - n4 * 4);
        const int64_t threads = n4 + tail;
        if ((threads + 255) / 256 > 0x7fffffffLL) return K4_ERR_BAD_ARG;
        hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                           reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n4, a, b, out, tail);
        return k4_check_launch();
    });
}

// y[2Y + py][2X + px][c] = x[Y][X][c]: one thread per float4 of the INPUT, four stores
__global__ __launch_bounds__(256) void k_up2x(const float4* __restrict__ x, int H, int W, int c4, float4* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W * c4) return;
    const int c = (int)(i % c4);
    const int64_t p = i / c4;
    const int X = (int)(p % W), Y = (int)(p / W);
    const float4 v = x[i];
    float4* o = y + ((int64_t)(2 * Y) * (2 * W) + 2 * X) * c4 + c;
    o[0] = v; o[c4] = v;
    o += (int64_t)2 * W * c4;
    o[0] = v; o[c4] = v;
}
// gx[Y][X][c] = (gy[2Y][2X] + gy[2Y][2X+1]) + (gy[2Y+1][2X] + gy[2Y+1][2X+1]): the backward of the two repeat_interleave ops, W pairs first
__global__ __launch_bounds__(256) void k_up2x_bwd(const float4* __restrict__ gy, int H, int W, int c4, float4* __restrict__ gx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)H * W * c4) return;
    const int c = (int)(i % c4);
    const int64_t p = i / c4;
    const int X = (int)(p % W), Y = (int)(p / W);
    const float4* s = gy + ((int64_t)(2 * Y) * (2 * W) + 2 * X) * c4 + c;
    const float4 a = s[0], b = s[c4];
    s += (int64_t)2 * W * c4;
    const float4 d = s[0], e = s[c4];
    gx[i] = make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
}
static int up2x_check(const void* a, const void* b, int H, int W, int C) {
    if (!a || !b || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (((uintptr_t)a | (uintptr_t)b) & 15u)) return K4_ERR_BAD_ARG;
    if (((int64_t)H * W * (C / 4) + 255) / 256 > 0x7fffffffLL) return K4_ERR_BAD_ARG;
    return K4_OK;
}
extern "C" int k4_upsample2x_nhwc(const float* x, int32_t H, int32_t W, int32_t channels, float* y, void* stream) {
    return k4_taped(stream, [=](void* stream) -> int {
        const int rc = up2x_check(x, y, H, W, channels);
        if (rc) return rc;
        const int64_t n = (int64_t)H * W * (channels / 4);
        hipLaunchKernelGGL(k_up2x, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), H, W, channels / 4,
                           reinterpret_cast<float4*>(y));
        return k4_check_launch();
    });
}
extern "C" int k4_upsample2x_bwd_nhwc(const float* grad_y, int32_t H, int32_t W, int32_t channels, float* grad_x, void* stream) {
    return k4_taped(stream, [=](void* stream) -> int {
        const int rc = up2x_check(grad_y, grad_x, H, W, channels);
        if (rc) return rc;
        const int64_t n = (int64_t)H * W * (channels / 4);
        hipLaunchKernelGGL(k_up2x_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(grad_y), H, W, channels / 4,
                           reinterpret_cast<float4*>(grad_x));
        return k4_check_launch();
    });
}
