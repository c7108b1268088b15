// Do 16-byte buffer loads / stores with a compile-time channel offset in the SCALAR offset operand (an SGPR for values > 64) address
// base + lane offset + scalar offset for every lane?  (k4_sft_b6p_kernel first passed (mb2 * 32 + 8 * q) * 4 bytes there.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int MODE>
__global__ void k(const unsigned* x, unsigned* y, int n_pix, int stride) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int pix = blockIdx.x * 32 + l31;
    const __amdgpu_buffer_rsrc_t xr = rsrc(x, n_pix * stride * 4), yr = rsrc(y, n_pix * stride * 4);
    const unsigned off = pix < n_pix ? (unsigned)(pix * stride * 4 + half * 16) : 0x80000000u;
    u32x4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        q[i] = MODE == 0 ? __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, i * 32, 0)
                         : __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(off + i * 32), 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(q[i], yr, (int)off, i * 32, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(q[i], yr, (int)(off + i * 32), 0, 0);
    }
}
int main() {
    const int n = 1000, stride = 64;
    std::vector<unsigned> h(n * stride), o(n * stride);
    for (int i = 0; i < n * stride; ++i) h[i] = i * 2654435761u + 12345u;
    unsigned *x, *y;
    hipMalloc(&x, h.size() * 4); hipMalloc(&y, h.size() * 4);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(y, 0, h.size() * 4);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((n + 31) / 32), dim3(64), 0, 0, x, y, n, stride);
        else hipLaunchKernelGGL(k<1>, dim3((n + 31) / 32), dim3(64), 0, 0, x, y, n, stride);
        hipMemcpy(o.data(), y, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        int per_chan16[4] = {0, 0, 0, 0};
        for (int i = 0; i < n * stride; ++i)
            if (o[i] != h[i]) { if (first < 0) first = i; ++bad; ++per_chan16[(i % stride) / 16]; }
        printf("mode %d (%s): %d of %d words differ, first at pixel %d channel %d; by 16-channel group %d %d %d %d\n", mode,
               mode == 0 ? "scalar offset operand" : "lane offset", bad, n * stride, first < 0 ? -1 : first / stride, first < 0 ? -1 : first % stride,
               per_chan16[0], per_chan16[1], per_chan16[2], per_chan16[3]);
    }
    return 0;
}
