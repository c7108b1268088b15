// Do VALU instructions issue while a wave's (or another wave's) MFMAs occupy the matrix pipe of a SIMD?  gfx950.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
// Per loop iteration a wave issues 4 independent v_mfma_f32_32x32x16_f16 (4 accumulators) and NV independent v_fma_f32 on other registers.
// If the two overlap, cycles per iteration stay at 4 x 32 until NV x (VALU issue cycles) exceeds it; if they serialise, they add.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.25f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) { const int q = (i * (NV / 4) + j) & 7; v[q] = __builtin_fmaf(v[q], c1, c2); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, bool MF>
static void run(int wg_per_cu, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<NV, MF>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, MF>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%s 4 MFMA + %3d VALU per iteration, %d wave(s) per SIMD: %7.1f s_memtime ticks, %7.1f ns per iteration of one wave (kernel time / iterations)\n", MF ? "   " : "no ", NV, wg_per_cu, (double)c / iters, ms * 1e6 / iters);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 3; w += 2) {
        run<0, true>(w, out, cyc); run<8, true>(w, out, cyc); run<16, true>(w, out, cyc); run<32, true>(w, out, cyc); run<64, true>(w, out, cyc);
        run<16, false>(w, out, cyc); run<32, false>(w, out, cyc); run<64, false>(w, out, cyc);
    }
    return 0;
}
