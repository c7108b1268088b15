// Microbenchmark: cycles per v_mfma_f32_32x32x16_{f16,bf16} as a function of the number of independent accumulators a wave rotates
// through (1 = every MFMA depends on the previous one) and of the waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_dep.hip -o mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x16)(0.f);
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    const bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, bool BF>
static void run(int wg_per_cu, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BF>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BF>), dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)iters * 12;
    printf("%s accumulators %d, %d wave(s) per SIMD: %.1f s_memtime ticks per MFMA per wave, %.1f ns per MFMA per SIMD -> %.0f TFLOP/s chip\n",
           BF ? "bf16" : "f16 ", NACC, wg_per_cu, (double)c / n, ms * 1e6 / (n * wg_per_cu), 2.0 * 32 * 32 * 16 * n * wg_per_cu * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 3; ++w) {
        run<1, false>(w, out, cyc); run<2, false>(w, out, cyc); run<3, false>(w, out, cyc); run<4, false>(w, out, cyc); run<6, false>(w, out, cyc);
    }
    run<1, true>(1, out, cyc); run<2, true>(1, out, cyc); run<4, true>(1, out, cyc);
    return 0;
}
