// Micro-benchmark (measurement tool, not part of the product): sustained rate of v_mfma_f32_32x32x16_bf16 on gfx950 for the
// instruction mixes of the decoder's convolution kernel -- dependent / independent accumulator chains, 1 or 2 waves per SIMD,
// with and without the 3 ds_read_b128 per 6 MFMAs of the A-fragment ring.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // bit0: independent accumulators, bit1: ds_reads, bit2: 12 MFMAs per A fragment (two B sets)
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, int lds_pad) {
    extern __shared__ uint4 smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) smem[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    f32x16 acc[4];
    for (int r = 0; r < 4; ++r) acc[r] = (f32x16)(0.f);
    uint4 a[3][3], b[3], b2[3];
    for (int q = 0; q < 3; ++q) { b[q] = smem[lane + q * 64]; b2[q] = smem[lane + q * 64 + 200]; for (int s = 0; s < 3; ++s) a[s][q] = smem[lane + 64 * (q + 3 * s)]; }
    const uint4* ap = smem + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 36; ++u) {
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < 3; ++q) a[(u + 2) % 3][q] = ap[q * 1224 + (u % 7) * 34];
            }
            __builtin_amdgcn_sched_barrier(0);
#define MF(R, A, B) acc[R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[R], 0, 0, 0)
            const int r = u % 4;
            if (MODE & 1) {
                MF(0, a[u % 3][2], b[0]); MF(1, a[u % 3][0], b[2]); MF(2, a[u % 3][1], b[1]);
                MF(3, a[u % 3][1], b[0]); MF(0, a[u % 3][0], b[1]); MF(1, a[u % 3][0], b[0]);
            } else {
                MF(r, a[u % 3][2], b[0]); MF(r, a[u % 3][0], b[2]); MF(r, a[u % 3][1], b[1]);
                MF(r, a[u % 3][1], b[0]); MF(r, a[u % 3][0], b[1]); MF(r, a[u % 3][0], b[0]);
            }
            if (MODE & 4) {
                const int r2 = (u + 2) % 4;
                MF(r2, a[u % 3][2], b2[0]); MF(r2, a[u % 3][0], b2[2]); MF(r2, a[u % 3][1], b2[1]);
                MF(r2, a[u % 3][1], b2[0]); MF(r2, a[u % 3][0], b2[1]); MF(r2, a[u % 3][0], b2[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 4; ++r) for (int e = 0; e < 16; ++e) s += acc[r][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int wgs_per_cu, float* out) {
    const int iters = 200;
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 70 * 1024;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, out, iters, 0);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, out, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double mfma = (double)grid * 4 * iters * 36 * ((MODE & 4) ? 12 : 6);
    const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-58s %d WG/CU: %8.3f ms  %7.1f TFLOP/s bf16 = %.3f of 2500;  cycles per MFMA per SIMD at 2.4 GHz: %.1f\n", name, wgs_per_cu, ms, tf, tf / 2500.0,
           ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
}

int main() {
    float* out; hipMalloc(&out, 1 << 22);
    run<0>("dependent chains of 6, no LDS", 1, out);
    run<0>("dependent chains of 6, no LDS", 2, out);
    run<1>("independent accumulators, no LDS", 1, out);
    run<1>("independent accumulators, no LDS", 2, out);
    run<2>("dependent chains of 6 + 3 ds_read_b128 per 6 MFMA", 1, out);
    run<2>("dependent chains of 6 + 3 ds_read_b128 per 6 MFMA", 2, out);
    run<3>("independent + 3 ds_read_b128 per 6 MFMA", 2, out);
    run<6>("dependent chains, 3 ds_read_b128 per 12 MFMA", 2, out);
    run<4>("dependent chains, 12 MFMA per step, no LDS", 2, out);
    hipDeviceSynchronize();
    return 0;
}
