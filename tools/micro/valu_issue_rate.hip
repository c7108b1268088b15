// How fast does ONE gfx950 SIMD issue vector instructions, as a function of the waves resident on it?  (round 5: the marcher's
// shading kernel is sized against this: ~600 vector instructions + 60 MFMAs per 32-sample tile)
//   hipcc --offload-arch=gfx950 -O3 valu_issue_rate.hip -o valu_issue_rate && ./valu_issue_rate
// One workgroup per CU (100 KB of LDS), W waves per SIMD; every wave runs ITER x 32 independent instructions of one class (8 register
// chains, so no dependent-issue stalls) and stamps s_memtime around it.  Reported: SIMD cycles per wave-instruction = cycles / (W * n).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
enum { FMA, PKFMA, CVTPK, ANDB, MAXI, LSHL, PKADD, MOV, FMAC_MIX, SALU, MIX_SV, N_OPS };
static const char* NAME[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_and_b32", "v_max_i32", "v_lshlrev_b32", "v_pk_add_f32", "v_mov_b32",
                                  "v_fma_f32 x v_and alternating", "s_add_u32", "1 s_add_u32 per v_fma_f32"};
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* cyc, float* sink, int iters) {
    __shared__ float pad[25 * 1024];
    if (threadIdx.x == 0) pad[0] = 0.f;
    __syncthreads();
    float a[8]; double d[8]; unsigned s0 = blockIdx.x, s1 = 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; d[i] = a[i]; }
    const float c1 = 1.0001f, c2 = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                else if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[i]));
                else if (OP == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                else if (OP == ANDB) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                else if (OP == MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                else if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                else if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(d[i]));
                else if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c1));
                else if (OP == FMAC_MIX) { if (i & 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2)); }
                else if (OP == SALU) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1));
                else if (OP == MIX_SV) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1)); }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)s0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
    if (s == 12345.678f) sink[0] = s + pad[threadIdx.x];
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}
template <int OP>
static void run(unsigned long long* dc, float* sink) {
    const int iters = 2000;
    for (int W : {1, 2, 4}) {
        hipMemset(dc, 0, 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * W), 0, 0, dc, sink, 10);      // warm
        hipMemset(dc, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * W), 0, 0, dc, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        const double waves = 256.0 * 4 * W, per_wave = (double)c / waves;
        const double n = (double)iters * 32 * (OP == MIX_SV ? 2 : 1);
        printf("%-34s W=%d waves/SIMD: %.2f cycles per instruction per wave, %.2f SIMD cycles per wave-instruction (wall %.3f ms => %.2f GHz)\n",
               NAME[OP], W, per_wave / n, per_wave / n / W, ms, per_wave / (ms * 1e6));
    }
}
int main() {
    unsigned long long* dc; float* sink;
    hipMalloc(&dc, 8); hipMalloc(&sink, 4096 * 4);
    run<FMA>(dc, sink); run<PKFMA>(dc, sink); run<CVTPK>(dc, sink); run<ANDB>(dc, sink); run<MAXI>(dc, sink); run<LSHL>(dc, sink);
    run<PKADD>(dc, sink); run<MOV>(dc, sink); run<FMAC_MIX>(dc, sink); run<SALU>(dc, sink); run<MIX_SV>(dc, sink);
    return 0;
}
