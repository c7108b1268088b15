#include <hip/hip_runtime.h>
// hi/lo fp16 split of 4 scaled floats with v_fma_mix: 8 instructions
__device__ __forceinline__ void split4(const float4& v, float sc, uint2& hi, uint2& lo) {
    unsigned h0, h1, l0, l1;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v.x), "v"(sc));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v.y), "v"(sc));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v.z), "v"(sc));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v.w), "v"(sc));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "v"(sc), "v"(h0));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "v"(sc), "v"(h0));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "v"(sc), "v"(h1));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "v"(sc), "v"(h1));
    hi = make_uint2(h0, h1); lo = make_uint2(l0, l1);
}
__global__ void k(const float4* x, float sc, uint2* hi, uint2* lo) {
    uint2 h, l;
    split4(x[threadIdx.x], sc, h, l);
    hi[threadIdx.x] = h; lo[threadIdx.x] = l;
}
// reference form
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void kref(const float4* x, float sc, uint2* hi, uint2* lo) {
    const float4 v = x[threadIdx.x];
    const float f[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    _Float16 h[4], l[4];
    for (int i = 0; i < 4; ++i) { h[i] = (_Float16)f[i]; l[i] = (_Float16)(f[i] - (float)h[i]); }
    h2 h01 = {h[0], h[1]}, h23 = {h[2], h[3]}, l01 = {l[0], l[1]}, l23 = {l[2], l[3]};
    hi[threadIdx.x] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    lo[threadIdx.x] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}
#include <cstdio>
#include <cmath>
#include <vector>
int main() {
    const int n = 256;
    std::vector<float> x(n * 4);
    for (int i = 0; i < n * 4; ++i) x[i] = (float)((i * 2654435761u % 100003) / 50001.5 - 1.0) * powf(2.f, (float)((i * 7) % 30 - 15));
    x[5] = 0.f; x[6] = -0.f; x[7] = 1e-30f; x[8] = INFINITY; x[9] = NAN; x[10] = 65504.f; x[11] = 3e-8f;
    float4* dx; uint2 *dh, *dl, *rh, *rl;
    hipMalloc(&dx, n * 16); hipMalloc(&dh, n * 8); hipMalloc(&dl, n * 8); hipMalloc(&rh, n * 8); hipMalloc(&rl, n * 8);
    hipMemcpy(dx, x.data(), n * 16, hipMemcpyHostToDevice);
    int bad = 0;
    for (float sc : {1.f, 0.25f, 16384.f, 1.52587890625e-05f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, dx, sc, dh, dl);
        hipLaunchKernelGGL(kref, dim3(1), dim3(n), 0, 0, dx, sc, rh, rl);
        std::vector<unsigned> a(n * 2), b(n * 2), c(n * 2), d(n * 2);
        hipMemcpy(a.data(), dh, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), rh, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dl, n * 8, hipMemcpyDeviceToHost); hipMemcpy(d.data(), rl, n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n * 2; ++i) if (a[i] != b[i] || c[i] != d[i]) { if (bad < 8) printf("sc %g i %d hi %08x ref %08x lo %08x ref %08x\n", sc, i, a[i], b[i], c[i], d[i]); ++bad; }
    }
    printf("mismatches %d\n", bad);
    return 0;
}
