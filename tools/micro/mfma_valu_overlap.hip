// Do vector instructions issue while MFMAs occupy the matrix pipe of a gfx950 SIMD?  Three questions, all answered in CYCLES (s_memtime,
// per wave) so that the chip's power-dependent clock drops out; wall time beside them.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
//  (1) ONE wave per SIMD, an instruction stream of 1 v_mfma_f32_32x32x16_f16 + VPM vector instructions, interleaved and pinned (asm volatile):
//      cycles per MFMA as VPM grows.  Overlap => flat at 32 until VPM x issue cost > 32; serial => 32 + VPM x cost.
//  (2) SPECIALISED waves on one SIMD (MI355X_MICROARCH.md "Wave scheduling"): wave A issues MFMAs only, wave B (and C) vector instructions
//      only.  Waves find their SIMD with s_getreg HW_ID and take roles per SIMD through an LDS ticket, so the pairing does not depend on
//      the dispatcher's placement.  Overlap => each runs at its stand-alone pace; serial => each is slowed by the other's issue time.
//  (3) the same for the instruction classes the 3x3 decoder kernel's staging is made of: v_fma_f32, integer add, v_cndmask, v_fma_mix
//      (the hi/lo split), v_max3_u32 (chunk maximum), ds_write_b64 / ds_read_b128 (LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { OP_FMA = 0, OP_IADD = 1, OP_CNDMASK = 2, OP_FMAMIX = 3, OP_MAX3 = 4, OP_DSWRITE = 5, OP_DSREAD = 6, OP_MOV = 7, OP_PKFMA = 8, N_OPS = 9 };
static const char* OP_NAME[N_OPS] = {"v_fma_f32", "v_add_u32", "v_cndmask_b32", "v_fma_mixlo_f16", "v_max3_u32", "ds_write_b64", "ds_read_b128", "v_mov_b32", "v_pk_fma_f32"};

template <int OP>
__device__ __forceinline__ void valu(float& v, float& w, float c1, float c2, unsigned lds_off, f32x4& rd, double& dd) {
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2));
    else if (OP == OP_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(c1));
    else if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(c1));
    else if (OP == OP_FMAMIX) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(v) : "v"(w), "v"(c1), "v"(c2));
    else if (OP == OP_MAX3) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(v) : "v"(c1), "v"(c2));
    else if (OP == OP_DSWRITE) asm volatile("ds_write_b64 %0, %1" :: "v"(lds_off), "v"(dd) : "memory");
    else if (OP == OP_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(rd) : "v"(lds_off) : "memory");
    else if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(c1));
    else if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(dd));
}

__device__ __forceinline__ void mfma(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// role of a wave: bit 0 = issues MFMAs, bit 1 = issues vector instructions.  roles[rank on its SIMD] (rank by LDS ticket).
template <int VPM, int OP>
__global__ __launch_bounds__(768) void k(float* out, int iters, unsigned long long* cyc, const int* roles, int* simd_of) {
    __shared__ int ticket[4];
    __shared__ __attribute__((aligned(16))) float lds[1024 * 4 + 64];
    if (threadIdx.x < 4) ticket[threadIdx.x] = 0;
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = (hwid >> 4) & 3;
    int rank = 0;
    if ((threadIdx.x & 63) == 0) rank = atomicAdd(&ticket[simd], 1);
    rank = __builtin_amdgcn_readfirstlane(rank);
    const int role = roles[rank];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.f);
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float v[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.25f + i; w[i] = v[i] * 0.5f; }
    float c1 = 1.0001f + threadIdx.x * 1e-9f, c2 = 0.5f;
    f32x4 rd[4]; double dd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { rd[i] = (f32x4)(1.f + i); dd[i] = 1.0 + i; }
    const unsigned lds_off = (unsigned)(size_t)lds + (threadIdx.x & 1023) * 16;     // conflict-free b128 / b64 per lane
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(v[0]), "v"(c2) : "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mfma(acc[i], a, b);
#pragma unroll
                for (int j = 0; j < VPM; ++j) { const int q = (i * VPM + j) & 7; valu<OP>(v[q], w[q], c1, c2, lds_off, rd[q & 3], dd[q & 3]); }
            }
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mfma(acc[i], a, b);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < VPM; ++j) { const int q = (i * VPM + j) & 7; valu<OP>(v[q], w[q], c1, c2, lds_off, rd[q & 3], dd[q & 3]); }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += rd[i][0] + rd[i][3] + (float)dd[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[threadIdx.x >> 6] = t1 - t0; simd_of[threadIdx.x >> 6] = simd * 16 + rank; }
}

struct Res { double cyc_mfma = 0, cyc_valu = 0, cyc_both = 0, ns = 0; int n_mfma = 0, n_valu = 0, n_both = 0; };

template <int VPM, int OP>
static Res run(int waves_per_simd, const int roles_h[4], int blocks, float* out, unsigned long long* cyc, int* roles_d, int* simd_of) {
    const int iters = 2000;
    hipMemcpy(roles_d, roles_h, 16, hipMemcpyHostToDevice);
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<VPM, OP>), dim3(blocks), dim3(threads), 0, 0, out, 10, cyc, roles_d, simd_of);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VPM, OP>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc, roles_d, simd_of);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[16]; int so[16];
    hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    hipMemcpy(so, simd_of, sizeof(so), hipMemcpyDeviceToHost);
    Res r; r.ns = ms * 1e6 / iters;
    for (int wv = 0; wv < 4 * waves_per_simd; ++wv) {
        const int role = roles_h[so[wv] & 15];
        const double per = (double)c[wv] / iters;
        if (role == 1) { r.cyc_mfma += per; r.n_mfma++; } else if (role == 2) { r.cyc_valu += per; r.n_valu++; } else if (role == 3) { r.cyc_both += per; r.n_both++; }
    }
    if (r.n_mfma) r.cyc_mfma /= r.n_mfma;
    if (r.n_valu) r.cyc_valu /= r.n_valu;
    if (r.n_both) r.cyc_both /= r.n_both;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

static float* g_out; static unsigned long long* g_cyc; static int* g_roles; static int* g_simd;

template <int VPM, int OP>
static void same_wave(int blocks) {
    const int r3[4] = {3, 0, 0, 0};
    Res a = run<VPM, OP>(1, r3, blocks, g_out, g_cyc, g_roles, g_simd);
    printf("  same wave, %2d x %-16s per MFMA: %6.1f cycles per MFMA  (%6.1f ns per 4 MFMA)\n", VPM, OP_NAME[OP], a.cyc_both / 4, a.ns);
}

template <int VPM, int OP>
static void specialised(int blocks) {
    // stand-alone paces (the partner waves exit at once), then together; 2 waves per SIMD (A = MFMA, B = vector) and 3 (A, B, C = vector)
    const int ra[4] = {1, 0, 0, 0}, rb[4] = {0, 2, 0, 0}, rab[4] = {1, 2, 0, 0}, rbc[4] = {0, 2, 2, 0}, rabc[4] = {1, 2, 2, 0};
    Res a = run<VPM, OP>(2, ra, blocks, g_out, g_cyc, g_roles, g_simd);
    Res b = run<VPM, OP>(2, rb, blocks, g_out, g_cyc, g_roles, g_simd);
    Res ab = run<VPM, OP>(2, rab, blocks, g_out, g_cyc, g_roles, g_simd);
    Res bc = run<VPM, OP>(3, rbc, blocks, g_out, g_cyc, g_roles, g_simd);
    Res abc = run<VPM, OP>(3, rabc, blocks, g_out, g_cyc, g_roles, g_simd);
    printf("  specialised, %2d x %-16s per 4-MFMA iteration: alone A %6.1f  B %6.1f | A+B: A %6.1f  B %6.1f | B+C alone %6.1f | A+B+C: A %6.1f  B,C %6.1f   cycles per iteration;"
           " wall ns/iter A %.1f B %.1f A+B %.1f A+B+C %.1f\n",
           VPM * 4, OP_NAME[OP], a.cyc_mfma, b.cyc_valu, ab.cyc_mfma, ab.cyc_valu, bc.cyc_valu, abc.cyc_mfma, abc.cyc_valu, a.ns, b.ns, ab.ns, abc.ns);
}

template <int OP>
static void sweep(int blocks) {
    printf("== %s, %d workgroup(s) ==\n", OP_NAME[OP], blocks);
    same_wave<0, OP>(blocks); same_wave<1, OP>(blocks); same_wave<2, OP>(blocks); same_wave<3, OP>(blocks); same_wave<4, OP>(blocks);
    same_wave<5, OP>(blocks); same_wave<6, OP>(blocks); same_wave<8, OP>(blocks); same_wave<12, OP>(blocks); same_wave<16, OP>(blocks);
    specialised<4, OP>(blocks); specialised<8, OP>(blocks); specialised<16, OP>(blocks);
}

int main(int argc, char** argv) {
    hipMalloc(&g_out, sizeof(float) * 1024 * 1024); hipMalloc(&g_cyc, 8 * 16); hipMalloc(&g_roles, 16); hipMalloc(&g_simd, 4 * 16);
    for (int blocks : {1, 256}) {
        sweep<OP_FMA>(blocks); sweep<OP_IADD>(blocks); sweep<OP_CNDMASK>(blocks); sweep<OP_FMAMIX>(blocks); sweep<OP_MAX3>(blocks);
        sweep<OP_MOV>(blocks); sweep<OP_PKFMA>(blocks); sweep<OP_DSWRITE>(blocks); sweep<OP_DSREAD>(blocks);
    }
    return 0;
}
