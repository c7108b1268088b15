// How fast can a CU bring L2-resident data into LDS?  gfx950.   hipcc --offload-arch=gfx950 -O3 lds_fill_rate.hip -o lds_fill_rate
// Every workgroup (256 threads) streams `iters` tiles of 16 KB from a small per-workgroup region (L2 / MALL resident after the first pass)
//   mode 0: buffer_load_dwordx4 ... lds   (LDS-DMA, 1 KB per wave-instruction)
//   mode 1: global_load_dwordx4 -> VGPR, discarded (XOR-folded)                 = what the vector memory path returns to registers
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128                        = register staging
// pattern 0: lanes read 1 KB contiguous per instruction; pattern 1: 4 lanes x 16 B per 64-byte piece, pieces 768 B apart (one pixel's
// 16-channel chunk of a [H][W][192] image).  Reports GB/s per CU (event-timed, workgroups per CU = 1, 2, 3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PAT>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, unsigned* __restrict__ out, int iters, int region_units) {
    __shared__ __attribute__((aligned(16))) u32x4 lds[2][1024];          // 2 x 16 KB
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32x4* base = src + (size_t)blockIdx.x * region_units;
    const unsigned long long a = (unsigned long long)base;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a)),
        0, region_units * 16, 0x00020000);
    u32x4 accv = {0u, 0u, 0u, 0u};
    // item i (0..1023) of a tile: unit index inside the region
    unsigned off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = (wv * 4 + j) * 64 + lane;
        off[j] = PAT == 0 ? (unsigned)(i * 16) : (unsigned)(((i >> 2) * 48 + (i & 3)) * 16);      // pattern 1: pixel stride 768 B = 48 units
    }
    const int tile_bytes = PAT == 0 ? 16384 : 256 * 768;                  // bytes of the region one tile spans
    const int ntiles = region_units * 16 / tile_bytes;
    for (int it = 0; it < iters; ++it) {
        const int so = (it % ntiles) * tile_bytes;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(&lds[it & 1][(wv * 4 + j) * 64]), 16, (int)off[j], so, 0, 0);
            if ((it & 3) == 3) __syncthreads();                          // drains the DMA (vmcnt(0)) every 4 tiles: 16 instructions per wave in flight
        } else {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[j], so, 0);
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) accv ^= v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) lds[it & 1][(wv * 4 + j) * 64 + lane] = v[j];
                if ((it & 3) == 3) __syncthreads();
            }
        }
    }
    __syncthreads();
    const u32x4 l = lds[0][tid];
    out[blockIdx.x * 256 + tid] = accv.x ^ accv.y ^ accv.z ^ accv.w ^ l.x ^ l.w;
}

template <int MODE, int PAT>
static void run(const u32x4* src, unsigned* out, int wg_per_cu, int region_units) {
    const int iters = 2000, blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(256), 0, 0, src, out, 50, region_units);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(256), 0, 0, src, out, iters, region_units);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 16384.0;
    static const char* mn[3] = {"buffer_load ... lds (DMA)", "global_load -> VGPR      ", "global_load -> ds_write  "};
    printf("%s  %s  %d WG/CU: %7.1f GB/s per CU, %6.2f TB/s chip\n", mn[MODE], PAT ? "64-B pieces 768 B apart" : "1 KB contiguous        ", wg_per_cu,
           bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int region_units = 4 * 256 * 48;                               // 786 KB per workgroup: 4 tiles of pattern 1 (L2 / MALL resident: 768 WG x 786 KB = 600 MB > L2, so also a small-region run)
    u32x4* src; unsigned* out;
    hipMalloc(&src, (size_t)768 * region_units * 16); hipMemset(src, 1, (size_t)768 * region_units * 16);
    hipMalloc(&out, 768 * 256 * 4);
    for (int w = 1; w <= 3; ++w) {
        run<0, 0>(src, out, w, region_units); run<0, 1>(src, out, w, region_units);
        run<1, 0>(src, out, w, region_units); run<1, 1>(src, out, w, region_units);
        run<2, 0>(src, out, w, region_units); run<2, 1>(src, out, w, region_units);
    }
    printf("-- small regions (32 KB per workgroup: L2-resident) --\n");
    for (int w = 1; w <= 3; ++w) {
        run<0, 0>(src, out, w, 2048); run<1, 0>(src, out, w, 2048); run<2, 0>(src, out, w, 2048);
    }
    return 0;
}
