// fp32 global atomic-add element rate on gfx950: how fast can a scatter go at all?  (grid_sample_3d_backward is one atomic per touched
// (voxel, channel): csrc/k4_staged.hip k_grid_sample_bwd.)   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// mode 0: every element once, lanes consecutive (perfectly coalesced, conflict free)
// mode 1: lanes consecutive, every element hit `rep` times by DIFFERENT workgroups far apart in time (grid-stride repeats)
// mode 2: lane i -> pseudo-random element (scattered lines)
// mode 3: plain read-modify-write (no atomic), consecutive
// mode 4: rep adjacent workgroups hit the same 256 elements at about the same time (hot lines)
__global__ void k(float* __restrict__ a, int64_t n, int mode, int rep) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) { if (t < n) unsafeAtomicAdd(a + t, 1.f); }
    else if (mode == 1) { for (int r = 0; r < rep; ++r) { const int64_t i = (t + (int64_t)r * (n / rep) * 0 + (int64_t)r * 0) % n; if (t < n) unsafeAtomicAdd(a + i, 1.f); } }
    else if (mode == 2) { if (t < n) { uint64_t h = (uint64_t)t * 0x9E3779B97F4A7C15ull; h ^= h >> 29; unsafeAtomicAdd(a + (h % (uint64_t)n), 1.f); } }
    else if (mode == 3) { if (t < n) a[t] += 1.f; }
    else { if (t < n) unsafeAtomicAdd(a + (t / (256 * (int64_t)rep)) * 256 + (t & 255), 1.f); }
}

int main() {
    const int64_t n = 1ll << 27;                       // 128 Mi floats = 512 MiB
    float* a; CK(hipMalloc(&a, n * 4)); CK(hipMemset(a, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { int mode, rep; const char* name; } runs[] = {{0, 1, "coalesced, each element once"}, {1, 4, "coalesced, 4 atomics per thread to its element"},
        {2, 1, "scattered (hashed) elements"}, {3, 1, "plain RMW, no atomic"}, {4, 4, "4 workgroups share 256 elements"}, {4, 16, "16 workgroups share 256 elements"}};
    for (auto& r : runs) {
        for (int it = 0; it < 3; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, n, r.mode, r.rep);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = (double)n * (r.mode == 1 ? r.rep : 1);
            if (it == 2) printf("%-48s %8.3f ms  %7.1f G elem/s  %7.1f GB/s (8 B/elem)\n", r.name, ms, ops / ms / 1e6, ops * 8 / ms / 1e6);
        }
    }
    return 0;
}
