// Where does the time of the coherent grid_sample_3d backward scatter go?  8192 NDC-like rays x 256 samples, 12 channels, 417x353x256 grid
// (bench.py training_step_kernels' coherent case).  Variants of the one-thread-per-sample atomic scatter.
//   hipcc --offload-arch=gfx950 -O3 gsb_variants.hip -o gsb_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int X = 417, Y = 353, Z = 256, C = 12;

struct Tri { int x0, y0, z0; float w[8]; };
__device__ inline Tri setup(const float* p) {
    const float fx = (p[0] + 1.f) * 0.5f * (X - 1), fy = (p[1] + 1.f) * 0.5f * (Y - 1), fz = (p[2] + 1.f) * 0.5f * (Z - 1);
    Tri t; t.x0 = (int)floorf(fx); t.y0 = (int)floorf(fy); t.z0 = (int)floorf(fz);
    const float ax = fx - t.x0, ay = fy - t.y0, az = fz - t.z0;
#pragma unroll
    for (int c = 0; c < 8; ++c) t.w[c] = ((c & 4) ? ax : 1.f - ax) * ((c & 2) ? ay : 1.f - ay) * ((c & 1) ? az : 1.f - az);
    return t;
}
__device__ inline long long corner(const Tri& t, int c) {
    const int x = t.x0 + ((c >> 2) & 1), y = t.y0 + ((c >> 1) & 1), z = t.z0 + (c & 1);
    const bool ok = (unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z;
    return ok ? (long long)(((size_t)x * Y + y) * Z + z) : -1;
}
// variant 0: channel-major planes, lane = sample, 96 atomics        (the shipped kernel without the z-merge)
// variant 1: same, but every channel goes to plane 0                (is the 12-plane spread the cost?)
// variant 2: same as 0 with plain stores instead of atomics         (is it the atomic unit?)
// variant 3: channel-LAST target [X][Y][Z][12], lane = sample       (12 consecutive floats per corner, one plane)
// variant 4: no scatter at all: loads + arithmetic only, result folded into one store per thread
__global__ void k(const float* __restrict__ gout, const float* __restrict__ xyz, int64_t n, float* __restrict__ gg, int variant) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Tri t = setup(xyz + i * 3);
    long long a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = corner(t, c);
    const size_t plane = (size_t)X * Y * Z;
    float sink = 0.f;
    for (int ch = 0; ch < C; ++ch) {
        const float g = gout[i * C + ch];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (a[c] < 0) continue;
            const float v = g * t.w[c];
            if (variant == 0) unsafeAtomicAdd(gg + plane * ch + a[c], v);
            else if (variant == 1) unsafeAtomicAdd(gg + a[c], v);
            else if (variant == 2) gg[plane * ch + a[c]] = v;
            else if (variant == 3) unsafeAtomicAdd(gg + a[c] * C + ch, v);
            else sink += v;
        }
    }
    if (variant == 4) gg[i] = sink;
}
// variant 5: channel-last, lane = (sample, channel): 5 samples x 12 channels per wave-row of 60 lanes
__global__ void k5(const float* __restrict__ gout, const float* __restrict__ xyz, int64_t n, float* __restrict__ gg) {
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t0 / 16;                  // 16 lanes per sample, 12 active
    const int ch = (int)(t0 & 15);
    if (i >= n || ch >= C) return;
    const Tri t = setup(xyz + i * 3);
    const float g = gout[i * C + ch];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const long long a = corner(t, c);
        if (a >= 0) unsafeAtomicAdd(gg + a * C + ch, g * t.w[c]);
    }
}

// variant 6: the shipped kernel's z-merge (my z0+1 corner == next lane's z0 corner -> one atomic), channel-major
__global__ void k6(const float* __restrict__ gout, const float* __restrict__ xyz, int64_t n, float* __restrict__ gg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int lane = threadIdx.x & 63;
    const int64_t ic = valid ? i : 0;
    const Tri t = setup(xyz + ic * 3);
    long long a[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = valid ? corner(t, c) : -1;
    bool take_next[4], skip_lo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long nxt_lo = __shfl_down(a[2 * r], 1);
        take_next[r] = lane < 63 && a[2 * r + 1] >= 0 && nxt_lo == a[2 * r + 1];
        const int prev_took = __shfl_up((int)take_next[r], 1);
        skip_lo[r] = lane > 0 && prev_took != 0;
    }
    const size_t plane = (size_t)X * Y * Z;
    for (int ch = 0; ch < C; ++ch) {
        const float g = valid ? gout[ic * C + ch] : 0.f;
        float* const gp = gg + plane * ch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vlo = g * t.w[2 * r], vhi = g * t.w[2 * r + 1];
            const float nxt = __shfl_down(vlo, 1);
            if (a[2 * r + 1] >= 0) unsafeAtomicAdd(gp + a[2 * r + 1], take_next[r] ? vhi + nxt : vhi);
            if (a[2 * r] >= 0 && !skip_lo[r]) unsafeAtomicAdd(gp + a[2 * r], vlo);
        }
    }
}
// variant 7: channel-last, lane = (sample, channel) with the z-merge across the 16-lane groups
__global__ void k7(const float* __restrict__ gout, const float* __restrict__ xyz, int64_t n, float* __restrict__ gg) {
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t0 / 16;
    const int ch = (int)(t0 & 15);
    const int lane = threadIdx.x & 63;
    const bool valid = i < n && ch < C;
    const int64_t ic = i < n ? i : 0;
    const Tri t = setup(xyz + ic * 3);
    const float g = valid ? gout[ic * C + ch] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long lo = valid ? corner(t, 2 * r) : -1, hi = valid ? corner(t, 2 * r + 1) : -1;
        const long long nxt_lo = __shfl_down(lo, 16);
        const bool take = lane < 48 && hi >= 0 && nxt_lo == hi;
        const int prev_took = __shfl_up((int)take, 16);
        const bool skip = lane >= 16 && prev_took != 0;
        const float vlo = g * t.w[2 * r], vhi = g * t.w[2 * r + 1];
        const float nxt = __shfl_down(vlo, 16);
        if (hi >= 0) unsafeAtomicAdd(gg + hi * C + ch, take ? vhi + nxt : vhi);
        if (lo >= 0 && !skip) unsafeAtomicAdd(gg + lo * C + ch, vlo);
    }
}

int main() {
    const int64_t R = 8192, S = 256, n = R * S;
    std::vector<float> pts(n * 3), go(n * C);
    for (int64_t r = 0; r < R; ++r) {
        const int px = (int)(r % 1008), py = (int)(r / 1008);
        const float ox = -0.9f + 1.8f * px / 1007.f, oy = -0.9f + 1.8f * py / 755.f;
        const float dx = 0.1f * ox, dy = 0.1f * oy;                        // slow lateral drift along the ray
        for (int64_t s = 0; s < S; ++s) {
            const float tt = (float)s / (S - 1);
            float* p = &pts[(r * S + s) * 3];
            p[0] = ox + dx * tt; p[1] = oy + dy * tt; p[2] = -1.f + 2.f * tt;
        }
    }
    for (auto& v : go) v = 0.001f * (float)(rand() % 1000);
    float *dp, *dg, *gg;
    const size_t gbytes = (size_t)X * Y * Z * C * 4;
    CK(hipMalloc(&dp, pts.size() * 4)); CK(hipMalloc(&dg, go.size() * 4)); CK(hipMalloc(&gg, gbytes));
    CK(hipMemcpy(dp, pts.data(), pts.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, go.data(), go.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"channel-major planes, 96 atomics/sample", "all channels into plane 0", "plain stores instead of atomics", "channel-last, lane = sample",
                           "no scatter (loads + math)", "channel-last, lane = (sample, channel)", "channel-major + z-merge (shipped)", "channel-last (sample, channel) + z-merge"};
    for (int v = 0; v < 8; ++v) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            CK(hipMemset(gg, 0, gbytes));
            CK(hipEventRecord(e0));
            if (v < 5) hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dg, dp, n, gg, v);
            else if (v == 5) hipLaunchKernelGGL(k5, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, dg, dp, n, gg);
            else if (v == 6) hipLaunchKernelGGL(k6, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dg, dp, n, gg);
            else hipLaunchKernelGGL(k7, dim3((unsigned)((n * 16 + 255) / 256)), dim3(256), 0, 0, dg, dp, n, gg);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        printf("v%d %-44s %8.3f ms  %7.1f GB/s (828 B/sample)\n", v, names[v], best, n * 828.0 / best / 1e6);
    }
    return 0;
}
