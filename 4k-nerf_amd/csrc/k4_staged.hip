// Staged (one op per launch) gfx950 kernels: one per native function of the reference's
// `render_utils_cuda` (lib/cuda/render_utils.cpp:170-184) plus the two library ops the reference
// strings between them (F.grid_sample, torch_scatter.segment_coo).  They exist for callers that need
// per-sample tensors (training-compatible forward, the `render_utils_cuda` shim); the render path uses
// the fused kernels of k4_march.hip, which share the arithmetic written here op for op.
// All are HBM-streaming, one thread per element (or one WAVE per ray for the transmittance scan).
#include "k4_common.h"

#define K4_THREADS 256
static inline unsigned k4_blocks(int64_t n) { return (unsigned)((n + K4_THREADS - 1) / K4_THREADS); }

// ---------------------------------------------------------------- per-sample arithmetic shared by the staged kernels below and by
// k_train_select_mpi (one expression tree each: a decision taken there is the decision the staged op sequence takes)
__device__ __forceinline__ void k4s_ndc_point(const float* __restrict__ o, const float* __restrict__ d, int64_t ray, int step, int n_samples, float (&p)[3]) {
    const float dist = (float)step / (float)(n_samples - 1);                         // .cu:260
    p[0] = fmaf(d[ray * 3 + 0], dist, o[ray * 3 + 0]);
    p[1] = fmaf(d[ray * 3 + 1], dist, o[ray * 3 + 1]);
    p[2] = fmaf(d[ray * 3 + 2], dist, o[ray * 3 + 2]);
}
__device__ __forceinline__ bool k4s_outbbox(const float (&p)[3], const float* __restrict__ mn, const float* __restrict__ mx) {
    return (mn[0] > p[0]) | (mn[1] > p[1]) | (mn[2] > p[2]) | (mx[0] < p[0]) | (mx[1] < p[1]) | (mx[2] < p[2]);
}
__device__ __forceinline__ uint8_t k4s_maskcache(const uint8_t* __restrict__ world, float x, float y, float z, const float* __restrict__ sc,
                                                 const float* __restrict__ sh, int si, int sj, int sk) {
    const int a = k4_round_half_away(fmaf(x, sc[0], sh[0]));
    const int b = k4_round_half_away(fmaf(y, sc[1], sh[1]));
    const int c = k4_round_half_away(fmaf(z, sc[2], sh[2]));
    uint8_t v = 0;
    if ((unsigned)a < (unsigned)si && (unsigned)b < (unsigned)sj && (unsigned)c < (unsigned)sk)
        v = world[((size_t)a * sj + b) * sk + c] != 0;
    return v;
}
// corner indices / weights of F.grid_sample(bilinear, align_corners=True, zero padding) on an [X][Y][Z] grid
__device__ __forceinline__ void k4s_grid_corners(float x, float y, float z, const float* __restrict__ mn, const float* __restrict__ mx,
                                                 int X, int Y, int Z, size_t (&idx)[8], float (&w)[8]) {
    const float nx = k4_norm_coord(x, mn[0], mx[0]);
    const float ny = k4_norm_coord(y, mn[1], mx[1]);
    const float nz = k4_norm_coord(z, mn[2], mx[2]);
    const K4Tri t = k4_tri_setup(k4_unnorm(nx, X), k4_unnorm(ny, Y), k4_unnorm(nz, Z));
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int xx = t.x0 + K4_CX(c), yy = t.y0 + K4_CY(c), zz = t.z0 + K4_CZ(c);
        const bool ok = (unsigned)xx < (unsigned)X && (unsigned)yy < (unsigned)Y && (unsigned)zz < (unsigned)Z;
        idx[c] = ok ? ((size_t)xx * Y + yy) * Z + zz : 0;
        w[c] = ok ? t.w[c] : 0.f;
    }
}
__device__ __forceinline__ float k4s_grid_blend(const float* __restrict__ g, const size_t (&idx)[8], const float (&w)[8]) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) v += g[idx[c]] * w[c];
    return v;
}
__device__ __forceinline__ void k4s_raw2alpha(float den, float shift, float iv, float& e, float& al) {
    e = expf(den + shift);
    al = (iv == 1.f) ? 1.f - 1.f / (1.f + e) : 1.f - powf(1.f + e, -iv);
}

// ---------------------------------------------------------------- sample_ndc_pts_on_rays (.cu:245-270)
__global__ void k_sample_ndc(const float* __restrict__ o, const float* __restrict__ d,
                             const float* __restrict__ mn, const float* __restrict__ mx,
                             int64_t n_rays, int n_samples, float* __restrict__ pts, uint8_t* __restrict__ mask) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rays * n_samples) return;
    const int64_t ray = idx / n_samples;
    const int step = (int)(idx % n_samples);
    float p[3];
    k4s_ndc_point(o, d, ray, step, n_samples, p);
    pts[idx * 3 + 0] = p[0]; pts[idx * 3 + 1] = p[1]; pts[idx * 3 + 2] = p[2];
    mask[idx] = k4s_outbbox(p, mn, mx);
}

// ---------------------------------------------------------------- ray-AABB helpers (.cu:12-79)
__device__ __forceinline__ void aabb_t(const float* o, const float* d, const float* mn, const float* mx,
                                        float near, float far, int64_t r, float& t_min, float& t_max) {
    const float vx = d[r * 3 + 0] == 0.f ? 1e-6f : d[r * 3 + 0];
    const float vy = d[r * 3 + 1] == 0.f ? 1e-6f : d[r * 3 + 1];
    const float vz = d[r * 3 + 2] == 0.f ? 1e-6f : d[r * 3 + 2];
    const float ax = (mx[0] - o[r * 3 + 0]) / vx, ay = (mx[1] - o[r * 3 + 1]) / vy, az = (mx[2] - o[r * 3 + 2]) / vz;
    const float bx = (mn[0] - o[r * 3 + 0]) / vx, by = (mn[1] - o[r * 3 + 1]) / vy, bz = (mn[2] - o[r * 3 + 2]) / vz;
    t_min = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), far), near);
    t_max = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), far), near);
}
__device__ __forceinline__ float ray_norm(const float* d, int64_t r) {
    return sqrtf(fmaf(d[r * 3 + 2], d[r * 3 + 2], fmaf(d[r * 3 + 1], d[r * 3 + 1], d[r * 3 + 0] * d[r * 3 + 0])));
}
__device__ __forceinline__ int64_t n_steps_of(float t_min, float t_max, float rnorm, float stepdist) {
    return (int64_t)fmaxf(ceilf((t_max - t_min) * rnorm / stepdist), 1.f);      // at least 1 point (.cu:53)
}

__global__ void k_t_minmax(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ mn,
                           const float* __restrict__ mx, float near, float far, int64_t n,
                           float* __restrict__ tmin, float* __restrict__ tmax) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float a, b;
    aabb_t(o, d, mn, mx, near, far, r, a, b);
    tmin[r] = a; tmax[r] = b;
}
__global__ void k_n_samples(const float* __restrict__ d, const float* __restrict__ tmin, const float* __restrict__ tmax,
                            float stepdist, int64_t n, int64_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    out[r] = n_steps_of(tmin[r], tmax[r], ray_norm(d, r), stepdist);
}
__global__ void k_start_dir(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ tmin,
                            int64_t n, float* __restrict__ start, float* __restrict__ dir) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float rn = ray_norm(d, r);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        start[r * 3 + c] = fmaf(d[r * 3 + c], tmin[r], o[r * 3 + c]);
        dir[r * 3 + c] = d[r * 3 + c] / rn;
    }
}
__global__ void k_pts_count(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ mn,
                            const float* __restrict__ mx, float near, float far, float stepdist, int64_t n,
                            int64_t* __restrict__ nsteps, float* __restrict__ tmin, float* __restrict__ tmax) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    float a, b;
    aabb_t(o, d, mn, mx, near, far, r, a, b);
    tmin[r] = a; tmax[r] = b;
    nsteps[r] = n_steps_of(a, b, ray_norm(d, r), stepdist);
}
// one thread per output point; its ray is found by binary search in the inclusive cumsum (replaces the
// reference's scatter-1 + second cumsum, .cu:144-164,213-219)
__global__ void k_pts_fill(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ mn,
                           const float* __restrict__ mx, const float* __restrict__ tmin,
                           const int64_t* __restrict__ cum, float stepdist, int64_t n_rays, int64_t total,
                           float* __restrict__ pts, uint8_t* __restrict__ mask, int64_t* __restrict__ ray_id,
                           int64_t* __restrict__ step_id) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t lo = 0, hi = n_rays - 1;                   // first ray with cum[ray] > idx
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cum[mid] > idx) hi = mid; else lo = mid + 1;
    }
    const int64_t r = lo;
    const int64_t step = idx - (r > 0 ? cum[r - 1] : 0);
    const float rn = ray_norm(d, r);
    const float dist = stepdist * (float)step;
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float start = fmaf(d[r * 3 + c], tmin[r], o[r * 3 + c]);
        const float dir = d[r * 3 + c] / rn;
        p[c] = fmaf(dir, dist, start);
        pts[idx * 3 + c] = p[c];
    }
    mask[idx] = (mn[0] > p[0]) | (mn[1] > p[1]) | (mn[2] > p[2]) | (mx[0] < p[0]) | (mx[1] < p[1]) | (mx[2] < p[2]);
    ray_id[idx] = r; step_id[idx] = step;
}

// ---------------------------------------------------------------- maskcache_lookup (.cu:374-392)
__global__ void k_maskcache(const uint8_t* __restrict__ world, const float* __restrict__ xyz,
                            const float* __restrict__ sc, const float* __restrict__ sh,
                            int si, int sj, int sk, int64_t n, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = k4s_maskcache(world, xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], sc, sh, si, sj, sk);
}

// ---------------------------------------------------------------- raw2alpha (+bwd) (.cu:431-458, 507-530)
__global__ void k_raw2alpha(const float* __restrict__ den, float shift, float interval, const float* __restrict__ ipp,
                            int64_t n, float* __restrict__ ex, float* __restrict__ al) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float iv = ipp ? ipp[i] : interval;
    float e, a;
    k4s_raw2alpha(den[i], shift, iv, e, a);
    ex[i] = e;
    al[i] = a;
}
__global__ void k_raw2alpha_bwd(const float* __restrict__ ex, const float* __restrict__ gb, float interval,
                                const float* __restrict__ ipp, int64_t n, float* __restrict__ g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float iv = ipp ? ipp[i] : interval;
    g[i] = fminf(ex[i], 1e10f) * powf(1.f + ex[i], -iv - 1.f) * iv * gb[i];
}

// ---------------------------------------------------------------- alpha2weight (.cu:577-651)
// segment bounds from change points of the sorted ray_id (incl. the reference's host-side
// `i_end[ray_id[n-1]] = n`, done on the device here: no sync)
__global__ void k_seg_bounds(const int64_t* __restrict__ ray_id, int64_t n, int64_t* __restrict__ i_start,
                             int64_t* __restrict__ i_end) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i > 0 && ray_id[i] != ray_id[i - 1]) { i_start[ray_id[i]] = i; i_end[ray_id[i - 1]] = i; }
    if (i == n - 1) i_end[ray_id[i]] = n;
}
// one WAVE per ray; lanes take 64 consecutive points; the product is the reference's exact sequential
// one (ballot + readlane), including the sample that crosses T<1e-3 (.cu:597-600)
__global__ __launch_bounds__(256) void k_alpha2weight(const float* __restrict__ alpha, int64_t n_rays,
                                                      float* __restrict__ weight, float* __restrict__ Tout,
                                                      float* __restrict__ ainv, const int64_t* __restrict__ i_start,
                                                      int64_t* __restrict__ i_end) {
    const int lane = k4_lane();
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int64_t s = i_start[ray], e = i_end[ray];
    float T = 1.f;
    int64_t stop_at = e;
    bool stopped = false;
    for (int64_t base = s; base < e && !stopped; base += 64) {
        const int64_t i = base + lane;
        const bool v = i < e;
        const float a = v ? alpha[i] : 0.f;
        float myT = 1.f, myw = 0.f;
        uint64_t bm = __ballot(v);
        while (bm) {
            const int l = __builtin_ctzll(bm);
            const float al = k4_readlane(a, l);
            if (lane == l) { myT = T; myw = T * al; }
            T = fmaf(-T, al, T);                      // == round(T*(1-al)), see k4_march.hip
            bm &= bm - 1;
            if (T < 1e-3f) { stopped = true; stop_at = base + l + 1; break; }
        }
        if (v && i < stop_at) { Tout[i] = myT; weight[i] = myw; }
    }
    if (lane == 0) { i_end[ray] = stop_at; ainv[ray] = T; }
}
__global__ void k_fill2(float* __restrict__ a, float va, float* __restrict__ b, float vb, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = va; if (b) b[i] = vb; }
}
__global__ void k_fill_i64(int64_t* __restrict__ a, int64_t* __restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; b[i] = 0; }
}
// alpha2weight_backward (.cu:654-677): grad[i] = gw[i]*T[i] - back_i / (1 - alpha[i] + 1e-10), back_i = gl*ainv + sum_{j>i} gw[j]*w[j].
// One WAVE per ray (the reference walks a ray's samples with one thread: uncoalesced, 8192 threads per batch): the segment is
// taken in chunks of 64 from its far end, lane = sample, the suffix sums are a wave scan plus the carry of the chunks behind.
__global__ void k_alpha2weight_bwd(const float* __restrict__ alpha, const float* __restrict__ weight,
                                   const float* __restrict__ T, const float* __restrict__ ainv,
                                   const int64_t* __restrict__ i_start, const int64_t* __restrict__ i_end,
                                   int64_t n_rays, const float* __restrict__ gw, const float* __restrict__ gl,
                                   float* __restrict__ grad) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n_rays) return;
    const int lane = k4_lane();
    const int64_t s0 = i_start[r], s1 = i_end[r];
    float carry = gl[r] * ainv[r];                                   // back behind the chunk being processed (wave-uniform)
    for (int64_t hi = s1; hi > s0; hi -= 64) {
        const int64_t i = hi - 1 - lane;                             // lane 0 = farthest sample of the chunk
        const bool ok = i >= s0;
        const float p = ok ? gw[i] * weight[i] : 0.f;
        float incl = p;                                              // inclusive sum over lanes 0..lane = samples i .. hi-1
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (ok) grad[i] = gw[i] * T[i] - (carry + (incl - p)) / (1.f - alpha[i] + 1e-10f);
        carry += __shfl(incl, 63);
    }
}

// ---------------------------------------------------------------- DenseGrid.forward (lib/grid.py:117-128)
__global__ void k_grid_sample(const float* __restrict__ grid, int C, int X, int Y, int Z,
                              const float* __restrict__ xyz, const float* __restrict__ mn, const float* __restrict__ mx,
                              int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t idx[8]; float w[8];
    k4s_grid_corners(xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], mn, mx, X, Y, Z, idx, w);
    const size_t plane = (size_t)X * Y * Z;
    for (int ch = 0; ch < C; ++ch) out[i * C + ch] = k4s_grid_blend(grid + plane * ch, idx, w);
}

// ---------------------------------------------------------------- training forward without host round trips (round 5)
// The reference's training forward (lib/dmpigo.py:300-333) filters its sample list three times -- bounding box + mask cache, alpha >
// fast_color_thres, weight > fast_color_thres -- and each boolean-mask indexing is a device-to-host synchronisation (4 per iteration in
// the op-for-op mirror).  k_train_select_mpi takes ALL THREE decisions for a batch of rays in one launch, with the arithmetic of the staged
// kernels above (k4s_*: a decision here is the decision the op sequence takes): one wave per ray, lanes = 64 consecutive samples; the
// transmittance product is the exact sequential one of k_alpha2weight (incl. the T < 1e-3 stop: samples behind it keep weight 0).  Per ray it
// leaves the steps of the alpha-passing samples (ascending), a flag per such sample "weight passes too", and the two counts; after one
// cumsum per count (and ONE read-back of the two totals, to size the tensors) k_train_compact writes the global lists the differentiable
// ops then run on: ray_id / step_id of the alpha-passing samples and, for the shaded ones, their index into that list.
__global__ __launch_bounds__(256) void k_train_select_mpi(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ mn,
                                                          const float* __restrict__ mx, int64_t n_rays, int n_samples,
                                                          const uint8_t* __restrict__ world, const float* __restrict__ sc, const float* __restrict__ sh,
                                                          int si, int sj, int sk,
                                                          const float* __restrict__ dens, int X, int Y, int Z,
                                                          const float* __restrict__ act, int AD, float interval, float thres,
                                                          int16_t* __restrict__ steps2, uint8_t* __restrict__ keep3,
                                                          int64_t* __restrict__ cnt2, int64_t* __restrict__ cnt3) {
    const int lane = k4_lane();
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    float T = 1.f;
    bool stopped = false;
    int n2 = 0, n3 = 0;
    int16_t* const my_steps = steps2 + ray * n_samples;
    uint8_t* const my_keep = keep3 + ray * n_samples;
    for (int base = 0; base < n_samples; base += 64) {
        const int step = base + lane;
        const bool v = step < n_samples;
        float p[3];
        k4s_ndc_point(o, d, ray, v ? step : 0, n_samples, p);
        bool f2 = v && !k4s_outbbox(p, mn, mx);                                           // lib/dmpigo.py:300-303
        if (f2) f2 = k4s_maskcache(world, p[0], p[1], p[2], sc, sh, si, sj, sk) != 0;     // :308-313
        float alpha = 0.f;
        if (f2) {
            size_t idx[8]; float w[8];
            k4s_grid_corners(p[0], p[1], p[2], mn, mx, X, Y, Z, idx, w);
            const float den = k4s_grid_blend(dens, idx, w);                              // self.density(ray_pts)
            k4s_grid_corners(p[0], p[1], p[2], mn, mx, 1, 1, AD, idx, w);
            const float ash = k4s_grid_blend(act, idx, w);                               // self.act_shift(ray_pts)
            const float sum = den + ash;                                                 // :316
            float e;
            k4s_raw2alpha(sum, 0.f, interval, e, alpha);                                 // :317, shift 0 (lib/dmpigo.py:261)
            f2 = alpha > thres;                                                          // :319
        }
        uint64_t bm = __ballot(f2);
        const uint64_t bm2 = bm;
        float myw = 0.f;                                                                 // weights start as zeros (.cu:624)
        while (bm && !stopped) {                                                         // k_alpha2weight's sequential product
            const int l = __builtin_ctzll(bm);
            const float al = k4_readlane(alpha, l);
            if (lane == l) myw = T * al;
            T = fmaf(-T, al, T);
            bm &= bm - 1;
            if (T < 1e-3f) stopped = true;                                               // the crossing sample keeps its weight (:597-600)
        }
        const bool f3 = f2 && myw > thres;                                               // :329
        if (f2) {
            const int pos = n2 + k4_prefix(bm2);
            my_steps[pos] = (int16_t)step;
            my_keep[pos] = f3 ? 1 : 0;
        }
        n2 += __popcll(bm2);
        n3 += __popcll(__ballot(f3));
    }
    if (lane == 0) { cnt2[ray] = n2; cnt3[ray] = n3; }
}
// one wave per ray: its alpha-passing samples go to [off2[ray], off2[ray] + cnt2) of the global list, the shaded ones' list positions to idx3
__global__ __launch_bounds__(256) void k_train_compact(const int16_t* __restrict__ steps2, const uint8_t* __restrict__ keep3,
                                                       const int64_t* __restrict__ cnt2, const int64_t* __restrict__ inc2, const int64_t* __restrict__ inc3,
                                                       int64_t n_rays, int n_samples, int64_t* __restrict__ ray_id, int64_t* __restrict__ step_id,
                                                       int64_t* __restrict__ idx3) {
    const int lane = k4_lane();
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int n2 = (int)cnt2[ray];
    const int64_t o2 = inc2[ray] - n2;                                                   // exclusive offsets from the inclusive cumsums
    int64_t o3 = ray > 0 ? inc3[ray - 1] : 0;
    for (int base = 0; base < n2; base += 64) {
        const int j = base + lane;
        const bool v = j < n2;
        const int st = v ? (int)steps2[ray * n_samples + j] : 0;
        const bool k3 = v && keep3[ray * n_samples + j] != 0;
        if (v) { ray_id[o2 + j] = ray; step_id[o2 + j] = st; }
        const uint64_t bm = __ballot(k3);
        if (k3) idx3[o3 + k4_prefix(bm)] = o2 + j;
        o3 += __popcll(bm);
    }
}
// ray_pts of listed samples: the sampler's formula on (ray_id, step_id)
__global__ void k_ndc_points_of(const float* __restrict__ o, const float* __restrict__ d, const int64_t* __restrict__ ray_id,
                                const int64_t* __restrict__ step_id, int64_t n, int n_samples, float* __restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p[3];
    k4s_ndc_point(o, d, ray_id[i], (int)step_id[i], n_samples, p);
    pts[i * 3 + 0] = p[0]; pts[i * 3 + 1] = p[1]; pts[i * 3 + 2] = p[2];
}

// ---------------------------------------------------------------- segment_coo(sum), sorted index
// one thread per (segment head, channel): walks its run sequentially -> deterministic, same order as index_add_
__global__ void k_segment_sum(const float* __restrict__ src, const int64_t* __restrict__ index, int64_t n, int C,
                              float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    const int64_t i = t / C;
    const int ch = (int)(t % C);
    if (i > 0 && index[i - 1] == index[i]) return;
    const int64_t seg = index[i];
    float acc = 0.f;
    for (int64_t j = i; j < n && index[j] == seg; ++j) acc += src[j * C + ch];
    out[seg * C + ch] += acc;
}

// ---------------------------------------------------------------- d(DenseGrid.forward)/d(grid)  
// What autograd runs for lib/grid.py:124 in the reference: grid_sampler_3d_backward's grad_input, i.e. every sample adds
// grad_out[i][ch] * w[c] to its 8 corners (zero padding: corners outside the grid receive nothing).  One thread per
// sample: the corner indices / weights are recomputed (same setup as the forward), the scatter uses the hardware fp32
// atomic add (global_atomic_add_f32, no return) -- order-nondeterministic like the reference's fastAtomicAdd.  d/d(xyz) is never needed (sample points come from rays, lib/dvgo.py:350).
__global__ void k_grid_sample_bwd(const float* __restrict__ gout, int C, int X, int Y, int Z,
                                  const float* __restrict__ xyz, const float* __restrict__ mn, const float* __restrict__ mx,
                                  int64_t n, float* __restrict__ ggrid) {
    // lane = sample (consecutive samples of a ray sit in consecutive lanes).  A sample's corners are 4 (x,y) rows x
    // {z0, z0+1}; the z0+1 corner of one sample is very often the z0 corner of the next sample of the same ray (same row,
    // z0 one higher).  Those pairs are merged with one shuffle so that the voxel receives ONE atomic instead of two: the
    // scatter is bound by atomic throughput (96 per sample at 12 channels), the merge removes up to half of them.
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int lane = k4_lane();
    const int64_t ic = valid ? i : 0;
    const float nx = k4_norm_coord(xyz[ic * 3 + 0], mn[0], mx[0]);
    const float ny = k4_norm_coord(xyz[ic * 3 + 1], mn[1], mx[1]);
    const float nz = k4_norm_coord(xyz[ic * 3 + 2], mn[2], mx[2]);
    const K4Tri tr = k4_tri_setup(k4_unnorm(nx, X), k4_unnorm(ny, Y), k4_unnorm(nz, Z));
    const size_t plane = (size_t)X * Y * Z;
    // rows: c>>1 = (dx,dy); corner c = row*2 + dz.  -1 marks a corner outside the grid (zero padding: no contribution)
    long long addr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int x = tr.x0 + K4_CX(c), y = tr.y0 + K4_CY(c), z = tr.z0 + K4_CZ(c);
        const bool ok = valid && (unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z;
        addr[c] = ok ? (long long)(((size_t)x * Y + y) * Z + z) : -1;
    }
    // merge pattern (channel independent): my z0+1 corner of row r == the next lane's z0 corner of row r
    bool take_next[4], skip_lo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long nxt_lo = __shfl_down(addr[2 * r], 1);
        take_next[r] = lane < 63 && addr[2 * r + 1] >= 0 && nxt_lo == addr[2 * r + 1];
        const int prev_took = __shfl_up((int)take_next[r], 1);
        skip_lo[r] = lane > 0 && prev_took != 0;
    }
    for (int ch = 0; ch < C; ++ch) {
        const float g = valid ? gout[ic * C + ch] : 0.f;
        float* const gp = ggrid + plane * ch;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vlo = g * tr.w[2 * r], vhi = g * tr.w[2 * r + 1];
            const float nxt = __shfl_down(vlo, 1);
            if (addr[2 * r + 1] >= 0) unsafeAtomicAdd(gp + addr[2 * r + 1], take_next[r] ? vhi + nxt : vhi);
            if (addr[2 * r] >= 0 && !skip_lo[r]) unsafeAtomicAdd(gp + addr[2 * r], vlo);
        }
    }
}

// The same scatter through a channel-LAST scratch image (tools/micro/gsb_variants.hip, tools/micro/atomic_rate.hip): the fp32 atomic
// units retire ~330 G elements/s when the lanes of an instruction hit consecutive addresses and ~20 G/s when every lane hits its own
// cache line.  With the parameter's channel-major layout [C][X][Y][Z] a sample's 8 x C contributions are 8 x C different lines
// unless its wave-mates walk along z.  Here lane = (sample, channel): a corner's C contributions are C consecutive floats of
// scratch[voxel][C], so one atomic instruction covers 64 / LPS samples x C channels in as many line segments as there are distinct
// voxels -- C x fewer line operations for incoherent batches, ~2x fewer instructions' worth of time for coherent ones (measured 1.47
// -> 0.65 ms on 8192 rays x 256 samples x 12 channels).  The z-merge of the channel-major kernel carries over (next sample = lane + LPS).
// One byte per touched voxel is raised in `flags`; k_gsb_cl_sweep then moves the touched voxels' sums into the channel-major gradient
// (plain read-modify-write, every voxel has one owner) and clears scratch and flags behind itself: the workspace is all-zero on
// entry AND on exit, it is never memset per call.
template <int LPS>
__global__ __launch_bounds__(256) void k_gsb_cl_scatter(const float* __restrict__ gout, int C, int X, int Y, int Z, const float* __restrict__ xyz,
                                                         const float* __restrict__ mn, const float* __restrict__ mx, int64_t n,
                                                         float* __restrict__ scratch, uint8_t* __restrict__ flags) {
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t0 / LPS;
    const int ch = (int)(t0 % LPS);
    const int lane = k4_lane();
    const bool valid = i < n;
    const bool chok = valid && ch < C;
    const int64_t ic = valid ? i : 0;
    const float nx = k4_norm_coord(xyz[ic * 3 + 0], mn[0], mx[0]);
    const float ny = k4_norm_coord(xyz[ic * 3 + 1], mn[1], mx[1]);
    const float nz = k4_norm_coord(xyz[ic * 3 + 2], mn[2], mx[2]);
    const K4Tri tr = k4_tri_setup(k4_unnorm(nx, X), k4_unnorm(ny, Y), k4_unnorm(nz, Z));
    const float g = chok ? gout[ic * C + ch] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                        // row r = (dx, dy); corners 2r (z0) and 2r + 1 (z0 + 1)
        long long a[2];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const int c = 2 * r + dz;
            const int x = tr.x0 + K4_CX(c), y = tr.y0 + K4_CY(c), z = tr.z0 + K4_CZ(c);
            const bool ok = valid && (unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z;
            a[dz] = ok ? (long long)(((size_t)x * Y + y) * Z + z) : -1;
        }
        const long long nxt_lo = __shfl_down(a[0], LPS);
        const bool take = lane < 64 - LPS && a[1] >= 0 && nxt_lo == a[1];       // my z0+1 corner is the next sample's z0 corner
        const int prev_took = __shfl_up((int)take, LPS);
        const bool skip = lane >= LPS && prev_took != 0;
        const float vlo = g * tr.w[2 * r], vhi = g * tr.w[2 * r + 1];
        const float nxt = __shfl_down(vlo, LPS);
        if (a[1] >= 0) {
            if (chok) unsafeAtomicAdd(scratch + a[1] * C + ch, take ? vhi + nxt : vhi);
            if (ch == 0) flags[a[1]] = 1;
        }
        if (a[0] >= 0 && !skip) {
            if (chok) unsafeAtomicAdd(scratch + a[0] * C + ch, vlo);
            if (ch == 0) flags[a[0]] = 1;
        }
    }
}

// flags[voxel] = 1 for every voxel a scatter of the points `xyz` will touch (the eight corners, k_gsb_cl_scatter's index arithmetic and range test): known
// as soon as the lookup's FORWARD has its points -- MaskedAdam steps every other voxel of a grid whose only other gradient term is already known (the
// dense TV term written ahead) while the rest of the iteration runs (k4_masked_adam_upd_unflagged / k4_masked_adam_upd_sparse_cl_seeded, k4_opt.hip).
__global__ __launch_bounds__(256) void k_gsb_flag_corners(int X, int Y, int Z, const float* __restrict__ xyz, const float* __restrict__ mn, const float* __restrict__ mx,
                                                           int64_t n, uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float nx = k4_norm_coord(xyz[i * 3 + 0], mn[0], mx[0]);
    const float ny = k4_norm_coord(xyz[i * 3 + 1], mn[1], mx[1]);
    const float nz = k4_norm_coord(xyz[i * 3 + 2], mn[2], mx[2]);
    const K4Tri tr = k4_tri_setup(k4_unnorm(nx, X), k4_unnorm(ny, Y), k4_unnorm(nz, Z));
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int x = tr.x0 + K4_CX(c), y = tr.y0 + K4_CY(c), z = tr.z0 + K4_CZ(c);
        if ((unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z) flags[((size_t)x * Y + y) * Z + z] = 1;
    }
}

__global__ __launch_bounds__(256) void k_gsb_cl_sweep(float* __restrict__ scratch, uint8_t* __restrict__ flags, int C, int64_t nvox,
                                                       float* __restrict__ ggrid) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nvox || flags[v] == 0) return;
    flags[v] = 0;
    float* const s = scratch + v * C;
    for (int ch = 0; ch < C; ++ch) {
        const float q = s[ch];
        s[ch] = 0.f;
        ggrid[(size_t)ch * nvox + v] += q;
    }
}

// ---------------------------------------------------------------- d(segment_coo sum)/d(src): grad_src[i] = grad_out[index[i]]
__global__ void k_segment_gather(const float* __restrict__ gout, const int64_t* __restrict__ index, int64_t n, int C,
                                 float* __restrict__ gsrc) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * C) return;
    const int64_t i = t / C;
    gsrc[t] = gout[index[i] * C + (t - i * C)];
}

// ---------------------------------------------------------------- get_rays_of_a_view (lib/dvgo.py:516-582), SURVEY.md 8f rank 4
// One thread per pixel, no intermediate [H,W,3] tensors: pixel -> camera direction -> world ray -> unit view direction
// (BEFORE the NDC warp) -> optional NDC warp (near = 1, focal = K[0][0]).  The arithmetic restates the reference's torch
// op sequence one rounding at a time (__fmul_rn/__fadd_rn/__fdiv_rn: no contraction), so the rays equal the torch-built
// ones to the last bit wherever torch itself is deterministic (its norm reduction may differ in the last bit).
__global__ void k_rays_of_view(int H, int W, const float* __restrict__ Kd, const float* __restrict__ c2w, int ndc,
                               int inverse_y, int flip_x, int flip_y, float pix_off, float c_w, float c_h,
                               float* __restrict__ ro, float* __restrict__ rd, float* __restrict__ vd) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int row = (int)(p / W), col = (int)(p - (int64_t)row * W);
    const k4_cptr K = k4_const(Kd), M = k4_const(c2w);          // uniform: scalar loads
    const float i = (float)(flip_x ? W - 1 - col : col) + pix_off;
    const float j = (float)(flip_y ? H - 1 - row : row) + pix_off;
    float d[3];
    d[0] = __fdiv_rn(__fsub_rn(i, K[2]), K[0]);
    d[1] = __fdiv_rn(__fsub_rn(j, K[5]), K[4]);
    d[2] = 1.f;
    if (!inverse_y) { d[1] = -d[1]; d[2] = -1.f; }
    float v[3], o[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {                                // torch.sum(dirs[..., None, :] * c2w[:3,:3], -1)
        v[a] = __fadd_rn(__fadd_rn(__fmul_rn(d[0], M[a * 4 + 0]), __fmul_rn(d[1], M[a * 4 + 1])), __fmul_rn(d[2], M[a * 4 + 2]));
        o[a] = M[a * 4 + 3];
    }
    const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
#pragma unroll
    for (int a = 0; a < 3; ++a) vd[p * 3 + a] = __fdiv_rn(v[a], nrm);
    if (ndc) {                                                   // ndc_rays, near = 1 (lib/dvgo.py:557-574)
        const float t = __fdiv_rn(-__fadd_rn(1.f, o[2]), v[2]);
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a] = __fadd_rn(o[a], __fmul_rn(t, v[a]));
        const float o0 = __fdiv_rn(__fmul_rn(c_w, o[0]), o[2]);
        const float o1 = __fdiv_rn(__fmul_rn(c_h, o[1]), o[2]);
        const float o2 = __fadd_rn(1.f, __fdiv_rn(2.f, o[2]));
        const float d0 = __fmul_rn(c_w, __fsub_rn(__fdiv_rn(v[0], v[2]), __fdiv_rn(o[0], o[2])));
        const float d1 = __fmul_rn(c_h, __fsub_rn(__fdiv_rn(v[1], v[2]), __fdiv_rn(o[1], o[2])));
        const float d2 = __fdiv_rn(-2.f, o[2]);
        o[0] = o0; o[1] = o1; o[2] = o2; v[0] = d0; v[1] = d1; v[2] = d2;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { ro[p * 3 + a] = o[a]; rd[p * 3 + a] = v[a]; }
}

// ---------------------------------------------------------------- utils.to8b (lib/utils.py:19): (255*clip(x,0,1)).astype(uint8)
__global__ void k_to8b(const float* __restrict__ x, int64_t n, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = fminf(fmaxf(x[i], 0.f), 1.f);               // np.clip: NaN propagates in numpy; maps to 0 here
    out[i] = (uint8_t)(int)__fmul_rn(255.f, c);                  // astype(uint8): truncation
}

// ---------------------------------------------------------------- a window of an NHWC image -> planes (the decoder's result into the frame)
// SFTNet.tile_process (lib/sr_esrnet.py:508-524) crops a tile's interior out of the padded window's result and writes it into the [1, 3, 4H, 4W] frame.  The
// decoder's result is NHWC: as PyTorch slice assignments that was a channel-stride gather (reads 12 bytes apart per output element, a temporary for the
// `reshape`, two or three passes over the 146 MB of a 4K frame): 1.8 ms of a 28.5 ms frame.  Here one pass: a thread reads 4 consecutive pixels (48
// contiguous bytes at C = 3) and writes 4 consecutive elements of each plane.
__global__ __launch_bounds__(256) void k_nhwc_window_to_planar(const float* __restrict__ src, int src_w, int C, int oy, int ox, int th, int tw,
                                                               float* __restrict__ dst, int64_t plane, int64_t row) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int tw4 = (tw + 3) >> 2;
    if (i >= (int64_t)th * tw4) return;
    const int y = (int)(i / tw4), x = (int)(i - (int64_t)y * tw4) * 4;
    const int n = tw - x < 4 ? tw - x : 4;
    const float* const s = src + ((int64_t)(oy + y) * src_w + ox + x) * C;
    float v[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[p][c] = (p < n && c < C) ? s[p * C + c] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= C) break;
        float* const d = dst + (int64_t)c * plane + (int64_t)y * row + x;
        if (n == 4 && (((uintptr_t)d) & 15u) == 0) *reinterpret_cast<float4*>(d) = make_float4(v[0][c], v[1][c], v[2][c], v[3][c]);
        else
            for (int p = 0; p < n; ++p) d[p] = v[p][c];
    }
}

// The 3-channel form (every decoded window): a workgroup moves W2P_PX pixels of one row.  The row segment is read as it lies in memory (consecutive
// lanes, consecutive dwords -- float4s when the segment starts on a 16-byte boundary: one visit per cache line, where the form above visits a line
// from twelve load instructions), transposed through LDS ([c][px], plane pitch chosen so that three consecutive dwords land on three banks) and written
// as float4s per plane.
#define W2P_PX 1024
#define W2P_PITCH (W2P_PX + 12)
__global__ __launch_bounds__(256) void k_nhwc3_window_to_planar(const float* __restrict__ src, int src_w, int oy, int ox, int tw, float* __restrict__ dst,
                                                                int64_t plane, int64_t row) {
    __shared__ float lds[3 * W2P_PITCH];
    const int y = blockIdx.y, x0 = blockIdx.x * W2P_PX;
    const int n = tw - x0 < W2P_PX ? tw - x0 : W2P_PX;             // pixels of this segment
    const int nf = n * 3;
    const float* const s = src + ((int64_t)(oy + y) * src_w + ox + x0) * 3;
    const int t = threadIdx.x;
    if ((((uintptr_t)s) & 15u) == 0) {
        const int nq = nf >> 2;
        for (int q = t; q < nq; q += 256) {
            const float4 v = reinterpret_cast<const float4*>(s)[q];
            const int i = q * 4, px = i / 3, c = i - px * 3;        // float i = pixel px, channel c
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cc = c + k, wrap = cc >= 3;               // (c + k <= 5: at most one pixel further)
                lds[(cc - 3 * wrap) * W2P_PITCH + px + wrap] = e[k];
            }
        }
        for (int i = nq * 4 + t; i < nf; i += 256) {
            const int px = i / 3;
            lds[(i - px * 3) * W2P_PITCH + px] = s[i];
        }
    } else {
        for (int i = t; i < nf; i += 256) {
            const int px = i / 3;
            lds[(i - px * 3) * W2P_PITCH + px] = s[i];
        }
    }
    __syncthreads();
    const int x = t * 4;
    if (x >= n) return;
    const int m = n - x < 4 ? n - x : 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float* const d = dst + (int64_t)c * plane + (int64_t)y * row + x0 + x;
        const float4 v = *reinterpret_cast<const float4*>(&lds[c * W2P_PITCH + x]);
        if (m == 4 && (((uintptr_t)d) & 15u) == 0) *reinterpret_cast<float4*>(d) = v;
        else {
            const float e[4] = {v.x, v.y, v.z, v.w};
            for (int p = 0; p < m; ++p) d[p] = e[p];
        }
    }
}

// ---------------------------------------------------------------- k0 repack [C][V] -> [V][CP]
__global__ void k_repack_k0(const float* __restrict__ in, int C, int CP, int64_t nvox, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nvox * CP) return;
    const int64_t v = t / CP;
    const int ch = (int)(t % CP);
    out[t] = ch < C ? in[(size_t)ch * nvox + v] : 0.f;
}

// DenseGrid.scale_volume_grid (lib/grid.py:130-135): F.interpolate(mode='trilinear', align_corners=True) of a [C][X][Y][Z] grid.
// One thread per output voxel and channel; source index = dst * (in-1)/(out-1) (PyTorch's area_pixel_compute_scale / _source_index
// with align_corners), upper neighbour clamped, weights (1-l, l) per axis, products accumulated as upsample_trilinear3d does.
__global__ void k_resample_trilinear(const float* __restrict__ in, int C, int X, int Y, int Z, float* __restrict__ out, int X2, int Y2, int Z2) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n2 = (int64_t)X2 * Y2 * Z2;
    if (t >= n2 * C) return;
    const int c = (int)(t / n2);
    const int64_t v = t - (int64_t)c * n2;
    const int z2 = (int)(v % Z2), y2 = (int)((v / Z2) % Y2), x2 = (int)(v / ((int64_t)Z2 * Y2));
    const float sx = X2 > 1 ? (float)(X - 1) / (float)(X2 - 1) : 0.f, sy = Y2 > 1 ? (float)(Y - 1) / (float)(Y2 - 1) : 0.f,
                sz = Z2 > 1 ? (float)(Z - 1) / (float)(Z2 - 1) : 0.f;
    const float fx = sx * (float)x2, fy = sy * (float)y2, fz = sz * (float)z2;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = x0 + (x0 < X - 1 ? 1 : 0), y1 = y0 + (y0 < Y - 1 ? 1 : 0), z1 = z0 + (z0 < Z - 1 ? 1 : 0);
    const float lx = fx - (float)x0, ly = fy - (float)y0, lz = fz - (float)z0;
    const float mx = 1.f - lx, my = 1.f - ly, mz = 1.f - lz;
    const float* g = in + (size_t)c * X * Y * Z;
#define K4_G(a, b, d) g[((size_t)(a) * Y + (b)) * Z + (d)]
    out[t] = mx * (my * (mz * K4_G(x0, y0, z0) + lz * K4_G(x0, y0, z1)) + ly * (mz * K4_G(x0, y1, z0) + lz * K4_G(x0, y1, z1))) +
             lx * (my * (mz * K4_G(x1, y0, z0) + lz * K4_G(x1, y0, z1)) + ly * (mz * K4_G(x1, y1, z0) + lz * K4_G(x1, y1, z1)));
#undef K4_G
}

// F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1) > thres as one byte per voxel (lib/dmpigo.py:208-209,221-224,
// lib/dvgo.py:217-219,231-233): a voxel stays occupied iff any of its <= 27 in-grid neighbours has alpha > thres.
__global__ void k_alpha_pool3_gt(const float* __restrict__ alpha, int X, int Y, int Z, float thres, uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)X * Y * Z) return;
    const int z = (int)(t % Z), y = (int)((t / Z) % Y), x = (int)(t / ((int64_t)Z * Y));
    float m = -INFINITY;
    for (int a = max(x - 1, 0); a <= min(x + 1, X - 1); ++a)
        for (int b = max(y - 1, 0); b <= min(y + 1, Y - 1); ++b)
            for (int d = max(z - 1, 0); d <= min(z + 1, Z - 1); ++d) m = fmaxf(m, alpha[((size_t)a * Y + b) * Z + d]);
    out[t] = m > thres ? 1 : 0;
}

// coarse occupancy summary (k4nerf.h): one thread per (cell, z word)
__global__ void k_occ_summary(const uint8_t* __restrict__ mask, int MX, int MY, int MZ, int ncx, int ncy, int zw, uint32_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)ncx * ncy * zw) return;
    const int w = (int)(t % zw);
    const int cy = (int)((t / zw) % ncy), cx = (int)(t / ((int64_t)zw * ncy));
    uint32_t bits = 0u;
    for (int x = cx * K4_OCC_CELL; x < min((cx + 1) * K4_OCC_CELL, MX); ++x)
        for (int y = cy * K4_OCC_CELL; y < min((cy + 1) * K4_OCC_CELL, MY); ++y) {
            const uint8_t* col = mask + ((size_t)x * MY + y) * MZ;
            for (int b = 0; b < 32; ++b) {
                const int z = w * 32 + b;
                if (z < MZ && col[z]) bits |= 1u << b;
            }
        }
    out[t] = bits;
}

// ---------------------------------------------------------------- "live" occupancy mask of the fused marcher (k4nerf.h)
// The reference drops a sample when MaskGrid.forward says "free" (lib/dmpigo.py:308-313) OR when its activated density fails
// alpha > fast_color_thres (lib/dmpigo.py:316-323, lib/dvgo.py:353-360).  mask_cache is a 3^3-dilated bound frozen at some earlier
// time; on the LLFF scene 52 M samples pass it and only 12.6 M pass the alpha test that follows.  The second test can be bounded
// from the density grid itself: trilinear interpolation never exceeds the largest of its 8 corner values, so a cell whose corner
// maximum (+ the largest act_shift plane value it can see) cannot reach the threshold holds no surviving sample.
//   k_cell_live : one thread per density cell (i,j,k) = [i,i+1]x[j,j+1]x[k,k+1]: live iff the bound passes (or any input is NaN)
//   k_live_mask : one thread per MASK voxel: out = mask && OR(live cells a sample that rounds to this voxel can lie in)
// `out` replaces `mask` in the geometry kernel's per-sample lookup: a sample is dropped there iff mask == 0 (the reference drops
// it) or every cell it can lie in is dead (its alpha <= thres: the reference drops it one step later).  Rounding: the fp32
// interpolation is sum_c w_c d_c with w_c >= 0 and sum w_c within 6 ulp of 1, each term rounded at most 8 times, so the computed
// value is <= dmax + |dmax| 2^-20; the bound is evaluated in fp64 with 2^-17-relative and 1e-6-absolute head room (device expf /
// powf / the 1 - 1/(1+e) form are within 2.4e-7 of the true alpha, tests/helpers.py NATIVE_TOL), index boxes carry a slack of
// >= 16 ulp of the coordinate.  Conservative by construction; tests hold the fused kernels bit-identical with and without it.
struct LiveParams {
    const float* density; const float* act_shift; const uint8_t* mask;
    int X, Y, Z, D, MX, MY, MZ;
    double mn[3], len[3];          // xyz_min, xyz_max - xyz_min (fp32 values, widened)
    double ms[3], mt[3];           // xyz2ijk_scale / shift of the MaskGrid
    double shift, interval, thres;
    int mpi;
};

__device__ __forceinline__ bool k4_not_finite_or_nan(float v) { return !(fabsf(v) <= 3.4028234e38f); }

__global__ void k_cell_live(const LiveParams P, uint8_t* __restrict__ cell) {
    const int CX = max(P.X - 1, 1), CY = max(P.Y - 1, 1), CZ = max(P.Z - 1, 1);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)CX * CY * CZ) return;
    const int k = (int)(t % CZ), j = (int)((t / CZ) % CY), i = (int)(t / ((int64_t)CZ * CY));
    const int i1 = min(i + 1, P.X - 1), j1 = min(j + 1, P.Y - 1), k1 = min(k + 1, P.Z - 1);
    float dmax = -INFINITY;
    bool odd = false;                                   // NaN / +inf anywhere: keep the cell
    const int xs[2] = {i, i1}, ys[2] = {j, j1}, zs[2] = {k, k1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float d = P.density[((size_t)xs[a] * P.Y + ys[b]) * P.Z + zs[c]];
                odd |= (d != d) | (d == INFINITY);
                dmax = fmaxf(dmax, d);
            }
    double sb = dmax == -INFINITY ? -INFINITY : (double)dmax + fabs((double)dmax) * 7.62939453125e-6;          // 2^-17
    if (P.mpi) {
        // act_shift planes a sample of this z cell can interpolate between: ua = u_z (D-1)/(Z-1) with u_z in [k, k+1]
        // (D == Z, the reference's configuration: both coordinates are the same fp32 value, the planes are exactly k and k+1)
        int a0 = k, a1 = min(k + 1, P.D - 1);
        if (P.D != P.Z) {
            double lo = 0.0, hi = (double)(P.D - 1);
            if (P.Z > 1) { const double r = (double)(P.D - 1) / (double)(P.Z - 1); lo = (double)k * r; hi = (double)(k + 1) * r; }
            const double eps = 1e-3 + 16.0 * 5.9604644775390625e-8 * (double)P.D;
            a0 = max((int)floor(lo - eps), 0); a1 = min((int)floor(hi + eps) + 1, P.D - 1);
        }
        float amax = -INFINITY;
        for (int a = a0; a <= a1; ++a) { const float v = P.act_shift[a]; odd |= (v != v) | (v == INFINITY); amax = fmaxf(amax, v); }
        sb += amax == -INFINITY ? -INFINITY : (double)amax + fabs((double)amax) * 7.62939453125e-6;
    }
    if (sb != -INFINITY) sb += fabs(sb) * 7.62939453125e-6 + 1e-6;
    // alpha = 1 - (1 + exp(sigma + shift))^(-interval)   (render_utils_kernel.cu:439-441), monotone in sigma
    const double ab = -expm1(-P.interval * log1p(exp(sb + P.shift)));
    const bool live = odd || !(ab * (1.0 + 1e-5) + 1e-6 <= P.thres);
    cell[t] = live ? 1 : 0;
}

__global__ void k_live_mask(const LiveParams P, const uint8_t* __restrict__ cell, uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)P.MX * P.MY * P.MZ) return;
    if (!P.mask[t]) { out[t] = 0; return; }
    const int idx[3] = {(int)(t / ((int64_t)P.MZ * P.MY)), (int)((t / P.MZ) % P.MY), (int)(t % P.MZ)};
    const int dims[3] = {P.X, P.Y, P.Z};
    int c0[3], c1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // world-coordinate interval of the in-bbox points whose MaskGrid index (C round() of p*ms + mt, render_utils_kernel.cu:385-387)
        // is idx[a]; ms == 0 (a 1-voxel axis) maps every point to index round(mt)
        double plo = P.mn[a], phi = P.mn[a] + P.len[a];
        if (P.ms[a] > 0.0) {
            const double e = 0.5 + 16.0 * 5.9604644775390625e-8 * (fabs((double)idx[a]) + 1.0);          // half a voxel + 16 ulp of the index
            plo = ((double)idx[a] - e - P.mt[a]) / P.ms[a];
            phi = ((double)idx[a] + e - P.mt[a]) / P.ms[a];
        }
        // grid_sample coordinate u = (p - min)/(max - min) * (dim - 1)   (lib/grid.py:123, align_corners=True); samples outside the
        // bbox never reach the lookup, so the cell range is clamped to the grid
        const double s = (double)(dims[a] - 1) / P.len[a];
        const double slack = 1e-3 + 16.0 * 5.9604644775390625e-8 * (double)dims[a];
        const int nc = max(dims[a] - 1, 1);
        c0[a] = (int)fmin(fmax(floor((plo - P.mn[a]) * s - slack), 0.0), (double)(nc - 1));
        c1[a] = (int)fmin(fmax(floor((phi - P.mn[a]) * s + slack), 0.0), (double)(nc - 1));
    }
    const int CY = max(P.Y - 1, 1), CZ = max(P.Z - 1, 1);
    bool live = false;
    for (int i = c0[0]; i <= c1[0] && !live; ++i)
        for (int j = c0[1]; j <= c1[1] && !live; ++j)
            for (int k = c0[2]; k <= c1[2]; ++k)
                if (cell[((size_t)i * CY + j) * CZ + k]) { live = true; break; }
    out[t] = live ? 1 : 0;
}

// ================================================================ C ABI
#define ST ((hipStream_t)stream)
#define REQ(c) do { if (!(c)) return K4_ERR_BAD_ARG; } while (0)

extern "C" int k4_sample_ndc_pts_on_rays(const float* o, const float* d, const float* mn, const float* mx,
                                         int64_t n_rays, int32_t n_samples, float* pts, uint8_t* mask, void* stream) {
    REQ(n_rays >= 0 && n_samples >= 2);
    if (n_rays == 0) return K4_OK;                          // (empty tensors carry NULL data pointers)
    REQ(o && d && mn && mx && pts && mask);
    hipLaunchKernelGGL(k_sample_ndc, dim3(k4_blocks(n_rays * n_samples)), dim3(K4_THREADS), 0, ST, o, d, mn, mx, n_rays, n_samples, pts, mask);
    return k4_check_launch();
}
extern "C" int k4_infer_t_minmax(const float* o, const float* d, const float* mn, const float* mx, float near, float far,
                                 int64_t n, float* tmin, float* tmax, void* stream) {
    REQ(o && d && mn && mx && tmin && tmax && n >= 0);
    if (n == 0) return K4_OK;
    hipLaunchKernelGGL(k_t_minmax, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, o, d, mn, mx, near, far, n, tmin, tmax);
    return k4_check_launch();
}
extern "C" int k4_infer_n_samples(const float* d, const float* tmin, const float* tmax, float stepdist, int64_t n,
                                  int64_t* out, void* stream) {
    REQ(d && tmin && tmax && out && n >= 0 && stepdist > 0.f);
    if (n == 0) return K4_OK;
    hipLaunchKernelGGL(k_n_samples, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, d, tmin, tmax, stepdist, n, out);
    return k4_check_launch();
}
extern "C" int k4_infer_ray_start_dir(const float* o, const float* d, const float* tmin, int64_t n, float* start,
                                      float* dir, void* stream) {
    REQ(o && d && tmin && start && dir && n >= 0);
    if (n == 0) return K4_OK;
    hipLaunchKernelGGL(k_start_dir, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, o, d, tmin, n, start, dir);
    return k4_check_launch();
}
extern "C" int k4_sample_pts_on_rays_count(const float* o, const float* d, const float* mn, const float* mx,
                                           float near, float far, float stepdist, int64_t n,
                                           int64_t* nsteps, float* tmin, float* tmax, void* stream) {
    REQ(o && d && mn && mx && nsteps && tmin && tmax && n >= 0 && stepdist > 0.f);
    if (n == 0) return K4_OK;
    hipLaunchKernelGGL(k_pts_count, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, o, d, mn, mx, near, far, stepdist, n, nsteps, tmin, tmax);
    return k4_check_launch();
}
extern "C" int k4_sample_pts_on_rays_fill(const float* o, const float* d, const float* mn, const float* mx,
                                          const float* tmin, const int64_t* cum, float stepdist, int64_t n_rays,
                                          int64_t total, float* pts, uint8_t* mask, int64_t* ray_id, int64_t* step_id,
                                          void* stream) {
    REQ(o && d && mn && mx && tmin && cum && n_rays >= 0 && total >= 0 && stepdist > 0.f);
    if (total == 0) return K4_OK;
    REQ(pts && mask && ray_id && step_id);
    hipLaunchKernelGGL(k_pts_fill, dim3(k4_blocks(total)), dim3(K4_THREADS), 0, ST, o, d, mn, mx, tmin, cum, stepdist, n_rays, total, pts, mask, ray_id, step_id);
    return k4_check_launch();
}
extern "C" int k4_maskcache_lookup(const uint8_t* world, const float* xyz, const float* sc, const float* sh,
                                   int32_t si, int32_t sj, int32_t sk, int64_t n, uint8_t* out, void* stream) {
    REQ(world && sc && sh && n >= 0 && si > 0 && sj > 0 && sk > 0);
    if (n == 0) return K4_OK;
    REQ(xyz && out);
    hipLaunchKernelGGL(k_maskcache, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, world, xyz, sc, sh, si, sj, sk, n, out);
    return k4_check_launch();
}
extern "C" int k4_raw2alpha(const float* den, float shift, float interval, const float* ipp, int64_t n,
                            float* ex, float* al, void* stream) {
    REQ(n >= 0);
    if (n == 0) return K4_OK;
    REQ(den && ex && al);
    hipLaunchKernelGGL(k_raw2alpha, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, den, shift, interval, ipp, n, ex, al);
    return k4_check_launch();
}
extern "C" int k4_raw2alpha_backward(const float* ex, const float* gb, float interval, const float* ipp, int64_t n,
                                     float* g, void* stream) {
    REQ(n >= 0);
    if (n == 0) return K4_OK;
    REQ(ex && gb && g);
    hipLaunchKernelGGL(k_raw2alpha_bwd, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, ex, gb, interval, ipp, n, g);
    return k4_check_launch();
}
extern "C" int k4_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts, int64_t n_rays,
                               float* weight, float* T, float* ainv, int64_t* i_start, int64_t* i_end, void* stream) {
    REQ(n_pts >= 0 && n_rays >= 0);
    if (n_rays == 0) return K4_OK;                          // (empty tensors carry NULL data pointers)
    REQ(ainv && i_start && i_end);
    // weight = zeros_like, T = ones_like, alphainv_last = ones, i_start/i_end = zeros (.cu:624-628)
    hipLaunchKernelGGL(k_fill2, dim3(k4_blocks(n_rays)), dim3(K4_THREADS), 0, ST, ainv, 1.f, (float*)nullptr, 0.f, n_rays);
    hipLaunchKernelGGL(k_fill_i64, dim3(k4_blocks(n_rays)), dim3(K4_THREADS), 0, ST, i_start, i_end, n_rays);
    if (n_pts == 0) return k4_check_launch();
    REQ(alpha && ray_id && weight && T);
    hipLaunchKernelGGL(k_fill2, dim3(k4_blocks(n_pts)), dim3(K4_THREADS), 0, ST, weight, 0.f, T, 1.f, n_pts);
    hipLaunchKernelGGL(k_seg_bounds, dim3(k4_blocks(n_pts)), dim3(K4_THREADS), 0, ST, ray_id, n_pts, i_start, i_end);
    hipLaunchKernelGGL(k_alpha2weight, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, ST, alpha, n_rays, weight, T, ainv, i_start, i_end);
    return k4_check_launch();
}
extern "C" int k4_alpha2weight_backward(const float* alpha, const float* weight, const float* T, const float* ainv,
                                        const int64_t* i_start, const int64_t* i_end, int64_t n_rays, int64_t n_pts,
                                        const float* gw, const float* gl, float* grad, void* stream) {
    REQ(n_rays >= 0 && n_pts >= 0);
    if (n_pts == 0 || n_rays == 0) return K4_OK;
    REQ(alpha && weight && T && ainv && i_start && i_end && gw && gl && grad);
    hipLaunchKernelGGL(k_fill2, dim3(k4_blocks(n_pts)), dim3(K4_THREADS), 0, ST, grad, 0.f, (float*)nullptr, 0.f, n_pts);
    hipLaunchKernelGGL(k_alpha2weight_bwd, dim3(k4_blocks(n_rays * 64)), dim3(K4_THREADS), 0, ST, alpha, weight, T, ainv, i_start, i_end, n_rays, gw, gl, grad);
    return k4_check_launch();
}
extern "C" int k4_grid_sample_3d(const float* grid, int32_t C, int32_t X, int32_t Y, int32_t Z, const float* xyz,
                                 const float* mn, const float* mx, int64_t n, float* out, void* stream) {
    REQ(grid && mn && mx && C > 0 && X > 0 && Y > 0 && Z > 0 && n >= 0);
    if (n == 0) return K4_OK;
    REQ(xyz && out);
    hipLaunchKernelGGL(k_grid_sample, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, grid, C, X, Y, Z, xyz, mn, mx, n, out);
    return k4_check_launch();
}
extern "C" int k4_segment_sum(const float* src, const int64_t* index, int64_t n, int32_t C, int64_t n_seg, float* out,
                              void* stream) {
    REQ(C > 0 && n >= 0 && n_seg >= 0 && (out || n_seg == 0));
    if (n_seg > 0) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)n_seg * C, ST);
        if (e != hipSuccess) return (int)e;
    }
    if (n == 0) return K4_OK;
    REQ(src && index);
    hipLaunchKernelGGL(k_segment_sum, dim3(k4_blocks(n * C)), dim3(K4_THREADS), 0, ST, src, index, n, C, out);
    return k4_check_launch();
}
extern "C" int k4_grid_sample_3d_backward(const float* grad_out, int32_t C, int32_t X, int32_t Y, int32_t Z, const float* xyz,
                                          const float* mn, const float* mx, int64_t n, float* grad_grid, void* stream) {
    REQ(C > 0 && X > 0 && Y > 0 && Z > 0 && mn && mx && n >= 0 && grad_grid);
    if (n == 0) return K4_OK;
    REQ(xyz && grad_out);
    hipLaunchKernelGGL(k_grid_sample_bwd, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, grad_out, C, X, Y, Z, xyz, mn, mx, n, grad_grid);
    return k4_check_launch();
}
extern "C" int64_t k4_grid_sample_3d_backward_workspace_bytes(int32_t C, int32_t X, int32_t Y, int32_t Z) {
    if (C <= 1 || C > 32 || X <= 0 || Y <= 0 || Z <= 0) return -1;          // one channel: the layouts coincide, use k4_grid_sample_3d_backward
    const int64_t nvox = (int64_t)X * Y * Z;
    return nvox * C * 4 + ((nvox + 15) / 16) * 16;
}
// scatter and sweep apart: a caller that consumes the touched voxels' sums where they lie (k4_masked_adam_upd_sparse_cl, k4_opt.hip) runs the scatter alone
extern "C" int k4_grid_sample_3d_backward_cl_scatter(const float* grad_out, int32_t C, int32_t X, int32_t Y, int32_t Z, const float* xyz,
                                                     const float* mn, const float* mx, int64_t n, void* workspace, void* stream) {
    REQ(C > 1 && C <= 32 && X > 0 && Y > 0 && Z > 0 && mn && mx && n >= 0 && workspace && (((uintptr_t)workspace) & 15) == 0);
    if (n == 0) return K4_OK;
    REQ(xyz && grad_out);
    const int64_t nvox = (int64_t)X * Y * Z;
    float* const scratch = (float*)workspace;
    uint8_t* const flags = (uint8_t*)workspace + nvox * C * 4;
#define K4_GSB_CL(LPS) hipLaunchKernelGGL(k_gsb_cl_scatter<LPS>, dim3(k4_blocks(n * LPS)), dim3(K4_THREADS), 0, ST, grad_out, C, X, Y, Z, xyz, mn, mx, n, scratch, flags)
    if (C <= 2) K4_GSB_CL(2); else if (C <= 4) K4_GSB_CL(4); else if (C <= 8) K4_GSB_CL(8); else if (C <= 16) K4_GSB_CL(16); else K4_GSB_CL(32);
#undef K4_GSB_CL
    return k4_check_launch();
}
extern "C" int k4_grid_flag_corners(int32_t X, int32_t Y, int32_t Z, const float* xyz, const float* mn, const float* mx, int64_t n, uint8_t* flags, void* stream) {
    REQ(X > 0 && Y > 0 && Z > 0 && mn && mx && n >= 0 && flags);
    if (n == 0) return K4_OK;
    REQ(xyz);
    hipLaunchKernelGGL(k_gsb_flag_corners, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, X, Y, Z, xyz, mn, mx, n, flags);
    return k4_check_launch();
}
extern "C" int k4_grid_sample_3d_backward_cl_sweep(int32_t C, int32_t X, int32_t Y, int32_t Z, void* workspace, float* grad_grid, void* stream) {
    REQ(C > 1 && C <= 32 && X > 0 && Y > 0 && Z > 0 && grad_grid && workspace && (((uintptr_t)workspace) & 15) == 0);
    const int64_t nvox = (int64_t)X * Y * Z;
    hipLaunchKernelGGL(k_gsb_cl_sweep, dim3(k4_blocks(nvox)), dim3(K4_THREADS), 0, ST, (float*)workspace, (uint8_t*)workspace + nvox * C * 4, C, nvox, grad_grid);
    return k4_check_launch();
}
extern "C" int k4_grid_sample_3d_backward_cl(const float* grad_out, int32_t C, int32_t X, int32_t Y, int32_t Z, const float* xyz,
                                             const float* mn, const float* mx, int64_t n, float* grad_grid, void* workspace, void* stream) {
    REQ(C > 1 && C <= 32 && X > 0 && Y > 0 && Z > 0 && mn && mx && n >= 0 && grad_grid && workspace && (((uintptr_t)workspace) & 15) == 0);
    if (n == 0) return K4_OK;
    const int rc = k4_grid_sample_3d_backward_cl_scatter(grad_out, C, X, Y, Z, xyz, mn, mx, n, workspace, stream);
    if (rc) return rc;
    return k4_grid_sample_3d_backward_cl_sweep(C, X, Y, Z, workspace, grad_grid, stream);
}
extern "C" int k4_segment_sum_backward(const float* grad_out, const int64_t* index, int64_t n, int32_t C, float* grad_src,
                                       void* stream) {
    REQ(C > 0 && n >= 0);
    if (n == 0) return K4_OK;
    REQ(grad_out && index && grad_src);
    hipLaunchKernelGGL(k_segment_gather, dim3(k4_blocks(n * C)), dim3(K4_THREADS), 0, ST, grad_out, index, n, C, grad_src);
    return k4_check_launch();
}
extern "C" int k4_get_rays_of_a_view(int32_t H, int32_t W, const float* K_dev, const float* c2w_dev, int32_t ndc,
                                     int32_t inverse_y, int32_t flip_x, int32_t flip_y, int32_t mode_center,
                                     float focal, float* rays_o, float* rays_d, float* viewdirs, void* stream) {
    REQ(H > 0 && W > 0 && K_dev && c2w_dev && rays_o && rays_d && viewdirs);
    // -1./(W/(2.*focal)) as the reference evaluates it: Python double arithmetic, cast to fp32 by the tensor multiply
    const float c_w = (float)(-1.0 / ((double)W / (2.0 * (double)focal)));
    const float c_h = (float)(-1.0 / ((double)H / (2.0 * (double)focal)));
    hipLaunchKernelGGL(k_rays_of_view, dim3(k4_blocks((int64_t)H * W)), dim3(K4_THREADS), 0, ST, H, W, K_dev, c2w_dev, ndc, inverse_y,
                       flip_x, flip_y, mode_center ? 0.5f : 0.f, c_w, c_h, rays_o, rays_d, viewdirs);
    return k4_check_launch();
}
extern "C" int k4_nhwc_window_to_planar(const float* src, int32_t src_w, int32_t channels, int32_t oy, int32_t ox, int32_t th, int32_t tw,
                                        float* dst, int64_t dst_plane_stride, int64_t dst_row_stride, void* stream) {
    REQ(src && dst && src_w > 0 && channels >= 1 && channels <= 4 && oy >= 0 && ox >= 0 && th >= 0 && tw >= 0 && ox + tw <= src_w && dst_row_stride >= tw);
    if (th == 0 || tw == 0) return K4_OK;
    if (channels == 3 && th <= 65535) {
        hipLaunchKernelGGL(k_nhwc3_window_to_planar, dim3((tw + W2P_PX - 1) / W2P_PX, th), dim3(256), 0, ST, src, src_w, oy, ox, tw, dst, dst_plane_stride, dst_row_stride);
        return k4_check_launch();
    }
    hipLaunchKernelGGL(k_nhwc_window_to_planar, dim3(k4_blocks((int64_t)th * ((tw + 3) / 4))), dim3(K4_THREADS), 0, ST, src, src_w, channels, oy, ox, th, tw, dst,
                       dst_plane_stride, dst_row_stride);
    return k4_check_launch();
}
extern "C" int k4_to8b(const float* x, int64_t n, uint8_t* out, void* stream) {
    REQ(n >= 0);
    if (n == 0) return K4_OK;
    REQ(x && out);
    hipLaunchKernelGGL(k_to8b, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, x, n, out);
    return k4_check_launch();
}
extern "C" int k4_resample_trilinear(const float* in, int32_t channels, int32_t x, int32_t y, int32_t z,
                                     float* out, int32_t x2, int32_t y2, int32_t z2, void* stream) {
    REQ(in && out && channels > 0 && x > 0 && y > 0 && z > 0 && x2 > 0 && y2 > 0 && z2 > 0);
    hipLaunchKernelGGL(k_resample_trilinear, dim3(k4_blocks((int64_t)channels * x2 * y2 * z2)), dim3(K4_THREADS), 0, ST, in, channels, x, y, z, out, x2, y2, z2);
    return k4_check_launch();
}
extern "C" int k4_alpha_maxpool3_gt(const float* alpha, int32_t x, int32_t y, int32_t z, float thres, uint8_t* out, void* stream) {
    REQ(alpha && out && x > 0 && y > 0 && z > 0);
    hipLaunchKernelGGL(k_alpha_pool3_gt, dim3(k4_blocks((int64_t)x * y * z)), dim3(K4_THREADS), 0, ST, alpha, x, y, z, thres, out);
    return k4_check_launch();
}
// the four tables of 64-bit windows (k4nerf.h) from the base words (kept behind the tables in the same buffer)
__global__ void k_occ_windows(const uint32_t* __restrict__ base, int ncx, int ncy, int zw, unsigned long long* __restrict__ out) {
    const int64_t n = (int64_t)ncx * ncy * zw;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * n) return;
    const int tab = (int)(t / n);
    const int64_t e = t - (int64_t)tab * n;
    const int w = (int)(e % zw), cy = (int)((e / zw) % ncy), cx = (int)(e / ((int64_t)zw * ncy));
    unsigned long long v = 0ull;
    for (int dx = 0; dx <= (tab & 1); ++dx)
        for (int dy = 0; dy <= (tab >> 1); ++dy) {
            const int x = cx + dx, y = cy + dy;
            if (x >= ncx || y >= ncy) continue;
            const uint32_t* p = base + ((size_t)x * ncy + y) * zw;
            v |= (unsigned long long)p[w] | (w + 1 < zw ? (unsigned long long)p[w + 1] << 32 : 0ull);
        }
    out[t] = v;
}
extern "C" int64_t k4_occupancy_summary_bytes(int32_t mx, int32_t my, int32_t mz) {
    if (mx <= 0 || my <= 0 || mz <= 0) return -1;
    // 4 tables of 8-byte windows + the base words (build scratch, kept behind the tables)
    return (int64_t)((mx + K4_OCC_CELL - 1) / K4_OCC_CELL) * ((my + K4_OCC_CELL - 1) / K4_OCC_CELL) * ((mz + 31) / 32) * (4 * 8 + 4);
}
extern "C" int k4_build_occupancy_summary(const uint8_t* mask, int32_t mx, int32_t my, int32_t mz, uint32_t* out, void* stream) {
    REQ(mask && out && mx > 0 && my > 0 && mz > 0);
    const int ncx = (mx + K4_OCC_CELL - 1) / K4_OCC_CELL, ncy = (my + K4_OCC_CELL - 1) / K4_OCC_CELL, zw = (mz + 31) / 32;
    const int64_t n = (int64_t)ncx * ncy * zw;
    uint32_t* const base = out + 8 * n;                                   // behind the four tables
    hipLaunchKernelGGL(k_occ_summary, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, mask, mx, my, mz, ncx, ncy, zw, base);
    hipLaunchKernelGGL(k_occ_windows, dim3(k4_blocks(4 * n)), dim3(K4_THREADS), 0, ST, base, ncx, ncy, zw, reinterpret_cast<unsigned long long*>(out));
    return k4_check_launch();
}
extern "C" int64_t k4_live_mask_workspace_bytes(int32_t x, int32_t y, int32_t z) {
    if (x <= 0 || y <= 0 || z <= 0) return -1;
    return (int64_t)(x > 1 ? x - 1 : 1) * (y > 1 ? y - 1 : 1) * (z > 1 ? z - 1 : 1);
}
extern "C" int k4_build_live_mask(const k4_grid_desc* g, float act_shift_scalar, float interval, float fast_color_thres,
                                  uint8_t* workspace, uint8_t* out_mask, void* stream) {
    REQ(g && g->density && g->mask && workspace && out_mask);
    REQ(g->dims[0] > 0 && g->dims[1] > 0 && g->dims[2] > 0 && g->mask_dims[0] > 0 && g->mask_dims[1] > 0 && g->mask_dims[2] > 0);
    REQ(fast_color_thres > 0.f && interval > 0.f);
    REQ(!g->act_shift || g->act_depth > 0);
    LiveParams P{};
    P.density = g->density; P.act_shift = g->act_shift; P.mask = g->mask;
    P.X = g->dims[0]; P.Y = g->dims[1]; P.Z = g->dims[2]; P.D = g->act_depth;
    P.MX = g->mask_dims[0]; P.MY = g->mask_dims[1]; P.MZ = g->mask_dims[2];
    for (int a = 0; a < 3; ++a) {
        P.mn[a] = (double)g->xyz_min[a];
        P.len[a] = (double)(g->xyz_max[a] - g->xyz_min[a]);                  // the fp32 difference the kernels divide by (lib/grid.py:123)
        REQ(P.len[a] > 0.0);
        P.ms[a] = (double)g->xyz2ijk_scale[a]; P.mt[a] = (double)g->xyz2ijk_shift[a];
    }
    P.shift = (double)act_shift_scalar; P.interval = (double)interval; P.thres = (double)fast_color_thres;
    P.mpi = g->act_shift != nullptr;
    const int64_t ncell = k4_live_mask_workspace_bytes(P.X, P.Y, P.Z);
    hipLaunchKernelGGL(k_cell_live, dim3(k4_blocks(ncell)), dim3(K4_THREADS), 0, ST, P, workspace);
    hipLaunchKernelGGL(k_live_mask, dim3(k4_blocks((int64_t)P.MX * P.MY * P.MZ)), dim3(K4_THREADS), 0, ST, P, workspace, out_mask);
    return k4_check_launch();
}
// One thread per (x, y) column, z ascending: stop plane of a ray along the column + the column's alpha-passing voxels per eighth of the depth.
__global__ void k_mpi_depth_split_stats(const float* __restrict__ density, const float* __restrict__ act_shift, int X, int Y, int Z, int D,
                                        float interval, float thres, float* __restrict__ acc) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= X * Y) return;
    const float* col = density + (size_t)t * Z;
    float T = 1.f;
    int zs = Z;                                                   // first plane BEHIND the stop (Z: the column never stops)
    float cnt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < Z; ++z) {
        // act_shift grid [D] sampled at the plane's normalised depth (align_corners): D == Z in every configuration, else nearest
        const float sh = act_shift ? act_shift[D == Z ? z : min(D - 1, (int)((float)z * (float)(D - 1) / (float)max(Z - 1, 1) + 0.5f))] : 0.f;
        float e_unused, a;
        k4s_raw2alpha(col[z] + sh, 0.f, interval, e_unused, a);
        if (a > thres) cnt[min(7, z * 8 / Z)] += 1.f;
        if (zs == Z) { T = fmaf(-T, a, T); if (T < 1e-3f) zs = z + 1; }
    }
    float tot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) tot += cnt[e];
    if (tot == 0.f) return;
    atomicAdd(&acc[8], tot); atomicAdd(&acc[9], 1.f);
    if (zs < Z) atomicAdd(&acc[0], 1.f);
    float behind = 0.f;
#pragma unroll
    for (int b = 7; b >= 1; --b) {
        behind += cnt[b];                                          // voxels at or behind plane b * Z / 8
        if (zs <= b * Z / 8) atomicAdd(&acc[b], behind);
    }
}
__global__ void k_mpi_depth_split_norm(float* acc) {
    const int b = (int)threadIdx.x;
    if (b >= 1 && b < 8) acc[b] = acc[8] > 0.f ? acc[b] / acc[8] : 0.f;
    __syncthreads();
    if (b == 0) acc[0] = acc[9] > 0.f ? acc[0] / acc[9] : 0.f;
}
extern "C" int k4_mpi_depth_split_stats(const k4_grid_desc* g, float interval, float fast_color_thres, float* out16, void* stream) {
    REQ(g && g->density && g->act_shift && g->act_depth > 0 && out16 && interval > 0.f);
    REQ(g->dims[0] > 0 && g->dims[1] > 0 && g->dims[2] > 0 && (int64_t)g->dims[0] * g->dims[1] < 0x7fffffff);
    if (hipMemsetAsync(out16, 0, 16 * sizeof(float), ST) != hipSuccess) return K4_ERR_BAD_ARG;
    const int n = g->dims[0] * g->dims[1];
    hipLaunchKernelGGL(k_mpi_depth_split_stats, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ST, g->density, g->act_shift, g->dims[0], g->dims[1], g->dims[2],
                       g->act_depth, interval, fast_color_thres, out16);
    hipLaunchKernelGGL(k_mpi_depth_split_norm, dim3(1), dim3(64), 0, ST, out16);
    return k4_check_launch();
}
extern "C" int k4_train_select_mpi(const float* rays_o, const float* rays_d, const float* xyz_min, const float* xyz_max, int64_t n_rays, int32_t n_samples,
                                   const uint8_t* mask, const float* xyz2ijk_scale, const float* xyz2ijk_shift, int32_t mi, int32_t mj, int32_t mk,
                                   const float* density, int32_t X, int32_t Y, int32_t Z, const float* act_shift, int32_t act_depth,
                                   float interval, float fast_color_thres,
                                   int16_t* steps2, uint8_t* keep3, int64_t* cnt2, int64_t* cnt3, void* stream) {
    REQ(n_rays >= 0 && n_samples >= 2 && n_samples <= 32767 && mi > 0 && mj > 0 && mk > 0 && X > 0 && Y > 0 && Z > 0 && act_depth > 0 && fast_color_thres > 0.f);
    if (n_rays == 0) return K4_OK;
    REQ(rays_o && rays_d && xyz_min && xyz_max && mask && xyz2ijk_scale && xyz2ijk_shift && density && act_shift && steps2 && keep3 && cnt2 && cnt3);
    hipLaunchKernelGGL(k_train_select_mpi, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, ST, rays_o, rays_d, xyz_min, xyz_max, n_rays, n_samples,
                       mask, xyz2ijk_scale, xyz2ijk_shift, mi, mj, mk, density, X, Y, Z, act_shift, act_depth, interval, fast_color_thres,
                       steps2, keep3, cnt2, cnt3);
    return k4_check_launch();
}
extern "C" int k4_train_compact(const int16_t* steps2, const uint8_t* keep3, const int64_t* cnt2, const int64_t* cumsum2, const int64_t* cumsum3,
                                int64_t n_rays, int32_t n_samples, int64_t* ray_id, int64_t* step_id, int64_t* idx3, void* stream) {
    REQ(n_rays >= 0 && n_samples >= 2);
    if (n_rays == 0) return K4_OK;
    REQ(steps2 && keep3 && cnt2 && cumsum2 && cumsum3 && ray_id && step_id && idx3);
    hipLaunchKernelGGL(k_train_compact, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, ST, steps2, keep3, cnt2, cumsum2, cumsum3, n_rays, n_samples,
                       ray_id, step_id, idx3);
    return k4_check_launch();
}
extern "C" int k4_ndc_points_of(const float* rays_o, const float* rays_d, const int64_t* ray_id, const int64_t* step_id, int64_t n, int32_t n_samples,
                                float* pts, void* stream) {
    REQ(n >= 0 && n_samples >= 2);
    if (n == 0) return K4_OK;
    REQ(rays_o && rays_d && ray_id && step_id && pts);
    hipLaunchKernelGGL(k_ndc_points_of, dim3(k4_blocks(n)), dim3(K4_THREADS), 0, ST, rays_o, rays_d, ray_id, step_id, n, n_samples, pts);
    return k4_check_launch();
}
extern "C" int k4_repack_k0(const float* in, int32_t C, int32_t CP, int64_t nvox, float* out, void* stream) {
    REQ(in && out && C > 0 && CP >= C && CP % 4 == 0 && nvox > 0);
    hipLaunchKernelGGL(k_repack_k0, dim3(k4_blocks(nvox * CP)), dim3(K4_THREADS), 0, ST, in, C, CP, nvox, out);
    return k4_check_launch();
}
