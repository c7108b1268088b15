// Voxel-grid ray marcher for gfx950 (MI355X): two launches per call, nothing else.
//
// Together they compute, per ray, what the reference spreads over ~25 launches and >=4 host syncs per
// 8192-ray chunk inside DirectMPIGO.forward (lib/dmpigo.py:292-427) / DirectVoxGO.forward
// (lib/dvgo.py:327-448):
//   sampler (render_utils_kernel.cu:245-270 NDC, :12-242 ray-AABB) -> bbox test -> occupancy byte
//   (:374-392) -> trilinear density (+ per-plane act_shift grid for MPI) (lib/grid.py:117-128) ->
//   raw2alpha (:431-443) -> alpha>thres -> sequential transmittance with the T<1e-3 early stop
//   (:577-605) -> w>thres -> trilinear k0 features + PE + rgbnet MLP + sigmoid
//   (lib/dmpigo.py:336-379, lib/dvgo.py:372-412) -> sum_w rgb, sum_w s, alphainv_last*bg
//   (lib/dmpigo.py:382-398,418-424).
//
// K1  k4_geom3_kernel  (VALU-issue / latency bound; HBM traffic ~0.2 GB per frame)
//     a WORKGROUP owns a bundle of 64 rays (an 8x8 pixel tile when the caller says the rays are an image, else 64
//     consecutive rays; workgroups are XCD-banded so neighbouring tiles share an L2); its 4 waves take one depth quarter of
//     every ray each and walk ONE ray at a time with lanes = 64 consecutive samples.  NDC/MPI rays advance mostly along Z,
//     the contiguous axis of the [X][Y][Z] grids, so the occupancy byte and the density gathers of a wave coalesce into a
//     few 128-B lines that the neighbouring ray re-uses from L1/L2.  Mask-passing samples are compacted across rays
//     before the density stage (all lanes busy); transmittance is the reference's exact sequential product, run
//     transposed (lane = ray) by wave 0 after a workgroup barrier.  Survivors (w > thres) are compacted straight into the
//     bundle's slice of a workspace as 8-byte {ray,step | weight} records -- ~9 per ray instead of the reference's
//     256 x 60 B of per-sample intermediates.
// K2  k4_shade_kernel  (VALU / gather bound)
//     persistent waves pull bundles from a queue; 64 records at a time, lane = sample: 8-corner k0 gather from the
//     channel-last repack, features to LDS, then the rgbnet MLP on the matrix cores as C^T[neuron][sample] = W . X --
//     by default with exact 3-term bf16 splits on v_mfma_f32_32x32x16_bf16 (fp32-equivalent), optionally with
//     v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).  Weights are pre-arranged in operand order and LDS resident; the C
//     layout of layer n IS the B-operand layout of layer n+1 (K is walked in accumulator-register order), so activations
//     never leave registers and there is no cross-lane traffic between layers.  A segmented wave scan folds w*rgb / w*s
//     into per-ray LDS accumulators; 5 floats per ray are the only other global writes.
// Superseded variants (a per-ray K1 with every stage on the same lanes, a K1 with global / per-XCD work queues, a K2 with
// LDS-DMA prefetch of the next batch's gathers at one workgroup per CU) were measured and dropped: DESIGN.md 5, profiles/.
// Deterministic: no global atomics on the data path, fixed summation order.
#include "k4_common.h"
#include <string.h>
#include <stdlib.h>
#include <type_traits>

#define MODE_MPI  0
#define MODE_DVGO 1

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MarchParams {
    const float* rays_o; const float* rays_d; const float* viewdirs;
    int n_rays; int img_w; int img_h;
    const float* density; const float* k0; const float* act_shift; const uint8_t* mask;
    const unsigned long long* occ;      // coarse occupancy summary (k4_build_occupancy_summary: four tables of 64-bit z windows) or NULL
    int X, Y, Z; int C; int CP; int k0_layout; int act_d;
    int MX, MY, MZ;
    float minx, miny, minz, maxx, maxy, maxz;
    float msx, msy, msz, mtx, mty, mtz;
    float lenx, leny, lenz, rlenx, rleny, rlenz;     // xyz_max - xyz_min and its correctly rounded reciprocal
    const float* mlp; int mlp_floats; int mlp_floats_b3; int mlp_floats_b2; int dim0; int k1p; int vpe; int spe; int k0_skip;
    int n_samples;          // MPI: samples per ray
    int max_steps;          // capacity per ray in the workspace
    int ent_stride;         // records per bundle in the workspace: 64 * max_steps rounded up to 256 (4 depth quarters)
    int depth_n;            // denominator of s = (k+0.5)/depth_n
    float nsm1;             // MPI: (float)(n_samples-1)
    float stepdist, near_, far_, shift, interval, thres, bg;
    float depth_n_inv;              // 1 / depth_n, correctly rounded
    float depth_fx, depth_fx_inv;   // fixed-point scale of the per-ray depth sums (a power of two: 2^30 / the largest s a record can carry) and its reciprocal
    uint2* entries; int* counts; int* qhead;       // workspace: [n_bundles][64*max_steps], [n_bundles], shading work-queue head
    int2* jobs;                                    // workspace: the shading queue = {bundle id, its record count}, most batches first (k4_order_kernel)
    int n_bundles;
    int split_k;            // MPI: > 0 = depth-ordered geometry stage -- samples [0, split_k) in a first launch, [split_k, n_samples) in a second one that
                            // drops every ray the first launch's transmittance scan stopped (k4_grid_desc.depth_split; a multiple of 64)
    int slab;               // 0: the whole depth range in one launch; 1 / 2: first / second launch of the split form
    int debug;              // K4_DEBUG ablation bits (profiling only; 0 in production)
    int serp;               // 1: serpentine ray order inside a tile (default)
    int band_blocks;        // geometry kernel: blocks per XCD band (0: one contiguous band per XCD)
    float* out_rgb; float* out_depth; float* out_ainv; unsigned long long* counters;
    unsigned long long* timing;     // K4_GEOM_TIMING builds: [9] stage ticks + wave count (the kernel runs WITHOUT the counting instantiation)
};

template <int MODE>
__device__ __forceinline__ float step_t(const MarchParams& P, int k) {
    // MPI : dist = (float)i_step / (N_samples-1)      render_utils_kernel.cu:260
    // DVGO: dist = stepdist * i_step                  render_utils_kernel.cu:184
    return MODE == MODE_MPI ? (float)k / P.nsm1 : P.stepdist * (float)k;
}

struct Bundle { int x, y, lin, id; };

__device__ __forceinline__ Bundle bundle_from_id(const MarchParams& P, int id) {
    Bundle b;
    const int wg = id >> 2, wv = id & 3;
    b.id = id;
    b.x = b.y = b.lin = 0;
    if (P.img_w > 0) {
        const int wgx = (P.img_w + 15) >> 4;
        b.x = (wg % wgx) * 16 + (wv & 1) * 8;
        b.y = (wg / wgx) * 16 + (wv >> 1) * 8;
    } else {
        b.lin = b.id * 64;
    }
    return b;
}
__device__ __forceinline__ Bundle bundle_of(const MarchParams& P, int wv) {
    return bundle_from_id(P, k4_xcd_remap((int)blockIdx.x, (int)gridDim.x) * 4 + wv);
}
__device__ __forceinline__ int ray_index(const MarchParams& P, const Bundle& b, int r) {
    if (P.img_w > 0) {
        // boustrophedon order inside the 8x8 tile: consecutive ray slots are always neighbouring pixels, so the grid rows
        // a ray touches were touched by the previous slot too (row reuse distance = one ray)
        const int py = b.y + (r >> 3);
        const int px = b.x + ((P.serp && ((r >> 3) & 1)) ? 7 - (r & 7) : (r & 7));
        return (px < P.img_w && py < P.img_h) ? py * P.img_w + px : -1;
    }
    const int ray = b.lin + r;
    return ray < P.n_rays ? ray : -1;
}

// p_k = start + dir * t_k.  MPI: (o, d).  DVGO: infer_t_minmax / infer_n_samples / infer_ray_start_dir
// (render_utils_kernel.cu:12-79).  Shared by both kernels so that they agree bit for bit.
template <int MODE>
__device__ __forceinline__ void ray_setup(const MarchParams& P, float ox, float oy, float oz, float vx, float vy, float vz,
                                          float& sx, float& sy, float& sz, float& dx, float& dy, float& dz, int& nsteps) {
    if (MODE == MODE_MPI) {
        sx = ox; sy = oy; sz = oz; dx = vx; dy = vy; dz = vz;
        nsteps = P.n_samples;
    } else {
        const float ex = (vx == 0.f) ? 1e-6f : vx, ey = (vy == 0.f) ? 1e-6f : vy, ez = (vz == 0.f) ? 1e-6f : vz;
        const float ax = (P.maxx - ox) / ex, ay = (P.maxy - oy) / ey, az = (P.maxz - oz) / ez;
        const float bx = (P.minx - ox) / ex, by = (P.miny - oy) / ey, bz = (P.minz - oz) / ez;
        const float t_min = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), P.far_), P.near_);
        const float t_max = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), P.far_), P.near_);
        const float rnorm = sqrtf(fmaf(vz, vz, fmaf(vy, vy, vx * vx)));
        const float nf = fmaxf(ceilf((t_max - t_min) * rnorm / P.stepdist), 1.f);     // at least 1 point (:53)
        nsteps = (int)fminf(nf, (float)P.max_steps);
        sx = fmaf(vx, t_min, ox); sy = fmaf(vy, t_min, oy); sz = fmaf(vz, t_min, oz);
        dx = vx / rnorm; dy = vy / rnorm; dz = vz / rnorm;
    }
}

// =====================================================================================================
// K1: geometry
// =====================================================================================================
// -----------------------------------------------------------------------------------------------------
// K1.  History (profiles/r01_*): the first, per-ray form (one wave walks a ray, every stage on the same 64 lanes) was
// VALU-issue bound with most lanes idle; v2 compacted mask-passing samples across rays before the density stage and ran
// the transmittance scan transposed; v3 split a bundle's depth range over the 4 waves of a workgroup (L2 hit 75 -> 96 %).
// In v3 the occupancy stage alone was ~0.6 of 1.0 ms although only 27 % of the in-bbox samples pass the mask: 141 M samples
// per frame paid 3 FMAs + 3 round() + bounds + a byte load to be dropped.  v4 (this kernel) does not visit them:
//   P. probe, lane = RAY: for each group of 16 consecutive samples of the wave's depth quarter the two END samples' occupancy
//      indices are computed exactly as the per-sample code would; the index map is monotone along a ray, so every sample of the
//      group lies in the per-axis interval they span.  The box is tested against a coarse summary of the mask (one bit per
//      8x8 (x,y) cell and z plane, k4_build_occupancy_summary, 76 KB for the LLFF grid, L2 resident; on the LLFF scene 41 % of the
//      groups are kept where 36 % hold an occupied sample; 4x4 cells would need 3x3 of them for 39 %): all bits clear => no sample of the group
//      can pass MaskGrid.forward => the group is never enumerated.  Boxes wider than 2x2 cells / 32 planes are kept unseen.
//      Skipping is conservative by construction: results are bit-identical with and without the summary (tests).
//   A. occupancy, lanes = 4 kept groups x 16 consecutive samples (ray-major, depth-ascending order as before): sample point,
//      closed-bbox test, MaskGrid index with C round(), occupancy byte; 4 items' bytes are fetched together (one round trip per
//      256 samples); mask-passing samples are compacted (ballot/mbcnt) into an LDS ring;
//   B. density / act_shift / raw2alpha on 64 ring entries at a time, 2 per lane, issued one group early (fetches fly under the
//      next group's occupancy round trip); alpha-passing samples are appended to the bundle's workspace slice as {ray,step|alpha};
//   C. transposed transmittance scan, lane = RAY: the reference's exact sequential product, writing w over alpha;
//   D. in-place compaction of the w > thres survivors -> the records the shading kernel consumes.
// -----------------------------------------------------------------------------------------------------
#ifndef K4_GEOM_MIN_WG
#define K4_GEOM_MIN_WG 1
#endif
#ifndef K4_SHADE_WG_PER_CU
#define K4_SHADE_WG_PER_CU 2      // 256 VGPRs per wave: at 3 (168 VGPRs) the batch loop spilled ~90 dwords and its scratch reloads cost 0.4 ms/frame
#endif
#ifndef K4_SHADE_SCAN
#define K4_SHADE_SCAN 0           // 1: per-ray sums by a segmented wave scan + one LDS atomic per run (the earlier form)
#endif
#define K4_RING 512
#define K4_TKTAB 256              // MPI: k/(Ns-1) for k < K4_TKTAB is tabulated once per workgroup (an IEEE division costs ~10 VALU per sample)
#define K4_ELIST 512              // kept (ray, group) entries enumerated at a time (ray sub-batches when 64 rays x groups exceed it)
#define K4_GRP 16                 // samples per skip group
struct Geom2Lds {
    float raytab[64][8];     // [0..2] start xyz, [3] bits(kq0) | [4..6] dir xyz, [7] bits(kq1): the bundle's rays and THIS wave's depth range [kq0,kq1) of each
    unsigned qk[K4_RING];    // ring of mask-passing samples: ray_local<<24 | step (residual < 64 + one group of 256)
    int acnt[64];            // alpha-passing samples per ray
    unsigned short elist[K4_ELIST];   // kept entries, ray-major / depth-ascending: ray_local<<10 | group
};

template <int MODE>
__device__ __forceinline__ float tk_of(const MarchParams& P, const float* tktab, int k) {
    if (MODE == MODE_MPI) return k < K4_TKTAB ? tktab[k] : (float)k / P.nsm1;         // render_utils_kernel.cu:260
    return P.stepdist * (float)k;                                                      // render_utils_kernel.cu:184
}

// The 4 waves of a workgroup share ONE bundle of 64 rays, wave w taking depth quarter w of every ray (a wave's footprint between
// two neighbouring rays is 4 rows x 256 B: the row reuse of adjacent rays hits L1/L2).  Records of quarter w go to quarter w of
// the bundle's workspace slice; after a workgroup barrier wave 0 runs the transmittance scan over the four runs of each ray in
// depth order and compacts the survivors.  (Since v4 there is no barrier: the last wave to finish its quarter does this.)
// COUNT: the sample counters of bench.py / the tests (k4_march_*_fwd `counters`) are a separate instantiation that visits EVERY
// sample (no skipping), so that the counters are the algorithm's sample counts and the render path carries no
// counting code.
// K4_GEOM_TIMING (profiling builds only, tools/geom_timing.py): s_memtime stamps at the stage boundaries of the geometry kernel, summed per wave and
// added to out_counters[8..15] by the COUNT-free instantiation when a counter buffer of >= 24 words is passed.  Not compiled into the product library.
#ifdef K4_GEOM_TIMING
#define K4_GSTAMP(SLOT) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                             gacc[SLOT] += now_ - glast; glast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define K4_GSTAMP(SLOT) do { } while (0)
#endif
template <int MODE, bool COUNT, int MINW>
__global__ __launch_bounds__(256, MINW) void k4_geom3_kernel(const MarchParams P) {
#ifdef K4_GEOM_TIMING
    unsigned long long gacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, glast = __builtin_amdgcn_s_memtime();
#endif
    __shared__ Geom2Lds lds_all[4];
    __shared__ int na_sh[4];
    __shared__ int arrived;                                            // waves of this workgroup that have finished their depth quarter
    __shared__ float tktab[MODE == MODE_MPI ? K4_TKTAB : 1];
    const int lane = k4_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Geom2Lds& L = lds_all[wv];
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.qhead = 0;            // work queue of the shading kernel that follows
    if (threadIdx.x == 0) arrived = 0;
    if (MODE == MODE_MPI) {
        for (int i = (int)threadIdx.x; i < K4_TKTAB; i += 256) tktab[i] = (float)i / P.nsm1;
    }
    __syncthreads();
    const bool unit_interval = P.interval == 1.f;
    const bool use_thres = P.thres > 0.f;
    unsigned long long n_inb = 0, n_mask = 0, n_alpha = 0, n_shade = 0, n_behind = 0;

    // static, XCD-banded bundle map, one bundle per workgroup: per-XCD or global work queues measured 6-25 % slower -- the
    // hardware's in-order dispatch already keeps neighbouring tiles on one XCD's L2.  (Heaviest-first by the previous frame's record
    // counts, as the shading kernel orders its queue, measured 4 % slower: 1.263 vs 1.216 ms per call -- neighbouring bundles no
    // longer share cache lines.  One ray table per workgroup instead of one per wave, 15.4 KB of LDS -> 10 resident workgroups
    // instead of 7: no gain, 1.23 vs 1.19-1.22 ms -- the 5 waves per SIMD of the register allocation are the cap.)
    for (bool once = true; once; once = false) {
    const int bid = k4_xcd_remap_banded((int)blockIdx.x, (int)gridDim.x, P.band_blocks);
    if (bid >= P.n_bundles) break;
    const Bundle B = bundle_from_id(P, bid);
    uint2* const ent_base = P.entries + (size_t)B.id * (size_t)P.ent_stride;
    // the four runs of records of this launch (one per wave, depth-ascending): quarters of the bundle's slice, or -- split form -- runs of
    // span0 (first launch) / span1 (second launch) samples per ray laid out one behind the other: 4 (span0 + span1) <= max_steps rounded up
    const int slab = (MODE == MODE_MPI && !COUNT) ? P.slab : 0;
    const int span0 = P.split_k >> 2, span1 = (((P.n_samples - P.split_k) + 63) >> 6) << 4;      // samples per ray and wave of the two launches
    const int quarter = slab == 0 ? (P.ent_stride >> 2) : (slab == 1 ? span0 : span1) * 64;
    const int run0 = slab == 2 ? 4 * span0 * 64 : 0;                              // records in front of this launch's first run
    uint2* const ent = ent_base + run0 + (size_t)wv * quarter;                   // this wave's run of records

    // ---- ray setup, lane = ray of the bundle ----
    const int myray = ray_index(P, B, lane);
    int ngrp;                                                                    // 16-sample groups of this wave's depth range of ray `lane`
    {
        const int rs = myray < 0 ? 0 : myray;
        float sx, sy, sz, dx, dy, dz;
        int nsteps;
        ray_setup<MODE>(P, P.rays_o[rs * 3 + 0], P.rays_o[rs * 3 + 1], P.rays_o[rs * 3 + 2],
                        P.rays_d[rs * 3 + 0], P.rays_d[rs * 3 + 1], P.rays_d[rs * 3 + 2], sx, sy, sz, dx, dy, dz, nsteps);
        if (myray < 0) nsteps = 0;
        int kq0, kq1;
        if (slab == 0) {
            const int nb4 = (((nsteps + 63) >> 6) + 3) >> 2;                     // 64-sample blocks per depth quarter of this ray
            kq0 = wv * nb4 * 64; kq1 = min((wv + 1) * nb4 * 64, nsteps);
        } else if (slab == 1) {
            kq0 = wv * span0; kq1 = min(kq0 + span0, nsteps);
        } else {
            // second launch: a ray whose transmittance fell below 1e-3 in the first launch takes no further sample (Alphas2Weights stops
            // there, render_utils_kernel.cu:597-600): nothing of it is probed, looked up or recorded
            const bool alive = myray >= 0 && !(P.out_ainv[rs] < 1e-3f);
            kq0 = P.split_k + wv * span1; kq1 = alive ? min(kq0 + span1, nsteps) : kq0;
        }
        ngrp = kq1 > kq0 ? (kq1 - kq0 + K4_GRP - 1) / K4_GRP : 0;
        *reinterpret_cast<float4*>(&L.raytab[lane][0]) = make_float4(sx, sy, sz, __int_as_float(kq0));
        *reinterpret_cast<float4*>(&L.raytab[lane][4]) = make_float4(dx, dy, dz, __int_as_float(kq1));
        L.acnt[lane] = 0;
    }
    K4_GSTAMP(0);                                                                // workgroup prologue + ray setup
    if (P.debug & 4096) ngrp = 0;                                                // ablation (WRONG results): prologue, arrival and the empty scan only -- the kernel's fixed cost
    if (P.debug & 8192) { if (lane == 0 && wv == 0) P.counts[B.id] = 0; break; } // ablation: not even the arrival / scan
    int gq = ngrp;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gq = max(gq, __shfl_xor(gq, off));
    gq = __builtin_amdgcn_readfirstlane(gq);
    const bool big = gq > 64;                                                    // more groups than mask bits: enumerate every group (g < 1024: host check)

    // ---- stage P: which 16-sample groups of my ray can touch an occupied voxel? ----
    unsigned long long keep = ngrp >= 64 ? ~0ull : ((1ull << ngrp) - 1ull);
    if (!COUNT && P.occ != nullptr && !big) {
        keep = 0ull;
        const float4 ra = *reinterpret_cast<const float4*>(&L.raytab[lane][0]);
        const float4 rb = *reinterpret_cast<const float4*>(&L.raytab[lane][4]);
        const int kq0 = __float_as_int(ra.w), kq1 = __float_as_int(rb.w);
        const int ncy = (P.MY + K4_OCC_CELL - 1) >> K4_OCC_SHIFT, zw = (P.MZ + 31) >> 5;
        const int ntab = ((P.MX + K4_OCC_CELL - 1) >> K4_OCC_SHIFT) * ncy * zw;                // entries per table
#pragma unroll 4
        for (int j = 0; j < gq; ++j) {                                            // (unrolled: the summary fetches of 4 groups fly together)
            const int ka = kq0 + K4_GRP * j, kb = min(ka + K4_GRP - 1, kq1 - 1);
            const float ta = tk_of<MODE>(P, tktab, j < ngrp ? ka : 0), tb = tk_of<MODE>(P, tktab, j < ngrp ? kb : 0);
            // the two end samples, with the arithmetic of stage A (same fmaf chain, same round())
            const int iax = k4_round_half_away(fmaf(fmaf(rb.x, ta, ra.x), P.msx, P.mtx)), ibx = k4_round_half_away(fmaf(fmaf(rb.x, tb, ra.x), P.msx, P.mtx));
            const int iay = k4_round_half_away(fmaf(fmaf(rb.y, ta, ra.y), P.msy, P.mty)), iby = k4_round_half_away(fmaf(fmaf(rb.y, tb, ra.y), P.msy, P.mty));
            const int iaz = k4_round_half_away(fmaf(fmaf(rb.z, ta, ra.z), P.msz, P.mtz)), ibz = k4_round_half_away(fmaf(fmaf(rb.z, tb, ra.z), P.msz, P.mtz));
            const int x0 = max(min(iax, ibx), 0), x1 = min(max(iax, ibx), P.MX - 1);
            const int y0 = max(min(iay, iby), 0), y1 = min(max(iay, iby), P.MY - 1);
            const int z0 = max(min(iaz, ibz), 0), z1 = min(max(iaz, ibz), P.MZ - 1);
            const bool empty = (x0 > x1) | (y0 > y1) | (z0 > z1);                // every sample's index is out of range on some axis
            const int cx0 = x0 >> K4_OCC_SHIFT, cy0 = y0 >> K4_OCC_SHIFT;
            const int sx = (x1 >> K4_OCC_SHIFT) - cx0, sy = (y1 >> K4_OCC_SHIFT) - cy0;
            const bool wide = (sx > 1) | (sy > 1) | (z1 - z0 >= 32);              // box beyond 2x2 cells / one 64-bit window: keep unseen
            // ONE 8-byte fetch: the window of table (x span, y span) at the box's first cell and z word (k4nerf.h)
            unsigned long long win = 0ull;
            if (!empty) {
                const unsigned tab = (unsigned)((sx > 0) + 2 * (sy > 0));
                win = P.occ[(size_t)tab * (size_t)ntab + (size_t)(cx0 * ncy + cy0) * (size_t)zw + (size_t)(z0 >> 5)];
            }
            const int len = z1 - z0 + 1;
            const unsigned long long bits = (len >= 64 ? ~0ull : ((1ull << (len & 63)) - 1ull)) << (z0 & 31);
            const bool hit = !empty && (wide || (win & bits) != 0ull);
            if (j < ngrp && hit) keep |= 1ull << j;
        }
    }
    K4_GSTAMP(1);                                                                // stage P: probe
    int qn = 0, qh = 0;          // ring fill / head (wave-uniform)
    int na = 0;                  // alpha-passing records written so far (wave-uniform)

    // ---- stage B: density + activation, lane = sample, up to 2 ring entries per lane.  Split in two halves so that the
    // 8 corner fetches of a batch are in flight while the wave does the occupancy stage of the NEXT 256 samples: the
    // batch is issued at the end of one group and finished at the end of the following one, so one memory round trip
    // (the occupancy bytes') covers both.  Per-stage ablation: the fetch wait was 0.52 ms of the kernel's 1.47 ms.
    int pend_n = 0;              // entries of the batch in flight (wave-uniform), 0 = none
    unsigned pkey[2];
    float pfx[2], pfy[2], pfz[2], pfa[2];
    float pdl[2][4], pdh[2][4];       // the 4 (x,y) rows' (zb, zb+1) density pairs
    float pa0[2], pa1[2];
    auto issue_b = [&](int nproc) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) if (s2 * 64 < nproc) {
            const bool lact = s2 * 64 + lane < nproc;
            const unsigned key = L.qk[(qh + s2 * 64 + (lact ? lane : 0)) & (K4_RING - 1)];
            const int rl = (int)(key >> 24);
            const int k = (int)(key & 0xffffffu);
            const float* rt = L.raytab[rl];
            const float tk = tk_of<MODE>(P, tktab, k);
            const float px = fmaf(rt[4], tk, rt[0]), py = fmaf(rt[5], tk, rt[1]), pz = fmaf(rt[6], tk, rt[2]);
            const float nx = k4_norm_coord_r(px, P.minx, P.lenx, P.rlenx);
            const float ny = k4_norm_coord_r(py, P.miny, P.leny, P.rleny);
            const float nz = k4_norm_coord_r(pz, P.minz, P.lenz, P.rlenz);
            const float ux = k4_unnorm(nx, P.X), uy = k4_unnorm(ny, P.Y), uz = k4_unnorm(nz, P.Z);
            const float fx = floorf(ux), fy = floorf(uy), fz = floorf(uz);
            const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
            // inside the closed bbox 0 <= u <= dim-1: only the "+1" corner can leave the grid and then its weight is exactly 0.
            // The two z neighbours are adjacent dwords: one 64-bit fetch at zb = min(z0, Z-2) (for z0 == Z-1 the pair is
            // shifted down by one and its upper element is the z0 corner; the z0+1 corner then has weight exactly 0)
            const int x1 = min(x0 + 1, P.X - 1), y1 = min(y0 + 1, P.Y - 1);
            const int zb = min(z0, max(P.Z - 2, 0));
            const unsigned r00 = (unsigned)(x0 * P.Y + y0) * (unsigned)P.Z, r01 = (unsigned)(x0 * P.Y + y1) * (unsigned)P.Z;
            const unsigned r10 = (unsigned)(x1 * P.Y + y0) * (unsigned)P.Z, r11 = (unsigned)(x1 * P.Y + y1) * (unsigned)P.Z;
            pkey[s2] = key; pfx[s2] = ux; pfy[s2] = uy; pfz[s2] = uz;
            if (P.debug & 16) {                                           // ablation: no density fetch
                const float c = __uint_as_float(0x3f000000u + (r00 & 0xffffu));
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) pdl[s2][c4] = pdh[s2][c4] = c;
            } else if (P.Z >= 2) {
                const float2 v0 = k4_ld2(P.density + r00 + zb), v1 = k4_ld2(P.density + r01 + zb);
                const float2 v2 = k4_ld2(P.density + r10 + zb), v3 = k4_ld2(P.density + r11 + zb);
                pdl[s2][0] = v0.x; pdh[s2][0] = v0.y; pdl[s2][1] = v1.x; pdh[s2][1] = v1.y;
                pdl[s2][2] = v2.x; pdh[s2][2] = v2.y; pdl[s2][3] = v3.x; pdh[s2][3] = v3.y;
            } else {                                                      // single-plane grid: both z corners are plane 0
                pdl[s2][0] = pdh[s2][0] = P.density[r00]; pdl[s2][1] = pdh[s2][1] = P.density[r01];
                pdl[s2][2] = pdh[s2][2] = P.density[r10]; pdl[s2][3] = pdh[s2][3] = P.density[r11];
            }
            if (MODE == MODE_MPI) {
                // act_shift grid [1,1,D]: x/y sizes are 1 -> only z interpolates (lib/dmpigo.py:48-58,316)
                const float ua = k4_unnorm(nz, P.act_d);
                const int a0 = (int)floorf(ua);
                pfa[s2] = ua;
                pa0[s2] = P.act_shift[a0]; pa1[s2] = P.act_shift[min(a0 + 1, P.act_d - 1)];
            }
        }
        pend_n = nproc;
        qh = (qh + nproc) & (K4_RING - 1);
        qn -= nproc;
    };
    auto finish_b = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) if (s2 * 64 < pend_n) {
            const bool lact = s2 * 64 + lane < pend_n;
            const unsigned key = pkey[s2];
            const float ux = pfx[s2], uy = pfy[s2], uz = pfz[s2];
            const K4Tri t = k4_tri_setup(ux, uy, uz);
            const bool top = t.z0 > max(P.Z - 2, 0);                      // pair shifted down: z0 corner = upper element
            // explicit FMA chain (grid_sampler accumulates corner by corner, contracted by nvcc): written out so that every
            // inlined copy of this stage rounds identically -- a ray must not depend on which slot its sample landed in
            float sigma = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float lo = pdl[s2][c], hi = pdh[s2][c];
                const float dlo = top ? hi : lo;
                sigma = fmaf(dlo, t.w[2 * c], sigma); sigma = fmaf(hi, t.w[2 * c + 1], sigma);
            }
            if (MODE == MODE_MPI) {
                const float ua = pfa[s2];
                const float fa = floorf(ua);
                sigma += fmaf(pa1[s2], ua - fa, pa0[s2] * ((fa + 1.f) - ua));
            }
            // raw2alpha: e = exp(d+shift); alpha = 1 - (1+e)^(-interval)   render_utils_kernel.cu:439-441
            const float e = expf(sigma + P.shift);
            float alpha;
            if (unit_interval) alpha = 1.f - 1.f / (1.f + e);
            else alpha = 1.f - powf(1.f + e, -P.interval);
            const bool act = lact && (use_thres ? (alpha > P.thres) : true);
            const uint64_t bm = __ballot(act);
            if (act) {
                ent[na + k4_prefix(bm)] = make_uint2(key, __float_as_uint(alpha));
                atomicAdd(&L.acnt[(int)(key >> 24)], 1);
            }
            na += __popcll(bm);
        }
        pend_n = 0;
    };
    auto pump_b = [&]() {        // end of a group: retire the batch in flight, start the next one
        if (pend_n) finish_b();
        while (qn >= 64) {
            issue_b(qn >= 128 ? 128 : 64);
            if (qn >= 64) finish_b(); else break;
        }
    };

    // ---- stage A: occupancy.  Work entries = kept (ray, group) pairs in ray-major, depth-ascending order, written to an LDS list
    // by their rays' lanes (prefix sum of the per-ray counts); an item = 4 consecutive entries, lanes 16q..16q+15 = the 16
    // consecutive samples of entry q.  Rays are taken in sub-batches of `rb` so that rb x groups-per-ray fits the list (one
    // batch of 64 rays for the LLFF configuration: 4 groups per ray and quarter).
    const int q16 = lane >> 4, l15 = lane & 15;
    int rb = 64;
    while (rb > 1 && rb * gq > K4_ELIST) rb >>= 1;
    for (int r0 = 0; r0 < 64; r0 += rb) {
        int total;
        {
            const bool mine = lane >= r0 && lane < r0 + rb;
            int c = mine ? (big ? ngrp : __popcll(keep)) : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            total = __builtin_amdgcn_readlane(incl, 63);
            int o = incl - c;
            if (big) { for (int g = 0; g < c; ++g) L.elist[o + g] = (unsigned short)((lane << 10) | g); }
            else {
                unsigned long long m = mine ? keep : 0ull;
                for (int j = 0; j < gq; ++j)                                  // wave-uniform trip count, <= 64
                    if ((m >> j) & 1ull) L.elist[o++] = (unsigned short)((lane << 10) | j);
            }
        }
        K4_GSTAMP(2);                                                            // entry list
        // groups of 4 items: the 4 occupancy bytes are fetched together (one memory round trip per 256 samples), branch-free
        // (clamped index, validity folded into the predicate); all 4 are consumed (ballots) before any density batch is
        // issued, so the only fetches in flight across the next group's wait are that batch's
        for (int e0 = 0; e0 < total; e0 += 16) {
            unsigned mbyte[4];
            bool inbv[4];
            unsigned ikey[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ei = e0 + 4 * j + q16;
                const bool ev = ei < total;
                const unsigned e = ev ? (unsigned)L.elist[ei] : 0u;
                const int r_ = (int)(e >> 10), g_ = (int)(e & 1023u);
                const float4 ra = *reinterpret_cast<const float4*>(&L.raytab[r_][0]);
                const float4 rbv = *reinterpret_cast<const float4*>(&L.raytab[r_][4]);
                const int k = __float_as_int(ra.w) + g_ * K4_GRP + l15;
                const float tk = tk_of<MODE>(P, tktab, k);
                const float px = fmaf(rbv.x, tk, ra.x), py = fmaf(rbv.y, tk, ra.y), pz = fmaf(rbv.z, tk, ra.z);
                const bool inb = ev && (k < __float_as_int(rbv.w)) &&
                    !((P.minx > px) | (P.miny > py) | (P.minz > pz) | (P.maxx < px) | (P.maxy < py) | (P.maxz < pz));
                const int mi = k4_round_half_away(fmaf(px, P.msx, P.mtx));
                const int mj = k4_round_half_away(fmaf(py, P.msy, P.mty));
                const int mk = k4_round_half_away(fmaf(pz, P.msz, P.mtz));
                const bool ok = inb && (unsigned)mi < (unsigned)P.MX && (unsigned)mj < (unsigned)P.MY && (unsigned)mk < (unsigned)P.MZ;
                const unsigned midx = ok ? (unsigned)(mi * P.MY + mj) * (unsigned)P.MZ + (unsigned)mk : 0u;    // < 2^32 mask voxels
                mbyte[j] = ok ? (unsigned)P.mask[midx] : 0u;
                inbv[j] = inb;
                ikey[j] = ((unsigned)r_ << 24) | (unsigned)k;
            }
            K4_GSTAMP(3);                                                        // stage A: index arithmetic + issue of the 4 byte fetches
            uint64_t mball[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mball[j] = __ballot(mbyte[j] != 0);
                if (COUNT) { n_inb += __popcll(__ballot(inbv[j])); n_mask += __popcll(mball[j]); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (mbyte[j] != 0) L.qk[(qh + qn + k4_prefix(mball[j])) & (K4_RING - 1)] = ikey[j];
                qn += __popcll(mball[j]);
            }
            K4_GSTAMP(4);                                                        // stage A: wait for the bytes, ballots, ring
            pump_b();
            K4_GSTAMP(5);                                                        // stage B: retire the density batch in flight, issue the next
        }
    }
    if (pend_n) finish_b();
    while (qn > 0) { issue_b(min(qn, 128)); finish_b(); }
    K4_GSTAMP(6);                                                                // drain

    // No barrier: a wave whose quarter is done leaves at once (depth quarters are very unequal -- waves parked at a barrier held 1/3
    // of the wave slots); the LAST wave to arrive finishes the bundle.  It re-reads records the other waves stored: every wave drains
    // its stores (vmcnt) before it counts itself in, the records go through this CU's L1 write-through and were never read before.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    int prev = 0;
    if (lane == 0) { na_sh[wv] = na; prev = __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP); }
    prev = __builtin_amdgcn_readfirstlane(prev);
    if (prev != 3) break;                                              // not the last: done
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- stage C: transposed transmittance scan, lane = ray (render_utils_kernel.cu:591-603); a ray's records are
    // 4 runs (one per depth quarter), visited in depth order ----
    float T = 1.f;
    bool stopped = false;
    if (slab == 2) { T = myray >= 0 ? P.out_ainv[myray] : 1.f; stopped = T < 1e-3f; }      // where the first launch's scan left this ray
    const bool any_rec = (na_sh[0] | na_sh[1] | na_sh[2] | na_sh[3]) != 0;                  // (wave-uniform) a bundle without records: nothing to scan or compact
#pragma unroll
    for (int w = 0; w < 4 && any_rec; ++w) {
        uint2* const run = ent_base + run0 + (size_t)w * quarter;
        const int c = lds_all[w].acnt[lane];
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int seg = incl - c;
        int maxc = c;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off));
        maxc = __builtin_amdgcn_readfirstlane(maxc);
        if (P.debug & 32) maxc = 0;                                    // ablation: no transmittance scan
        // (the NEXT four alphas are requested before the current four are folded into T: the loop is a load -> dependent chain -> store
        //  sequence per iteration, 25 % of the kernel's wave time in round 5's form)
        float an[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) an[u] = (u < c && maxc > 0) ? __uint_as_float(run[seg + u].y) : 0.f;
        for (int j0 = 0; j0 < maxc; j0 += 4) {
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = an[u];
            if (j0 + 4 < maxc) {
#pragma unroll
                for (int u = 0; u < 4; ++u) an[u] = (j0 + 4 + u < c) ? __uint_as_float(run[seg + j0 + 4 + u].y) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (COUNT) n_behind += __popcll(__ballot(j0 + u < c && stopped));      // density was evaluated for a sample the scan then drops
                if (j0 + u < c) {
                    float wgt = -1.f;                                  // behind the early stop: never shaded
                    if (!stopped) {
                        wgt = T * a[u];
                        T = fmaf(-T, a[u], T);                         // == (float)((double)T*(1.-a)): the CUDA source's fp64 product, one rounding
                        if (T < 1e-3f) stopped = true;                 // the crossing sample is still counted (:597-600)
                    }
                    run[seg + j0 + u].y = __float_as_uint(wgt);
                }
            }
        }
    }
    const float my_ainv = T;                                           // alphainv_last of ray `lane` (1 if it has no samples)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- stage D: keep w > thres, compacted to the front of the bundle's slice, run order preserved ----
    int cnt = slab == 2 ? __builtin_amdgcn_readfirstlane(P.counts[B.id]) : 0, na_all = 0;      // second launch: append behind the first launch's survivors
#pragma unroll
    for (int w = 0; w < 4 && any_rec; ++w) {
        const uint2* const run = ent_base + run0 + (size_t)w * quarter;
        int naw = na_sh[w];
        na_all += naw;
        if (P.debug & 128) naw = 0;                                    // ablation: no survivor compaction (nothing shaded)
        for (int base = 0; base < naw; base += 64) {
            const int i = base + lane;
            const bool v = i < naw;
            const uint2 e = run[v ? i : 0];
            const float wgt = __uint_as_float(e.y);
            const bool shade = v && (use_thres ? (wgt > P.thres) : (wgt >= 0.f));
            const uint64_t sm = __ballot(shade);
            if (shade) ent_base[cnt + k4_prefix(sm)] = e;               // cnt + prefix <= position of e: never overtakes the reads
            cnt += __popcll(sm);
        }
    }
    if (lane == 0) P.counts[B.id] = cnt;
    if (myray >= 0) {
        P.out_ainv[myray] = my_ainv;
        // a bundle without a shaded sample never enters the shading queue (k4_order_kernel): its rays' outputs are final here --
        // rgb_marched = alphainv_last * bg (lib/dmpigo.py:397), depth 0.  (First launch of the split form: the second one decides.)
        if (cnt == 0 && slab != 1) {
            const float ab = my_ainv * P.bg;
            P.out_rgb[(size_t)myray * 3 + 0] = ab; P.out_rgb[(size_t)myray * 3 + 1] = ab; P.out_rgb[(size_t)myray * 3 + 2] = ab;
            P.out_depth[myray] = 0.f;
        }
    }
    n_alpha += (unsigned long long)na_all; n_shade += (unsigned long long)cnt;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }   // bundle
#ifdef K4_GEOM_TIMING
    K4_GSTAMP(7);                                                                // arrival, stages C / D (last wave only)
    if (P.timing && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&P.timing[i], gacc[i]);
        atomicAdd(&P.timing[8], 1ull);
    }
#endif
    if (COUNT && P.counters && lane == 0) {
        atomicAdd(&P.counters[0], n_inb); atomicAdd(&P.counters[1], n_mask);
        atomicAdd(&P.counters[2], n_alpha); atomicAdd(&P.counters[3], n_shade); atomicAdd(&P.counters[4], n_behind);
    }
}

// =====================================================================================================
// K2: shading
// =====================================================================================================
#include "k4_march_mlp.h"      // the rgbnet on the matrix cores (operand layouts, splits, mlp_mfma / mlp_mfma_bx / mlp_pair64), k4_corner_setup

// ------------------------------------------------------------------------------------------------------------------
// Between K1 and K2: the order in which the shading kernel's persistent waves take the bundles.  A bundle is shaded by ONE
// wave, 64 records at a time, and bundles carry 0 .. thousands of records: taken in image order, a heavy bundle pulled near
// the end of the queue finished hundreds of microseconds after every other wave had run dry (SQ_WAVE_CYCLES: the average
// wave was alive for 67 % of the kernel).  Longest-processing-time-first: a counting sort of the bundle ids by their number
// of 64-record batches, descending -- the tail of the queue is then made of the lightest bundles.  One workgroup; the
// shading results do not depend on the order (exact per-ray sums, independent bundles).
// ------------------------------------------------------------------------------------------------------------------
#define K4_ORDER_CLASSES 256     // bundles of >= 255 batches share the first class
#define K4_ORDER_SUB 16          // sub-bins per class (bundle id & 15): 16x fewer same-address LDS atomics -- most bundles fall in a few classes
#define K4_QLEN 32               // qhead[K4_QLEN] = queue length: its own 128-byte line (the head word's line is busy with the queue's atomics)
__device__ __forceinline__ int k4_batches_of(int count) { return (int)(((unsigned)max(count, 0) + 63u) >> 6); }
__global__ __launch_bounds__(1024) void k4_order_kernel(const int* __restrict__ counts, int2* __restrict__ jobs, int n_bundles, int* qhead) {
    constexpr int NBIN = K4_ORDER_CLASSES * K4_ORDER_SUB;            // 4096 = 4 per thread
    __shared__ int hist[NBIN];
    __shared__ int wsum[16];
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) hist[tid * 4 + i] = 0;
    if (tid == 0) qhead[0] = 0;
    __syncthreads();
    // class 0 = the most batches; any int is a valid key (the workspace may hold anything before its first use)
    // (16 counts per thread are fetched together: one memory round trip per 16384 bundles instead of one per 1024)
    for (int base = 0; base < n_bundles; base += 1024 * 16) {
        int v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int b = base + i * 1024 + tid; v[i] = counts[b < n_bundles ? b : 0]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = base + i * 1024 + tid;
            const int nbat = k4_batches_of(v[i]);
            if (b < n_bundles && nbat > 0) atomicAdd(&hist[(K4_ORDER_CLASSES - 1 - min(nbat, K4_ORDER_CLASSES - 1)) * K4_ORDER_SUB + (b & (K4_ORDER_SUB - 1))], 1);
        }
    }
    __syncthreads();
    // exclusive prefix over the bins: 4 per thread, wave scans, scan of the 16 wave totals
    const int lane = tid & 63, wv = tid >> 6;
    int c[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = hist[tid * 4 + i]; mine += c[i]; }
    int inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int run = inc - mine;
    for (int w = 0; w < wv; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) { hist[tid * 4 + i] = run; run += c[i]; }
    if (tid == 1023) qhead[K4_QLEN] = run;                             // queue length
    __syncthreads();
    for (int base = 0; base < n_bundles; base += 1024 * 16) {
        int v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int b = base + i * 1024 + tid; v[i] = counts[b < n_bundles ? b : 0]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = base + i * 1024 + tid;
            const int nbat = k4_batches_of(v[i]);
            if (b < n_bundles && nbat > 0) jobs[atomicAdd(&hist[(K4_ORDER_CLASSES - 1 - min(nbat, K4_ORDER_CLASSES - 1)) * K4_ORDER_SUB + (b & (K4_ORDER_SUB - 1))], 1)] = make_int2(b, v[i]);   // the count rides along: one dependent fetch less per job
        }
    }
}

// A shading job: every record of one bundle, taken from the queue k4_order_kernel wrote.
struct ShadeJob { Bundle B; int total; };
__device__ __forceinline__ bool k4_next_job(const MarchParams& P, int lane, ShadeJob& J) {
    int bid = 0;
    if (lane == 0) bid = atomicAdd(P.qhead, 1);
    bid = __builtin_amdgcn_readfirstlane(bid);
    if (bid >= __builtin_amdgcn_readfirstlane(P.qhead[K4_QLEN])) return false;
    const int2 jb = P.jobs[bid];
    const int bundle = __builtin_amdgcn_readfirstlane(jb.x);
    J.B = bundle_from_id(P, bundle);
    J.total = __builtin_amdgcn_readfirstlane(jb.y);
    return true;
}
// ---- per-ray sums (segment_coo, lib/dmpigo.py:382-386,418-424) in FIXED POINT (round 6; rounds 2-5: fp64 with terms rounded to 2^-40) ----
// A record's four terms -- w rgb (each <= 1, and sum_w <= 1 per ray) and w s -- are rounded to multiples of 2^-30 (depth: of 1 / depth_fx,
// depth_fx = 2^30 / the largest s a record can carry rounded up to a power of two) and added as INTEGERS with two ds_add_u64 per record:
// word 0 = red | green << 32, word 1 = blue | depth << 32; a field stays below 2^31, so nothing carries between fields.  Integer sums are
// exact and commute: a ray's sums do not depend on how its records fall into batches or on the order the LDS unit serialises same-address
// lanes in -- whole frame, tile window and row band give bit-identical pixels, as with the fp64 form -- at 8 fp32-rate vector
// instructions + 2 LDS atomics per record instead of 12 fp64-rate + 4.  The 2^-31 rounding per term (a few hundred terms per ray) is the
// size of one fp32 rounding of the result.  A NaN term (NaN weights / features) cannot be carried by an integer: it sets bit 31 of its
// field, which no finite sum reaches, and the ray's output is NaN as in the reference.
__device__ __forceinline__ float k4_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
#define K4_FX_ONE 1073741824.f                     // 2^30
__device__ __forceinline__ unsigned k4_fx(float x, float scale) { return (unsigned)fmaf(x, scale, 0.5f); }      // x >= 0; NaN -> 0
__device__ __forceinline__ float k4_unfx(unsigned f, float inv) { return (f & 0x80000000u) ? __uint_as_float(0x7fc00000u) : (float)f * inv; }
// Per-ray outputs of a finished bundle: rgb_marched = sum + alphainv_last*bg (lib/dmpigo.py:397), depth.
__device__ __forceinline__ void k4_finish_bundle(const MarchParams& P, const ShadeJob& J, const unsigned long long* acc, int lane) {
    const unsigned long long a0 = acc[lane * 2 + 0], a1 = acc[lane * 2 + 1];
    const int ray = ray_index(P, J.B, lane);
    if (ray >= 0) {
        const float ab = P.out_ainv[ray] * P.bg;
        P.out_rgb[(size_t)ray * 3 + 0] = k4_unfx((unsigned)a0, 1.f / K4_FX_ONE) + ab;
        P.out_rgb[(size_t)ray * 3 + 1] = k4_unfx((unsigned)(a0 >> 32), 1.f / K4_FX_ONE) + ab;
        P.out_rgb[(size_t)ray * 3 + 2] = k4_unfx((unsigned)a1, 1.f / K4_FX_ONE) + ab;
        P.out_depth[ray] = k4_unfx((unsigned)(a1 >> 32), P.depth_fx_inv);
    }
}

// ARITH: 0 = fp32-input MFMA (mlp_mfma), 3 = exact 3-term bf16 splits, 2 = layer 2 on 2-term splits (the default; see k4_split2).
// WG = workgroups per CU the register allocation is bounded for (width 128 holds ~150 KB of LDS: one).
// FAST: the LLFF / plain-DVGO input shape fixed at compile time -- voxel-major k0 with 12 padded channels (9 used for MPI, 12 for DVGO),
// no positional-encoding frequencies, no k0_skip, 15 inputs + the bias input = ONE 16-wide k block, split-bf16 arithmetic: no per-channel
// branches in the batch loop, features through registers (RegFeat), MPI step positions from a table.  Same expression trees as the general
// path -> the same bits (tests/test_march_gpu.py::test_fast_shading_path_bit_identical).
template <int MODE, int WIDTH, int NHID, int ARITH, int WG = K4_SHADE_WG_PER_CU, bool FAST = false>
__global__ __launch_bounds__(256, WG) void k4_shade_kernel(const MarchParams P) {
    constexpr bool B3 = ARITH != 0;
    static_assert(!FAST || (B3 && WIDTH > 0 && WIDTH <= 64), "FAST needs the split-bf16 arithmetic");
    __shared__ float tktab[FAST && MODE == MODE_MPI ? K4_TKTAB : 1];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int W = WIDTH > 0 ? WIDTH : 32;
    constexpr int NB = W / 32;
    typedef MlpLayout<W, NHID> ML;
    const int lane = k4_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS carve: [mlp weights][per wave: acc 64 x 2 u64 | feat K1P x 64]
    // packed buffer = [fp32 section][b3 section][b2 section]; only the section of this arithmetic is staged (each carries its own fp32 tail)
    const int mlp_floats = WIDTH > 0 ? (ARITH == 3 ? P.mlp_floats_b3 : ARITH == 2 ? P.mlp_floats_b2 : P.mlp_floats) : 0;
    const int mlp_pad = (mlp_floats + 3) & ~3;
    float* const wl = smem;
    const int per_wave = 64 * 4 + (WIDTH > 0 && !FAST ? P.k1p * 64 : 0);
    unsigned long long* const acc = reinterpret_cast<unsigned long long*>(smem + mlp_pad + wv * per_wave);     // [64][2]  red | green, blue | depth (fixed point)
    float* const feat = reinterpret_cast<float*>(acc + 64 * 2);                        // [K1P][64]
    if (WIDTH > 0) {
        const float* const src = P.mlp + (ARITH == 3 ? P.mlp_floats : ARITH == 2 ? P.mlp_floats + P.mlp_floats_b3 : 0);
        for (int i = threadIdx.x; i < mlp_floats; i += 256) wl[i] = src[i];
    }
    if (FAST && MODE == MODE_MPI) {
        for (int i = (int)threadIdx.x; i < K4_TKTAB; i += 256) tktab[i] = (float)i / P.nsm1;      // == step_t<MPI>: the same IEEE division, once per workgroup
    }
    __syncthreads();
    const int half = lane >> 5;
    __amdgpu_buffer_rsrc_t k0rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.k0), 0, FAST ? (int)((unsigned)P.X * (unsigned)P.Y * (unsigned)P.Z * 48u) : 0, 0x00020000);
    (void)k0rs;
#ifdef K4_SHADE_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
#endif

    // persistent waves pull bundles from a global queue (bundles carry 0..thousands of records: static
    // assignment left SIMDs idle behind the slowest tile); a bundle is still shaded by ONE wave, in order.
    for (;;) {
    ShadeJob J;
    if (!k4_next_job(P, lane, J)) break;
    const Bundle B = J.B;
    acc[lane * 2 + 0] = 0ull; acc[lane * 2 + 1] = 0ull;
    const uint2* __restrict__ ent = P.entries + (size_t)B.id * (size_t)P.ent_stride;
    const int total = J.total;
    // the bundle's 64 rays live in registers, lane = ray slot; a record reads its ray's with ds_bpermute (__shfl).
    // One fetch round per bundle instead of a dependent one per 64-record batch (the batch loop was 59 % s_waitcnt:
    // record -> ray -> 3 rounds of corner fetches were five serial round trips; now the corner fetches are the only one)
    float my_sx, my_sy, my_sz, my_dx, my_dy, my_dz, my_vx, my_vy, my_vz;
    {
        const int mray = ray_index(P, B, lane);
        const int ms = mray < 0 ? 0 : mray;
        int nsteps_unused;
        ray_setup<MODE>(P, P.rays_o[ms * 3 + 0], P.rays_o[ms * 3 + 1], P.rays_o[ms * 3 + 2],
                        P.rays_d[ms * 3 + 0], P.rays_d[ms * 3 + 1], P.rays_d[ms * 3 + 2], my_sx, my_sy, my_sz, my_dx, my_dy, my_dz, nsteps_unused);
        my_vx = P.viewdirs[ms * 3 + 0]; my_vy = P.viewdirs[ms * 3 + 1]; my_vz = P.viewdirs[ms * 3 + 2];
    }
    uint2 en_next = ent[lane < total ? lane : 0];

    for (int base = 0; base < total; base += 64) {
        K4_TSTAMP(7);                                    // between batches: queue ticket, ray setup, per-ray outputs
        const int nproc = (total - base) < 64 ? (total - base) : 64;
        const bool lact = lane < nproc;
        const uint2 en = en_next;
        en_next = ent[(base + 64 + lane < total) ? base + 64 + lane : 0];       // next batch's records fly during this batch
        const float w = lact ? __uint_as_float(en.y) : 0.f;
        const int rl = lact ? (int)(en.x >> 24) : 0;
        const int k = (int)(en.x & 0xffffffu);
        const float sx = __shfl(my_sx, rl), sy = __shfl(my_sy, rl), sz = __shfl(my_sz, rl);
        const float dx = __shfl(my_dx, rl), dy = __shfl(my_dy, rl), dz = __shfl(my_dz, rl);
        const float tk = FAST ? tk_of<MODE>(P, tktab, k) : step_t<MODE>(P, k);
        const float px = fmaf(dx, tk, sx), py = fmaf(dy, tk, sy), pz = fmaf(dz, tk, sz);
        const float nx = k4_norm_coord_r(px, P.minx, P.lenx, P.rlenx);
        const float ny = k4_norm_coord_r(py, P.miny, P.leny, P.rleny);
        const float nz = k4_norm_coord_r(pz, P.minz, P.lenz, P.rlenz);
        unsigned cidx[8];                                        // voxel index (< 2^31 voxels)
        float cw[8];
        k4_corner_setup(P, nx, ny, nz, cidx, cw);
        K4_TSTAMP(0);                                    // record unpack, sample point, corner indices / weights
        float o0, o1, o2;
        if constexpr (FAST) {
            // 8 corners x 48 B = 24 independent 16-byte fetches, one memory round trip; per channel the corners accumulate in corner order
            // (the general path's expression: v_pk_fma_f32 pairs)
            // (buffer loads: one 32-bit multiply per corner instead of 64-bit address arithmetic; FAST requires the repacked grid < 4 GB)
            float4 q[3][8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned off = cidx[c] * 48u;
#pragma unroll
                for (int g4 = 0; g4 < 3; ++g4) {
                    const k4_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(k0rs, (int)off, g4 * 16, 0);
                    q[g4][c] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
                }
            }
            float f[16];
#pragma unroll
            for (int g4 = 0; g4 < 3; ++g4) {
                k4_f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const k4_f32x2 ww = {cw[c], cw[c]};
                    const k4_f32x2 qa = {q[g4][c].x, q[g4][c].y}, qb = {q[g4][c].z, q[g4][c].w};
                    va = __builtin_elementwise_fma(qa, ww, va);
                    vb = __builtin_elementwise_fma(qb, ww, vb);
                }
                f[g4 * 4 + 0] = va.x; f[g4 * 4 + 1] = va.y; f[g4 * 4 + 2] = vb.x; f[g4 * 4 + 3] = vb.y;
            }
            K4_TSTAMP(1);                                // 24 corner fetches + interpolation
            // layer-1 input in the order of W1ext's columns.  MPI (C = 9): k0[0..9) | pe_spa = (nz, ny, nx) (lib/dmpigo.py:338) | viewdirs | 1;
            // DVGO (C = 12): k0[0..12) | viewdirs | 1  (lib/dvgo.py:387-392)
            if (MODE == MODE_MPI) { f[9] = nz; f[10] = ny; f[11] = nx; }
            f[12] = __shfl(my_vx, rl); f[13] = __shfl(my_vy, rl); f[14] = __shfl(my_vz, rl); f[15] = 1.f;
            RegFeat fs;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const k4_u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(f[e]), __float_as_uint(f[8 + e]), false, false);
                fs.x[e] = __uint_as_float(sw.x); fs.y[e] = __uint_as_float(sw.y);
            }
            K4_TSTAMP(2);                                // remaining features, half-wave exchange
            constexpr int NT2 = ARITH == 3 ? 3 : 2, NT1 = ARITH == 3 ? 3 : K4_B2_L1_TERMS;
            if constexpr (K4_MLP_PAIR && W == 64 && NHID == 1) {
                if (nproc > 32) mlp_pair64<NT1, NT2>(wl, fs, lane, half, o0, o1, o2 K4_TPASS);
                else mlp_mfma_bx<W, NHID, NT1, NT2, RegFeat, true>(wl, fs, 16, lane, half, P.debug, nproc, o0, o1, o2 K4_TPASS);
            } else mlp_mfma_bx<W, NHID, NT1, NT2>(wl, fs, 16, lane, half, P.debug, nproc, o0, o1, o2 K4_TPASS);
        } else if (WIDTH == 0) {
            // rgbnet is None: rgb = sigmoid(k0)   (lib/dvgo.py:377-379)
            float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float q = (P.k0_layout != K4_K0_CHANNEL_MAJOR)
                        ? P.k0[(size_t)cidx[c] * P.CP + ch]
                        : P.k0[(size_t)ch * P.X * P.Y * P.Z + cidx[c]];
                    v[ch] += q * cw[c];
                }
            o0 = v[0]; o1 = v[1]; o2 = v[2];
        } else {
            // ---------------- features -> LDS feat[k][sample] ----------------
            float dif0 = 0.f, dif1 = 0.f, dif2 = 0.f;                // k0[:, :3] when rgbnet_direct=False
            if (P.k0_layout != K4_K0_CHANNEL_MAJOR && P.CP == 12 && !(P.debug & 1)) {
                // (NOT k4_gather12 + a copy loop: that form -- all 12 channels blended first, then written to LDS -- compiled to a stream
                // whose results differed from run to run in ~80 of the LLFF frame's 762,048 rays on the MI355X, by up to 5e-3; this
                // form, stores interleaved with the channel groups, is bit-reproducible.  profiles/r05_marcher_split_path.md)
                // 12 padded channels (rgbnet_dim 9..12): the 8 corners x 48 B are 24 independent 16-byte fetches, issued
                // together (one memory round trip); accumulation order per channel = corner order, as below
                float4 q[3][8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4* const src = reinterpret_cast<const float4*>(P.k0 + (size_t)cidx[c] * 12);
                    q[0][c] = src[0]; q[1][c] = src[1]; q[2][c] = src[2];
                }
#pragma unroll
                for (int g4 = 0; g4 < 3; ++g4) {
                    k4_f32x2 va = {0.f, 0.f}, vb = {0.f, 0.f};                 // v_pk_fma_f32: two channels per instruction, corner order kept
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const k4_f32x2 ww = {cw[c], cw[c]};
                        const k4_f32x2 qa = {q[g4][c].x, q[g4][c].y}, qb = {q[g4][c].z, q[g4][c].w};
                        va = __builtin_elementwise_fma(qa, ww, va);
                        vb = __builtin_elementwise_fma(qb, ww, vb);
                    }
                    const float vv[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int ch = g4 * 4 + cc;
                        if (ch < P.k0_skip) { if (ch == 0) dif0 = vv[cc]; else if (ch == 1) dif1 = vv[cc]; else dif2 = vv[cc]; }
                        else if (ch < P.C) feat[(ch - P.k0_skip) * 64 + lane] = vv[cc];
                    }
                }
            } else if (P.k0_layout != K4_K0_CHANNEL_MAJOR) {
                for (int g = 0; g < P.CP; g += 4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float4 q = (P.debug & 1) ? make_float4(cw[c], 0.5f, 0.25f, 1.f) : *reinterpret_cast<const float4*>(P.k0 + (size_t)cidx[c] * P.CP + g);
                        v.x += q.x * cw[c]; v.y += q.y * cw[c]; v.z += q.z * cw[c]; v.w += q.w * cw[c];
                    }
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int ch = g + cc;
                        if (ch < P.k0_skip) { if (ch == 0) dif0 = vv[cc]; else if (ch == 1) dif1 = vv[cc]; else dif2 = vv[cc]; }
                        else if (ch < P.C) feat[(ch - P.k0_skip) * 64 + lane] = vv[cc];
                    }
                }
            } else {
                const size_t plane = (size_t)P.X * P.Y * P.Z;
                for (int ch = 0; ch < P.C; ++ch) {
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) v += P.k0[plane * ch + cidx[c]] * cw[c];
                    if (ch < P.k0_skip) { if (ch == 0) dif0 = v; else if (ch == 1) dif1 = v; else dif2 = v; }
                    else feat[(ch - P.k0_skip) * 64 + lane] = v;
                }
            }
            K4_TSTAMP(1);                                // 24 corner fetches + interpolation
            int fi = P.C - P.k0_skip;
            if (MODE == MODE_MPI) {
                // pe_spa = normalised position flipped to (z,y,x); [v, sin(v x freq), cos(...)]   lib/dmpigo.py:338,350-351
                const float pe[3] = {nz, ny, nx};
#pragma unroll
                for (int c = 0; c < 3; ++c) feat[(fi + c) * 64 + lane] = pe[c];
                fi += 3;
                for (int f = 0; f < P.spe; ++f) {
                    const float fr = (float)(1 << f);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        feat[(fi + c * P.spe + f) * 64 + lane] = sinf(pe[c] * fr);
                        feat[(fi + 3 * P.spe + c * P.spe + f) * 64 + lane] = cosf(pe[c] * fr);
                    }
                }
                fi += 6 * P.spe;
            }
            {
                // viewdirs_emb[ray_id]   lib/dmpigo.py:347-349, lib/dvgo.py:387-389
                const float vd[3] = {__shfl(my_vx, rl), __shfl(my_vy, rl), __shfl(my_vz, rl)};
#pragma unroll
                for (int c = 0; c < 3; ++c) feat[(fi + c) * 64 + lane] = vd[c];
                fi += 3;
                for (int f = 0; f < P.vpe; ++f) {
                    const float fr = (float)(1 << f);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        feat[(fi + c * P.vpe + f) * 64 + lane] = sinf(vd[c] * fr);
                        feat[(fi + 3 * P.vpe + c * P.vpe + f) * 64 + lane] = cosf(vd[c] * fr);
                    }
                }
                fi += 6 * P.vpe;
            }
            feat[fi * 64 + lane] = 1.f;                                   // bias input
            for (int kx = fi + 1; kx < P.k1p; ++kx) feat[kx * 64 + lane] = 0.f;
            __builtin_amdgcn_wave_barrier();

            K4_TSTAMP(2);                                // remaining features -> LDS
            float l0, l1, l2;
            if (B3) {
                LdsFeat fs = {feat, P.k1p, lane & 31, half};
                constexpr int NT2 = ARITH == 3 ? 3 : 2, NT1 = ARITH == 3 ? 3 : K4_B2_L1_TERMS;
                if constexpr (K4_MLP_PAIR && W == 64 && NHID == 1) {
                    // a full batch of the LLFF shape: both tiles through one software pipeline (same bits); anything else one tile at a time
                    if (P.k1p == 16 && nproc > 32 && !(P.debug & 2)) mlp_pair64<NT1, NT2>(wl, fs, lane, half, l0, l1, l2 K4_TPASS);
                    else mlp_mfma_bx<W, NHID, NT1, NT2>(wl, fs, P.k1p, lane, half, P.debug, nproc, l0, l1, l2 K4_TPASS);
                } else mlp_mfma_bx<W, NHID, NT1, NT2>(wl, fs, P.k1p, lane, half, P.debug, nproc, l0, l1, l2 K4_TPASS);
            }
            else mlp_mfma<W, NHID>(wl, feat, P.k1p, lane, half, P.debug, l0, l1, l2);
            o0 = l0 + dif0; o1 = l1 + dif1; o2 = l2 + dif2;            // rgb_logit + k0_diffuse (lib/dvgo.py:412)
            __builtin_amdgcn_wave_barrier();
        }
        // sigmoid, blend, per-ray sums in fixed point (see k4_fx above)
        // sigmoid = rcp(1 + exp2(-x log2 e)) on v_exp_f32 / v_rcp_f32 (1 ulp each; ~4 instructions instead of ~20 for expf + an IEEE division: the
        // result moves by <= 2e-7, the rgbnet's own arithmetic by 2e-6); s = (step_id+0.5)/N_samples (lib/dmpigo.py:398) by the rounded reciprocal
        const float r0 = w * k4_sigmoid_fast(o0), r1 = w * k4_sigmoid_fast(o1), r2 = w * k4_sigmoid_fast(o2);
        const float r3 = w * (((float)k + 0.5f) * P.depth_n_inv);
        unsigned f0 = k4_fx(r0, K4_FX_ONE), f1 = k4_fx(r1, K4_FX_ONE), f2 = k4_fx(r2, K4_FX_ONE);
        const unsigned f3 = k4_fx(r3, P.depth_fx);
        const float rs = r0 + r1 + r2;
        if (__builtin_expect(rs != rs, 0)) {                                              // some colour term is NaN (rare path): its field gets the poison bit
            if (r0 != r0) f0 = 0x80000000u;
            if (r1 != r1) f1 = 0x80000000u;
            if (r2 != r2) f2 = 0x80000000u;
            if (lact) {                                                                   // OR, not add: two NaN records must not carry out of the field
                atomicOr(&acc[rl * 2 + 0], (unsigned long long)(f0 & 0x80000000u) | ((unsigned long long)(f1 & 0x80000000u) << 32));
                atomicOr(&acc[rl * 2 + 1], (unsigned long long)(f2 & 0x80000000u));
            }
            f0 &= 0x7fffffffu; f1 &= 0x7fffffffu; f2 &= 0x7fffffffu;
        }
        if (lact && !(P.debug & 512)) {                                  // (512: ablation, no per-ray sums)
            k4_lds_add(&acc[rl * 2 + 0], (unsigned long long)f0 | ((unsigned long long)f1 << 32));
            k4_lds_add(&acc[rl * 2 + 1], (unsigned long long)f2 | ((unsigned long long)f3 << 32));
        }
        __builtin_amdgcn_wave_barrier();
        K4_TSTAMP(6);                                    // sigmoid, blend, per-ray sums
    }

    // ---- per-ray outputs ----
    k4_finish_bundle(P, J, acc, lane);
    __builtin_amdgcn_wave_barrier();
    }   // job
#ifdef K4_SHADE_TIMING
    if (P.counters && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&P.counters[8 + i], tacc[i]);
        const unsigned long long life = __builtin_amdgcn_s_memtime() - tstart;
        atomicAdd(&P.counters[16], life); atomicMax(&P.counters[17], life); atomicAdd(&P.counters[18], 1ull);
    }
#endif
}


static int n_workgroups(int64_t n_rays, int img_w) {
    if (img_w > 0) return (int)(((img_w + 15) / 16) * (((n_rays / img_w) + 15) / 16));
    return (int)((n_rays + 255) / 256);
}

static size_t mlp_floats_b3_of(const k4_mlp_desc* m, int k1p, int nt1 = 3, int nt2 = 3) {       // split-bf16 section (MlpLayoutB3<W, NHID, nt1, nt2>)
    if (m->width == 0) return 0;
    const size_t nb = m->width / 32, kb1 = (size_t)(k1p + 15) / 16;
    size_t n = nb * kb1 * (size_t)nt1 * 64 * 4;
    if (m->n_hidden) n += nb * (m->width / 16) * (size_t)nt2 * 64 * 4 + nb * 2 * 16;
    n += nb * 16 * 2 * 4 + 4;
    return n;
}

static size_t mlp_floats_of(const k4_mlp_desc* m, int k1p) {
    if (m->width == 0) return 0;
    const size_t nb = m->width / 32;
    size_t n = nb * (k1p / 2) * 64;
    if (m->n_hidden) n += nb * nb * 16 * 64 + nb * 64;
    n += nb * 16 * 2 * 4 + 4;
    return n;
}

template <int MODE>
static int launch_march(const MarchParams& P, const k4_mlp_desc* mlp, hipStream_t st) {
    const int nwg = n_workgroups(P.n_rays, P.img_w);
    if (nwg <= 0) return K4_OK;
    const dim3 grid(nwg), block(256);
    const int n_cu = k4_num_cus();
    {
        // one workgroup (4 waves = 4 depth quarters) per bundle
        // MINW = waves per SIMD the register allocation is bounded for: 5 (85 VGPRs, no spills); 6 (80 VGPRs, 6 spilled) measured slower
#ifdef K4_GEOM_TIMING
        if (P.counters) { MarchParams Q = P; Q.timing = P.counters + 8; Q.counters = nullptr;
                          hipLaunchKernelGGL((k4_geom3_kernel<MODE, false, 5>), dim3((unsigned)nwg * 4), block, 0, st, Q); } else
#endif
        if (P.counters) hipLaunchKernelGGL((k4_geom3_kernel<MODE, true, 5>), dim3((unsigned)nwg * 4), block, 0, st, P);
        else if (MODE == MODE_MPI && P.split_k > 0) {
            // depth-ordered: front slab, then the back slab for the rays that are still alive (see MarchParams.split_k)
            MarchParams Q = P;
            Q.slab = 1;
            hipLaunchKernelGGL((k4_geom3_kernel<MODE, false, 5>), dim3((unsigned)nwg * 4), block, 0, st, Q);
            Q.slab = 2;
            hipLaunchKernelGGL((k4_geom3_kernel<MODE, false, 5>), dim3((unsigned)nwg * 4), block, 0, st, Q);
        }
        else hipLaunchKernelGGL((k4_geom3_kernel<MODE, false, 5>), dim3((unsigned)nwg * 4), block, 0, st, P);
    }
    int rc = k4_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k4_order_kernel, dim3(1), dim3(1024), 0, st, P.counts, P.jobs, P.n_bundles, P.qhead);
    rc = k4_check_launch();
    if (rc) return rc;
    const int width = mlp->width, nh = mlp->n_hidden;
    // rgbnet arithmetic (k4_mlp_desc.arith): default = layer 1 exact, layer 2 on 2-term bf16 splits (mlp_mfma_bx<.., 2>, every width);
    // K4_MLP_ARITH_B3 = the exact 3-term form of rounds 2-5 (width <= 64: the width-128 operand does not fit the LDS);
    // K4_MLP_ARITH_FP32 = fp32-input MFMA (bit-exact fp32 FMA chains, 2.7x the matrix-pipe time of B3)
    const int arith = width == 0 ? 0 : mlp->arith == K4_MLP_ARITH_FP32 ? 0 : mlp->arith == K4_MLP_ARITH_B3 ? 3 : 2;
    if (arith == 3 && width > 64) return K4_ERR_UNSUPPORTED;
    const int wg_per_cu = width == 128 ? 1 : K4_SHADE_WG_PER_CU;
    const size_t mlp_fl = arith == 3 ? P.mlp_floats_b3 : arith == 2 ? P.mlp_floats_b2 : P.mlp_floats;
    // FAST: the LLFF / plain-DVGO input shape (see k4_shade_kernel)
    const bool fast = arith >= 2 && width <= 64 && P.k0_layout == K4_K0_CHANNEL_LAST && P.CP == 12 && P.k0_skip == 0 && P.spe == 0 && P.vpe == 0 &&
                      P.dim0 == 15 && P.C == (MODE == MODE_MPI ? 9 : 12) && !k4_env().no_fast_shade &&
                      (uint64_t)P.X * (uint64_t)P.Y * (uint64_t)P.Z * 48ull < 0xffffffffull;
    const size_t lds = sizeof(float) * ((mlp_fl + 3) / 4 * 4 + 4 * (64 * 4 + (width && !fast ? (size_t)P.k1p * 64 : 0)));
    const dim3 sgrid((unsigned)min(nwg, n_cu * wg_per_cu));
    if (lds > 160 * 1024) return K4_ERR_UNSUPPORTED;
#define K4_LAUNCH_K(KERN) do { \
        if (lds > 64 * 1024) { \
            hipError_t e_ = hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e_ != hipSuccess) return (int)e_; } \
        hipLaunchKernelGGL(KERN, sgrid, block, lds, st, P); } while (0)      /* lds varies with the MLP shape: set per launch */
#define K4_LAUNCH(WD, NH) do { if (fast && arith == 2) K4_LAUNCH_K((k4_shade_kernel<MODE, (WD <= 64 ? WD : 32), NH, 2, K4_SHADE_WG_PER_CU, true>)); \
                               else if (fast) K4_LAUNCH_K((k4_shade_kernel<MODE, (WD <= 64 ? WD : 32), NH, 3, K4_SHADE_WG_PER_CU, true>)); \
                               else if (arith == 2) K4_LAUNCH_K((k4_shade_kernel<MODE, WD, NH, 2, (WD == 128 ? 1 : K4_SHADE_WG_PER_CU)>)); \
                               else if (arith == 3) K4_LAUNCH_K((k4_shade_kernel<MODE, (WD <= 64 ? WD : 32), NH, 3>)); \
                               else K4_LAUNCH_K((k4_shade_kernel<MODE, WD, NH, 0, (WD == 128 ? 1 : K4_SHADE_WG_PER_CU)>)); } while (0)
    if (width == 0) K4_LAUNCH_K((k4_shade_kernel<MODE, 0, 0, 0>));
    else if (width == 32 && nh == 0) K4_LAUNCH(32, 0);
    else if (width == 32 && nh == 1) K4_LAUNCH(32, 1);
    else if (width == 64 && nh == 0) K4_LAUNCH(64, 0);
    else if (width == 64 && nh == 1) K4_LAUNCH(64, 1);
    else if (width == 128 && nh == 0) K4_LAUNCH(128, 0);
    else if (width == 128 && nh == 1) K4_LAUNCH(128, 1);
    else return K4_ERR_UNSUPPORTED;
#undef K4_LAUNCH_K
#undef K4_LAUNCH
    return k4_check_launch();
}

// records per bundle: 64 rays x max_steps rounded up to a whole number of 64-sample blocks per depth quarter
static inline int64_t ent_stride_of(int32_t max_steps) { return 64 * (((int64_t)max_steps + 255) / 256 * 256); }

// workspace: [records nb x ent_stride x 8 B][counts nb] | [queue head .. queue length: 64 ints] | [jobs: nb x 8 B], each 256-aligned
struct WsLayout { int64_t counts, qhead, jobs, total; };
static WsLayout ws_layout(int64_t nb, int64_t ent_stride) {
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    WsLayout L;
    L.counts = nb * ent_stride * (int64_t)sizeof(uint2);
    L.qhead = up(L.counts + nb * 4);
    L.jobs = up(L.qhead + 64 * 4);
    L.total = up(L.jobs + nb * 8);
    return L;
}
extern "C" int64_t k4_march_workspace_bytes(int64_t n_rays, int32_t img_w, int32_t max_steps) {
    if (n_rays < 0 || img_w < 0 || max_steps <= 0 || (img_w > 0 && n_rays % img_w != 0)) return -1;
    const int64_t nb = (int64_t)n_workgroups(n_rays, img_w) * 4;
    return ws_layout(nb, ent_stride_of(max_steps)).total;
}

static int fill_common(MarchParams& P, const float* rays_o, const float* rays_d, const float* viewdirs,
                       int64_t n_rays, int32_t img_w, const k4_grid_desc* g, const k4_mlp_desc* m, int mode, int32_t max_steps,
                       void* workspace, int64_t workspace_bytes,
                       float* out_rgb, float* out_depth, float* out_ainv, uint64_t* counters) {
    if (n_rays < 0 || n_rays > 0x7fffffff / 4 || !g || !m) return K4_ERR_BAD_ARG;
    if (n_rays > 0 && (!rays_o || !rays_d || !viewdirs || !out_rgb || !out_depth || !out_ainv || !workspace)) return K4_ERR_BAD_ARG;
    if (!g->density || !g->k0 || !g->mask) return K4_ERR_BAD_ARG;
    if (img_w < 0 || (img_w > 0 && n_rays % img_w != 0)) return K4_ERR_BAD_ARG;
    if (g->k0_layout != K4_K0_CHANNEL_MAJOR && (g->k0_cpad % 4 != 0 || g->k0_cpad < g->k0_ch)) return K4_ERR_BAD_ARG;
    if (g->k0_layout != K4_K0_CHANNEL_MAJOR && g->k0_layout != K4_K0_CHANNEL_LAST) return K4_ERR_BAD_ARG;
    if (m->width == 0 && g->k0_ch != 3) return K4_ERR_BAD_ARG;
    if (m->width != 0 && !m->packed) return K4_ERR_BAD_ARG;
    if (m->k0_skip != 0 && m->k0_skip != 3) return K4_ERR_BAD_ARG;
    if (max_steps <= 0 || max_steps > 16777215) return K4_ERR_BAD_ARG;
    if (n_rays > 0 && workspace_bytes < k4_march_workspace_bytes(n_rays, img_w, max_steps)) return K4_ERR_BAD_ARG;
    P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs;
    P.n_rays = (int)n_rays; P.img_w = img_w; P.img_h = img_w > 0 ? (int)(n_rays / img_w) : 0;
    P.density = g->density; P.k0 = g->k0; P.act_shift = g->act_shift; P.mask = g->mask;
    P.occ = k4_env().geom_skip ? reinterpret_cast<const unsigned long long*>(g->occ_summary) : nullptr;
    P.X = g->dims[0]; P.Y = g->dims[1]; P.Z = g->dims[2];
    P.C = g->k0_ch; P.CP = g->k0_cpad; P.k0_layout = g->k0_layout; P.act_d = g->act_depth;
    P.MX = g->mask_dims[0]; P.MY = g->mask_dims[1]; P.MZ = g->mask_dims[2];
    P.minx = g->xyz_min[0]; P.miny = g->xyz_min[1]; P.minz = g->xyz_min[2];
    P.maxx = g->xyz_max[0]; P.maxy = g->xyz_max[1]; P.maxz = g->xyz_max[2];
    P.msx = g->xyz2ijk_scale[0]; P.msy = g->xyz2ijk_scale[1]; P.msz = g->xyz2ijk_scale[2];
    P.mtx = g->xyz2ijk_shift[0]; P.mty = g->xyz2ijk_shift[1]; P.mtz = g->xyz2ijk_shift[2];
    P.lenx = P.maxx - P.minx; P.leny = P.maxy - P.miny; P.lenz = P.maxz - P.minz;       // fp32 subtraction, as lib/grid.py:123
    P.rlenx = 1.0f / P.lenx; P.rleny = 1.0f / P.leny; P.rlenz = 1.0f / P.lenz;           // IEEE division: correctly rounded
    P.mlp = m->packed; P.dim0 = m->dim0; P.k1p = (m->dim0 + 2) & ~1;
    P.mlp_floats = (int)mlp_floats_of(m, P.k1p);
    P.mlp_floats_b3 = (int)mlp_floats_b3_of(m, P.k1p);
    P.mlp_floats_b2 = (int)mlp_floats_b3_of(m, P.k1p, K4_B2_L1_TERMS, 2);
    P.vpe = m->viewbase_pe; P.spe = m->spatial_pe; P.k0_skip = m->k0_skip;
    P.max_steps = max_steps;
    if (ent_stride_of(max_steps) > 0x7fffffff) return K4_ERR_BAD_ARG;
    P.ent_stride = (int)ent_stride_of(max_steps);
    const int64_t nb = (int64_t)n_workgroups(n_rays, img_w) * 4;
    const WsLayout L = ws_layout(nb, ent_stride_of(max_steps));
    P.entries = (uint2*)workspace;
    P.counts = (int*)((char*)workspace + L.counts);
    P.qhead = (int*)((char*)workspace + L.qhead);
    P.jobs = (int2*)((char*)workspace + L.jobs);
    P.n_bundles = (int)nb;
    P.debug = k4_env().debug; P.serp = 1;
    // XCD bands of the geometry kernel: ONE row of 16x16-pixel workgroup tiles (x 4 bundles) per band, dealt round-robin to
    // the 8 XCDs.  One contiguous band per XCD (0) left the XCDs 0.61..1.31 of the mean work on the LLFF frames -- the kernel ran at
    // the pace of the heaviest eighth of the image.
    P.band_blocks = img_w > 0 ? ((img_w + 15) / 16) * 4 : 64;
    if (max_steps > 65536) return K4_ERR_UNSUPPORTED;                                   // group index of the skip list is 10 bits per depth quarter
    P.out_rgb = out_rgb; P.out_depth = out_depth; P.out_ainv = out_ainv;
    P.counters = (unsigned long long*)counters;
    return K4_OK;
}

// fixed-point scale of the per-ray depth sums: sum_w s <= the largest s = (max_steps + 0.5) / depth_n a record can carry (sum_w <= 1);
// rounded up to a power of two so that 2^30 / it is one (exact scaling both ways)
static void set_depth_fx(MarchParams& P, int max_steps) {
    const double smax = ((double)max_steps + 0.5) / (double)(P.depth_n > 0 ? P.depth_n : 1);
    int e = 0;
    while (e < 60 && ldexp(1.0, e) < smax) ++e;
    P.depth_n_inv = (float)(1.0 / (double)(P.depth_n > 0 ? P.depth_n : 1));
    P.depth_fx = (float)ldexp(1.0, 30 - e);
    P.depth_fx_inv = (float)ldexp(1.0, e - 30);
}

extern "C" int k4_abi_version(void) { return K4_ABI_VERSION; }

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static const K4Env g_k4_env = {        // namespace-scope constant: initialised while the library is loaded, immutable afterwards
    env_int("K4_GEOM_SKIP", 1), env_int("K4_DEBUG", 0), env_int("K4_SR_DEBUG", 0), (env_int("K4_DEBUG", 0) & 1024) != 0};
const K4Env& k4_env() { return g_k4_env; }
int k4_num_cus() {
    static int n_cu[K4_MAX_DEVICES];
    const int dev = k4_device_ordinal();
    if (n_cu[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n_cu[dev] = v;
    }
    return n_cu[dev];
}

extern "C" int64_t k4_mlp_packed_floats(int32_t dim0, int32_t width, int32_t n_hidden) {
    if (width == 0) return 0;
    if ((width != 32 && width != 64 && width != 128) || n_hidden < 0 || n_hidden > 1 || dim0 <= 0) return -1;
    k4_mlp_desc m{};
    m.width = width; m.n_hidden = n_hidden; m.dim0 = dim0;
    return (int64_t)(mlp_floats_of(&m, (dim0 + 2) & ~1) + mlp_floats_b3_of(&m, (dim0 + 2) & ~1) + mlp_floats_b3_of(&m, (dim0 + 2) & ~1, K4_B2_L1_TERMS, 2));
}
extern "C" int k4_mlp_b2_layer1_terms(void) { return K4_B2_L1_TERMS;
}

extern "C" int k4_march_mpi_fwd(const float* rays_o, const float* rays_d, const float* viewdirs,
                                int64_t n_rays, int32_t img_w,
                                const k4_grid_desc* grid, const k4_mlp_desc* mlp,
                                int32_t n_samples, float interval, float fast_color_thres, float bg,
                                void* workspace, int64_t workspace_bytes,
                                float* out_rgb, float* out_depth, float* out_alphainv,
                                uint64_t* out_counters, void* stream) {
    MarchParams P{};
    int rc = fill_common(P, rays_o, rays_d, viewdirs, n_rays, img_w, grid, mlp, MODE_MPI, n_samples, workspace, workspace_bytes,
                         out_rgb, out_depth, out_alphainv, out_counters);
    if (rc) return rc;
    if (!grid->act_shift || grid->act_depth <= 0 || n_samples < 2) return K4_ERR_BAD_ARG;
    if (mlp->width != 0) {
        const int want = grid->k0_ch + 3 + 6 * mlp->spatial_pe + 3 + 6 * mlp->viewbase_pe;      // lib/dmpigo.py:85
        if (mlp->dim0 != want || mlp->k0_skip != 0) return K4_ERR_BAD_ARG;
    }
    P.n_samples = n_samples; P.depth_n = n_samples; P.nsm1 = (float)(n_samples - 1);
    if (grid->depth_split < 0 || grid->depth_split % 64 != 0 || grid->depth_split >= n_samples) return K4_ERR_BAD_ARG;
    P.split_k = k4_env().debug & 2048 ? 0 : grid->depth_split;                               // (K4_DEBUG & 2048: single launch whatever the descriptor says, A/B)
    set_depth_fx(P, n_samples);
    P.shift = 0.f;                                                                            // lib/dmpigo.py:261
    P.interval = interval; P.thres = fast_color_thres; P.bg = bg;
    return launch_march<MODE_MPI>(P, mlp, (hipStream_t)stream);
}

extern "C" int k4_march_dvgo_fwd(const float* rays_o, const float* rays_d, const float* viewdirs,
                                 int64_t n_rays, int32_t img_w,
                                 const k4_grid_desc* grid, const k4_mlp_desc* mlp,
                                 float near, float far, float stepdist, int32_t max_steps, int32_t depth_n_samples,
                                 float act_shift, float interval, float fast_color_thres, float bg,
                                 void* workspace, int64_t workspace_bytes,
                                 float* out_rgb, float* out_depth, float* out_alphainv,
                                 uint64_t* out_counters, void* stream) {
    MarchParams P{};
    int rc = fill_common(P, rays_o, rays_d, viewdirs, n_rays, img_w, grid, mlp, MODE_DVGO, max_steps, workspace, workspace_bytes,
                         out_rgb, out_depth, out_alphainv, out_counters);
    if (rc) return rc;
    if (!(stepdist > 0.f) || depth_n_samples <= 0 || mlp->spatial_pe != 0) return K4_ERR_BAD_ARG;
    if (mlp->width != 0) {
        const int want = grid->k0_ch - mlp->k0_skip + 3 + 6 * mlp->viewbase_pe;                 // lib/dvgo.py:94-101
        if (mlp->dim0 != want) return K4_ERR_BAD_ARG;
    }
    P.depth_n = depth_n_samples; P.stepdist = stepdist; P.near_ = near; P.far_ = far;
    set_depth_fx(P, max_steps);
    P.shift = act_shift; P.interval = interval; P.thres = fast_color_thres; P.bg = bg;
    return launch_march<MODE_DVGO>(P, mlp, (hipStream_t)stream);
}
