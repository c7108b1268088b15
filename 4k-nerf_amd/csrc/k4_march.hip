// Fused voxel-grid ray marcher for gfx950 (MI355X).
//
// ONE launch computes, per ray, what the reference spreads over ~25 launches and >=4 host syncs per
// 8192-ray chunk inside DirectMPIGO.forward (lib/dmpigo.py:292-427) / DirectVoxGO.forward
// (lib/dvgo.py:327-448):
//   sampler (render_utils_kernel.cu:245-270 NDC, :12-242 ray-AABB) -> bbox test -> occupancy byte
//   (:374-392) -> trilinear density (+ per-plane act_shift grid for MPI) (lib/grid.py:117-128) ->
//   raw2alpha (:431-443) -> alpha>thres -> sequential transmittance with the T<1e-3 early stop
//   (:577-605) -> w>thres -> trilinear k0 features + PE + rgbnet MLP + sigmoid
//   (lib/dmpigo.py:336-379, lib/dvgo.py:372-412) -> sum_w rgb, sum_w s, alphainv_last*bg
//   (lib/dmpigo.py:382-398,418-424).
// No per-sample intermediate ever reaches HBM; the only global writes are 5 floats per ray.
//
// Mapping (wave64, no inter-wave communication, deterministic):
//   * a wavefront owns a bundle of 64 rays (an 8x8 pixel tile when the caller says the rays are an
//     image, else 64 consecutive rays); a 256-thread workgroup = 4 independent waves = a 16x16 tile;
//     workgroups are XCD-banded so that neighbouring tiles share an L2.
//   * geometry phase: the wave walks ONE ray at a time, lanes = 64 consecutive samples.  NDC/MPI rays
//     advance mostly along Z, the contiguous axis of the [X][Y][Z] grids, so the 8 density gathers
//     and the occupancy byte of a wave coalesce into a few 256-B lines.  Transmittance is the exact
//     sequential product of the reference over the (few) alpha-passing lanes (ballot + readlane).
//   * surviving samples are compacted (ballot/mbcnt) into a per-wave LDS queue; whenever it holds
//     >= 64 entries the wave shades them with ALL 64 lanes busy: lane = sample, 8-corner k0 gather,
//     the MLP evaluated lane-per-sample with wave-uniform weights fetched by scalar loads (SGPR
//     operands of v_fmac), a segmented wave scan folds w*rgb / w*s into per-ray LDS accumulators.
//
// Roofline: HBM (gather/interpolation), see DESIGN.md for the algorithmic-byte accounting.
#include "k4_common.h"

#define MODE_MPI  0
#define MODE_DVGO 1
#define QCAP 128

struct MarchParams {
    const float* rays_o; const float* rays_d; const float* viewdirs;
    int n_rays; int img_w; int img_h;
    const float* density; const float* k0; const float* act_shift; const uint8_t* mask;
    int X, Y, Z; int C; int CP; int k0_layout; int act_d;
    int MX, MY, MZ;
    float minx, miny, minz, maxx, maxy, maxz;
    float msx, msy, msz, mtx, mty, mtz;
    const float* mlp; int dim0; int vpe; int spe; int k0_skip;
    int n_samples;          // MPI: samples per ray; DVGO: unused
    int depth_n;            // denominator of s = (k+0.5)/depth_n
    float nsm1;             // MPI: (float)(n_samples-1)
    float stepdist, near_, far_, shift, interval, thres, bg;
    float* out_rgb; float* out_depth; float* out_ainv; unsigned long long* counters;
};

struct WaveLds {
    float raytab[64][8];    // per ray of the bundle: start xyz, dir xyz (p_k = start + dir*t_k)
    float acc[64][8];       // r,g,b, depth, alphainv_last
    unsigned qkey[QCAP];    // ray_local<<24 | step
    float qw[QCAP];         // blending weight
};

template <int MODE>
__device__ __forceinline__ float step_t(const MarchParams& P, int k) {
    // MPI : dist = (float)i_step / (N_samples-1)      render_utils_kernel.cu:260
    // DVGO: dist = stepdist * i_step                  render_utils_kernel.cu:184
    return MODE == MODE_MPI ? (float)k / P.nsm1 : P.stepdist * (float)k;
}

__device__ __forceinline__ int ray_index(const MarchParams& P, int bundle_x, int bundle_y, int bundle_lin, int r) {
    if (P.img_w > 0) {
        const int px = bundle_x + (r & 7), py = bundle_y + (r >> 3);
        return (px < P.img_w && py < P.img_h) ? py * P.img_w + px : -1;
    }
    const int ray = bundle_lin + r;
    return ray < P.n_rays ? ray : -1;
}

template <int MODE, int WIDTH, int NHID>
__global__ __launch_bounds__(256) void k4_march_kernel(const MarchParams P) {
    __shared__ WaveLds lds_all[4];
    const int lane = k4_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WaveLds& L = lds_all[wv];

    // ---- bundle of 64 rays owned by this wave ----
    const int wg = k4_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int bundle_x = 0, bundle_y = 0, bundle_lin = 0;
    if (P.img_w > 0) {
        const int wgx = (P.img_w + 15) >> 4;
        bundle_x = (wg % wgx) * 16 + (wv & 1) * 8;
        bundle_y = (wg / wgx) * 16 + (wv >> 1) * 8;
    } else {
        bundle_lin = (wg * 4 + wv) * 64;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) L.acc[lane][c] = (c == 4) ? 1.f : 0.f;

    const k4_cptr c_o = k4_const(P.rays_o), c_d = k4_const(P.rays_d);
    const bool use_thres = P.thres > 0.f;

    unsigned long long n_inb = 0, n_mask = 0, n_alpha = 0, n_shade = 0;

    // ---- wave-uniform marching state ----
    int r = 0, it = 0, nsteps = 0, cnt = 0, head = 0;
    float T = 1.f;
    float sx = 0, sy = 0, sz = 0, dx = 0, dy = 0, dz = 0;

    for (;;) {
        const bool rays_left = r < 64;
        if (rays_left) {
            const int ray = __builtin_amdgcn_readfirstlane(ray_index(P, bundle_x, bundle_y, bundle_lin, r));
            if (ray < 0) { r += 1; continue; }
            if (it == 0) {
                const float ox = c_o[ray * 3 + 0], oy = c_o[ray * 3 + 1], oz = c_o[ray * 3 + 2];
                const float vx = c_d[ray * 3 + 0], vy = c_d[ray * 3 + 1], vz = c_d[ray * 3 + 2];
                if (MODE == MODE_MPI) {
                    sx = ox; sy = oy; sz = oz; dx = vx; dy = vy; dz = vz;
                    nsteps = P.n_samples;
                } else {
                    // infer_t_minmax / infer_n_samples / infer_ray_start_dir  (render_utils_kernel.cu:12-79)
                    const float ex = (vx == 0.f) ? 1e-6f : vx, ey = (vy == 0.f) ? 1e-6f : vy, ez = (vz == 0.f) ? 1e-6f : vz;
                    const float ax = (P.maxx - ox) / ex, ay = (P.maxy - oy) / ey, az = (P.maxz - oz) / ez;
                    const float bx = (P.minx - ox) / ex, by = (P.miny - oy) / ey, bz = (P.minz - oz) / ez;
                    const float t_min = fmaxf(fminf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), P.far_), P.near_);
                    const float t_max = fmaxf(fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), P.far_), P.near_);
                    const float rnorm = sqrtf(fmaf(vz, vz, fmaf(vy, vy, vx * vx)));
                    const float nf = fmaxf(ceilf((t_max - t_min) * rnorm / P.stepdist), 1.f);
                    nsteps = (int)fminf(nf, 16777215.f);
                    sx = fmaf(vx, t_min, ox); sy = fmaf(vy, t_min, oy); sz = fmaf(vz, t_min, oz);
                    dx = vx / rnorm; dy = vy / rnorm; dz = vz / rnorm;
                }
                nsteps = __builtin_amdgcn_readfirstlane(nsteps);
                if (lane == 0) {
                    L.raytab[r][0] = sx; L.raytab[r][1] = sy; L.raytab[r][2] = sz;
                    L.raytab[r][3] = dx; L.raytab[r][4] = dy; L.raytab[r][5] = dz;
                }
                T = 1.f;
            }

            // ------------------ geometry: 64 consecutive samples of ray r ------------------
            const int k = it * 64 + lane;
            const float tk = step_t<MODE>(P, k);
            const float px = fmaf(dx, tk, sx), py = fmaf(dy, tk, sy), pz = fmaf(dz, tk, sz);
            const bool inb = (k < nsteps) &&
                !((P.minx > px) | (P.miny > py) | (P.minz > pz) | (P.maxx < px) | (P.maxy < py) | (P.maxz < pz));
            bool m = false;
            if (inb) {
                const int mi = k4_round_half_away(fmaf(px, P.msx, P.mtx));
                const int mj = k4_round_half_away(fmaf(py, P.msy, P.mty));
                const int mk = k4_round_half_away(fmaf(pz, P.msz, P.mtz));
                if ((unsigned)mi < (unsigned)P.MX && (unsigned)mj < (unsigned)P.MY && (unsigned)mk < (unsigned)P.MZ)
                    m = P.mask[((size_t)mi * P.MY + mj) * P.MZ + mk] != 0;
            }
            float alpha = 0.f;
            bool act = false;
            const uint64_t mball = __ballot(m);
            if (P.counters) { n_inb += __popcll(__ballot(inb)); n_mask += __popcll(mball); }
            if (mball) {
                if (m) {
                    const float nx = k4_norm_coord(px, P.minx, P.maxx);
                    const float ny = k4_norm_coord(py, P.miny, P.maxy);
                    const float nz = k4_norm_coord(pz, P.minz, P.maxz);
                    const K4Tri t = k4_tri_setup(k4_unnorm(nx, P.X), k4_unnorm(ny, P.Y), k4_unnorm(nz, P.Z));
                    float sigma = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int x = t.x0 + K4_CX(c), y = t.y0 + K4_CY(c), z = t.z0 + K4_CZ(c);
                        const bool ok = (unsigned)x < (unsigned)P.X && (unsigned)y < (unsigned)P.Y && (unsigned)z < (unsigned)P.Z;
                        const float v = ok ? P.density[((size_t)x * P.Y + y) * P.Z + z] : 0.f;
                        sigma += v * t.w[c];
                    }
                    if (MODE == MODE_MPI) {
                        // act_shift grid [1,1,D]: x/y sizes are 1 -> only z interpolates (lib/dmpigo.py:48-58,316)
                        const float ua = k4_unnorm(nz, P.act_d);
                        const float fa = floorf(ua);
                        const int a0 = (int)fa;
                        const float lo = ((unsigned)a0 < (unsigned)P.act_d) ? P.act_shift[a0] : 0.f;
                        const float hi = ((unsigned)(a0 + 1) < (unsigned)P.act_d) ? P.act_shift[a0 + 1] : 0.f;
                        sigma += lo * ((fa + 1.f) - ua) + hi * (ua - fa);
                    }
                    // raw2alpha: e = exp(d+shift); alpha = 1 - (1+e)^(-interval)   render_utils_kernel.cu:439-441
                    const float e = expf(sigma + P.shift);
                    alpha = (P.interval == 1.f) ? 1.f - 1.f / (1.f + e) : 1.f - powf(1.f + e, -P.interval);
                    act = use_thres ? (alpha > P.thres) : true;
                }
                // ---- alpha2weight: exact sequential scan over the active lanes (render_utils_kernel.cu:591-603) ----
                uint64_t bm = __ballot(act);
                if (P.counters) n_alpha += __popcll(bm);
                float w = 0.f;
                bool proc = false, stop = false;
                while (bm) {
                    const int l = __builtin_ctzll(bm);
                    const float a = k4_readlane(alpha, l);
                    if (lane == l) { w = T * a; proc = true; }
                    T = (float)((double)T * (1.0 - (double)a));           // `T_cum *= (1. - alpha[i])`
                    bm &= bm - 1;
                    if ((double)T < 1e-3) { stop = true; break; }         // sample l is still counted (:597-600)
                }
                const bool shade = proc && (use_thres ? (w > P.thres) : true);
                const uint64_t sm = __ballot(shade);
                if (shade) {
                    const int pos = (head + cnt + k4_prefix(sm)) & (QCAP - 1);
                    L.qkey[pos] = ((unsigned)r << 24) | (unsigned)k;
                    L.qw[pos] = w;
                }
                const int ns = __popcll(sm);
                cnt += ns;
                if (P.counters) n_shade += ns;
                if (stop) it = 0x3fffff;                                   // force ray end
            }
            it += 1;
            if (it * 64 >= nsteps || it > 0x3fffff) {
                if (lane == 0) L.acc[r][4] = T;                            // alphainv_last (:603)
                r += 1; it = 0;
            }
        }

        // ------------------ shading: 64 queued samples, one per lane ------------------
        if (cnt >= 64 || (!rays_left && cnt > 0)) {
            const int nproc = cnt < 64 ? cnt : 64;
            const bool lact = lane < nproc;
            const int e = (head + (lact ? lane : 0)) & (QCAP - 1);
            const unsigned key = L.qkey[e];
            const float w = lact ? L.qw[e] : 0.f;
            const int rl = (int)(key >> 24);
            const int k = (int)(key & 0xffffffu);
            const float tk = step_t<MODE>(P, k);
            const float px = fmaf(L.raytab[rl][3], tk, L.raytab[rl][0]);
            const float py = fmaf(L.raytab[rl][4], tk, L.raytab[rl][1]);
            const float pz = fmaf(L.raytab[rl][5], tk, L.raytab[rl][2]);
            const float nx = k4_norm_coord(px, P.minx, P.maxx);
            const float ny = k4_norm_coord(py, P.miny, P.maxy);
            const float nz = k4_norm_coord(pz, P.minz, P.maxz);
            const K4Tri t = k4_tri_setup(k4_unnorm(nx, P.X), k4_unnorm(ny, P.Y), k4_unnorm(nz, P.Z));
            size_t cidx[8];
            float cw[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int x = t.x0 + K4_CX(c), y = t.y0 + K4_CY(c), z = t.z0 + K4_CZ(c);
                const bool ok = (unsigned)x < (unsigned)P.X && (unsigned)y < (unsigned)P.Y && (unsigned)z < (unsigned)P.Z;
                cidx[c] = ok ? ((size_t)x * P.Y + y) * P.Z + z : 0;
                cw[c] = ok ? t.w[c] : 0.f;
            }
            float o0, o1, o2;
            if (WIDTH == 0) {
                // rgbnet is None: rgb = sigmoid(k0)   (lib/dvgo.py:377-379)
                float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float q = (P.k0_layout == K4_K0_CHANNEL_LAST)
                            ? P.k0[cidx[c] * P.CP + ch]
                            : P.k0[(size_t)ch * P.X * P.Y * P.Z + cidx[c]];
                        v[ch] += q * cw[c];
                    }
                o0 = v[0]; o1 = v[1]; o2 = v[2];
            } else {
                constexpr int W = WIDTH > 0 ? WIDTH : 1;
                const k4_cptr M = k4_const(P.mlp);
                const k4_cptr W1T = M;                                  // [dim0][W]
                const k4_cptr B1 = M + (size_t)P.dim0 * W;              // [W]
                float h[W];
#pragma unroll
                for (int j = 0; j < W; ++j) h[j] = B1[j];
                float dif0 = 0.f, dif1 = 0.f, dif2 = 0.f;               // k0[:, :3] when rgbnet_direct=False
                auto rank1 = [&](int i, float x) {
                    const k4_cptr wr = W1T + (size_t)i * W;
#pragma unroll
                    for (int j = 0; j < W; ++j) h[j] = fmaf(wr[j], x, h[j]);
                };
                // --- features: trilinear k0 channels (lib/grid.py:117-128) ---
                if (P.k0_layout == K4_K0_CHANNEL_LAST) {
                    for (int g = 0; g < P.CP; g += 4) {
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float4 q = *reinterpret_cast<const float4*>(P.k0 + cidx[c] * P.CP + g);
                            v.x += q.x * cw[c]; v.y += q.y * cw[c]; v.z += q.z * cw[c]; v.w += q.w * cw[c];
                        }
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int ch = g + cc;
                            if (ch < P.k0_skip) { if (ch == 0) dif0 = vv[cc]; else if (ch == 1) dif1 = vv[cc]; else dif2 = vv[cc]; }
                            else if (ch < P.C) rank1(ch - P.k0_skip, vv[cc]);
                        }
                    }
                } else {
                    const size_t plane = (size_t)P.X * P.Y * P.Z;
                    for (int ch = 0; ch < P.C; ++ch) {
                        float v = 0.f;
#pragma unroll
                        for (int c = 0; c < 8; ++c) v += P.k0[plane * ch + cidx[c]] * cw[c];
                        if (ch < P.k0_skip) { if (ch == 0) dif0 = v; else if (ch == 1) dif1 = v; else dif2 = v; }
                        else rank1(ch - P.k0_skip, v);
                    }
                }
                int fi = P.C - P.k0_skip;
                if (MODE == MODE_MPI) {
                    // pe_spa = normalised position flipped to (z,y,x); [v, sin(v x freq), cos(...)]   lib/dmpigo.py:338,350-351
                    const float pe[3] = {nz, ny, nx};
#pragma unroll
                    for (int c = 0; c < 3; ++c) rank1(fi + c, pe[c]);
                    fi += 3;
                    for (int f = 0; f < P.spe; ++f) {
                        const float fr = (float)(1 << f);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            rank1(fi + c * P.spe + f, sinf(pe[c] * fr));
                            rank1(fi + 3 * P.spe + c * P.spe + f, cosf(pe[c] * fr));
                        }
                    }
                    fi += 6 * P.spe;
                }
                {
                    // viewdirs_emb[ray_id]   lib/dmpigo.py:347-349, lib/dvgo.py:387-389
                    const int ray = ray_index(P, bundle_x, bundle_y, bundle_lin, rl);
                    const int rs = ray < 0 ? 0 : ray;
                    const float vd[3] = {P.viewdirs[rs * 3 + 0], P.viewdirs[rs * 3 + 1], P.viewdirs[rs * 3 + 2]};
#pragma unroll
                    for (int c = 0; c < 3; ++c) rank1(fi + c, vd[c]);
                    fi += 3;
                    for (int f = 0; f < P.vpe; ++f) {
                        const float fr = (float)(1 << f);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            rank1(fi + c * P.vpe + f, sinf(vd[c] * fr));
                            rank1(fi + 3 * P.vpe + c * P.vpe + f, cosf(vd[c] * fr));
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < W; ++j) h[j] = fmaxf(h[j], 0.f);     // ReLU
                k4_cptr Q = B1 + W;
                if (NHID == 1) {
                    const k4_cptr W2 = Q, B2 = Q + (size_t)W * W;
                    const k4_cptr WoT = B2 + W, Bo = WoT + (size_t)W * 4;
                    o0 = Bo[0]; o1 = Bo[1]; o2 = Bo[2];
#pragma unroll 1
                    for (int j = 0; j < W; j += 4) {
                        float a0 = B2[j], a1 = B2[j + 1], a2 = B2[j + 2], a3 = B2[j + 3];
                        const k4_cptr r0 = W2 + (size_t)j * W;
#pragma unroll
                        for (int i = 0; i < W; ++i) {
                            a0 = fmaf(r0[i], h[i], a0);
                            a1 = fmaf(r0[W + i], h[i], a1);
                            a2 = fmaf(r0[2 * W + i], h[i], a2);
                            a3 = fmaf(r0[3 * W + i], h[i], a3);
                        }
                        a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); a2 = fmaxf(a2, 0.f); a3 = fmaxf(a3, 0.f);
                        const k4_cptr wo = WoT + (size_t)j * 4;
                        o0 = fmaf(wo[0], a0, o0); o1 = fmaf(wo[1], a0, o1); o2 = fmaf(wo[2], a0, o2);
                        o0 = fmaf(wo[4], a1, o0); o1 = fmaf(wo[5], a1, o1); o2 = fmaf(wo[6], a1, o2);
                        o0 = fmaf(wo[8], a2, o0); o1 = fmaf(wo[9], a2, o1); o2 = fmaf(wo[10], a2, o2);
                        o0 = fmaf(wo[12], a3, o0); o1 = fmaf(wo[13], a3, o1); o2 = fmaf(wo[14], a3, o2);
                    }
                } else {
                    const k4_cptr WoT = Q, Bo = WoT + (size_t)W * 4;
                    o0 = Bo[0]; o1 = Bo[1]; o2 = Bo[2];
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        o0 = fmaf(WoT[j * 4 + 0], h[j], o0);
                        o1 = fmaf(WoT[j * 4 + 1], h[j], o1);
                        o2 = fmaf(WoT[j * 4 + 2], h[j], o2);
                    }
                }
                o0 += dif0; o1 += dif1; o2 += dif2;                      // rgb_logit + k0_diffuse (lib/dvgo.py:412)
            }
            // sigmoid, blend
            float v0 = w * (1.f / (1.f + expf(-o0)));
            float v1 = w * (1.f / (1.f + expf(-o1)));
            float v2 = w * (1.f / (1.f + expf(-o2)));
            float v3 = w * (((float)k + 0.5f) / (float)P.depth_n);     // s = (step_id+0.5)/N_samples (lib/dmpigo.py:398)
            // segmented inclusive scan keyed by ray (entries are sorted by ray)
            const int keyr = lact ? rl : (256 + lane);
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int ok_ = __shfl_up(keyr, off);
                const float u0 = __shfl_up(v0, off), u1 = __shfl_up(v1, off), u2 = __shfl_up(v2, off), u3 = __shfl_up(v3, off);
                if (lane >= off && ok_ == keyr) { v0 += u0; v1 += u1; v2 += u2; v3 += u3; }
            }
            const int nextk = __shfl_down(keyr, 1);
            if (lact && (lane == 63 || nextk != keyr)) {
                L.acc[rl][0] += v0; L.acc[rl][1] += v1; L.acc[rl][2] += v2; L.acc[rl][3] += v3;
            }
            head = (head + nproc) & (QCAP - 1);
            cnt -= nproc;
        }
        if (!rays_left && cnt == 0) break;
    }

    // ---- per-ray outputs: rgb_marched = sum + alphainv_last*bg (lib/dmpigo.py:397), depth, alphainv_last ----
    {
        const int ray = ray_index(P, bundle_x, bundle_y, bundle_lin, lane);
        if (ray >= 0) {
            const float ainv = L.acc[lane][4];
            P.out_rgb[(size_t)ray * 3 + 0] = L.acc[lane][0] + ainv * P.bg;
            P.out_rgb[(size_t)ray * 3 + 1] = L.acc[lane][1] + ainv * P.bg;
            P.out_rgb[(size_t)ray * 3 + 2] = L.acc[lane][2] + ainv * P.bg;
            P.out_depth[ray] = L.acc[lane][3];
            P.out_ainv[ray] = ainv;
        }
    }
    if (P.counters && lane == 0) {
        atomicAdd(&P.counters[0], n_inb); atomicAdd(&P.counters[1], n_mask);
        atomicAdd(&P.counters[2], n_alpha); atomicAdd(&P.counters[3], n_shade);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int MODE>
static int launch_march(const MarchParams& P, const k4_mlp_desc* mlp, hipStream_t st) {
    int nwg;
    if (P.img_w > 0) nwg = ((P.img_w + 15) / 16) * ((P.img_h + 15) / 16);
    else nwg = (P.n_rays + 255) / 256;
    if (nwg <= 0) return K4_OK;
    const dim3 grid(nwg), block(256);
    const int width = mlp->width, nh = mlp->n_hidden;
#define K4_LAUNCH(WD, NH) hipLaunchKernelGGL((k4_march_kernel<MODE, WD, NH>), grid, block, 0, st, P)
    if (width == 0) K4_LAUNCH(0, 0);
    else if (width == 32 && nh == 0) K4_LAUNCH(32, 0);
    else if (width == 32 && nh == 1) K4_LAUNCH(32, 1);
    else if (width == 64 && nh == 0) K4_LAUNCH(64, 0);
    else if (width == 64 && nh == 1) K4_LAUNCH(64, 1);
    else if (width == 128 && nh == 0) K4_LAUNCH(128, 0);
    else if (width == 128 && nh == 1) K4_LAUNCH(128, 1);
    else return K4_ERR_UNSUPPORTED;
#undef K4_LAUNCH
    return k4_check_launch();
}

static int fill_common(MarchParams& P, const float* rays_o, const float* rays_d, const float* viewdirs,
                       int64_t n_rays, int32_t img_w, const k4_grid_desc* g, const k4_mlp_desc* m,
                       float* out_rgb, float* out_depth, float* out_ainv, uint64_t* counters) {
    if (!rays_o || !rays_d || !viewdirs || !g || !m || !out_rgb || !out_depth || !out_ainv) return K4_ERR_BAD_ARG;
    if (n_rays < 0 || n_rays > 0x7fffffff / 4) return K4_ERR_BAD_ARG;
    if (!g->density || !g->k0 || !g->mask) return K4_ERR_BAD_ARG;
    if (img_w < 0 || (img_w > 0 && n_rays % img_w != 0)) return K4_ERR_BAD_ARG;
    if (g->k0_layout == K4_K0_CHANNEL_LAST && (g->k0_cpad % 4 != 0 || g->k0_cpad < g->k0_ch)) return K4_ERR_BAD_ARG;
    if (m->width == 0 && g->k0_ch != 3) return K4_ERR_BAD_ARG;
    if (m->width != 0 && !m->packed) return K4_ERR_BAD_ARG;
    if (m->k0_skip != 0 && m->k0_skip != 3) return K4_ERR_BAD_ARG;
    P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs;
    P.n_rays = (int)n_rays; P.img_w = img_w; P.img_h = img_w > 0 ? (int)(n_rays / img_w) : 0;
    P.density = g->density; P.k0 = g->k0; P.act_shift = g->act_shift; P.mask = g->mask;
    P.X = g->dims[0]; P.Y = g->dims[1]; P.Z = g->dims[2];
    P.C = g->k0_ch; P.CP = g->k0_cpad; P.k0_layout = g->k0_layout; P.act_d = g->act_depth;
    P.MX = g->mask_dims[0]; P.MY = g->mask_dims[1]; P.MZ = g->mask_dims[2];
    P.minx = g->xyz_min[0]; P.miny = g->xyz_min[1]; P.minz = g->xyz_min[2];
    P.maxx = g->xyz_max[0]; P.maxy = g->xyz_max[1]; P.maxz = g->xyz_max[2];
    P.msx = g->xyz2ijk_scale[0]; P.msy = g->xyz2ijk_scale[1]; P.msz = g->xyz2ijk_scale[2];
    P.mtx = g->xyz2ijk_shift[0]; P.mty = g->xyz2ijk_shift[1]; P.mtz = g->xyz2ijk_shift[2];
    P.mlp = m->packed; P.dim0 = m->dim0; P.vpe = m->viewbase_pe; P.spe = m->spatial_pe; P.k0_skip = m->k0_skip;
    P.out_rgb = out_rgb; P.out_depth = out_depth; P.out_ainv = out_ainv;
    P.counters = (unsigned long long*)counters;
    return K4_OK;
}

extern "C" int k4_abi_version(void) { return K4_ABI_VERSION; }

extern "C" int k4_march_mpi_fwd(const float* rays_o, const float* rays_d, const float* viewdirs,
                                int64_t n_rays, int32_t img_w,
                                const k4_grid_desc* grid, const k4_mlp_desc* mlp,
                                int32_t n_samples, float interval, float fast_color_thres, float bg,
                                float* out_rgb, float* out_depth, float* out_alphainv,
                                uint64_t* out_counters, void* stream) {
    MarchParams P{};
    int rc = fill_common(P, rays_o, rays_d, viewdirs, n_rays, img_w, grid, mlp, out_rgb, out_depth, out_alphainv, out_counters);
    if (rc) return rc;
    if (!grid->act_shift || grid->act_depth <= 0 || n_samples < 2 || n_samples > 16777215) return K4_ERR_BAD_ARG;
    if (mlp->width != 0) {
        const int want = grid->k0_ch + 3 + 6 * mlp->spatial_pe + 3 + 6 * mlp->viewbase_pe;      // lib/dmpigo.py:85
        if (mlp->dim0 != want || mlp->k0_skip != 0) return K4_ERR_BAD_ARG;
    }
    P.n_samples = n_samples; P.depth_n = n_samples; P.nsm1 = (float)(n_samples - 1);
    P.shift = 0.f;                                                                            // lib/dmpigo.py:261
    P.interval = interval; P.thres = fast_color_thres; P.bg = bg;
    return launch_march<MODE_MPI>(P, mlp, (hipStream_t)stream);
}

extern "C" int k4_march_dvgo_fwd(const float* rays_o, const float* rays_d, const float* viewdirs,
                                 int64_t n_rays, int32_t img_w,
                                 const k4_grid_desc* grid, const k4_mlp_desc* mlp,
                                 float near, float far, float stepdist, int32_t depth_n_samples,
                                 float act_shift, float interval, float fast_color_thres, float bg,
                                 float* out_rgb, float* out_depth, float* out_alphainv,
                                 uint64_t* out_counters, void* stream) {
    MarchParams P{};
    int rc = fill_common(P, rays_o, rays_d, viewdirs, n_rays, img_w, grid, mlp, out_rgb, out_depth, out_alphainv, out_counters);
    if (rc) return rc;
    if (!(stepdist > 0.f) || depth_n_samples <= 0 || mlp->spatial_pe != 0) return K4_ERR_BAD_ARG;
    if (mlp->width != 0) {
        const int want = grid->k0_ch - mlp->k0_skip + 3 + 6 * mlp->viewbase_pe;                 // lib/dvgo.py:94-101
        if (mlp->dim0 != want) return K4_ERR_BAD_ARG;
    }
    P.depth_n = depth_n_samples; P.stepdist = stepdist; P.near_ = near; P.far_ = far;
    P.shift = act_shift; P.interval = interval; P.thres = fast_color_thres; P.bg = bg;
    return launch_march<MODE_DVGO>(P, mlp, (hipStream_t)stream);
}
