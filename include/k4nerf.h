/*
 * k4nerf.h -- C ABI of lib4k_hip.so: the MI355X (gfx950) replacement for the native layer of
 * the 4K-NeRF rendering hot path.
 *
 * What it replaces (reference, read-only at /root/reference):
 *   - the pybind11/libtorch extension `render_utils_cuda`, 13 functions taking torch::Tensor
 *     (lib/cuda/render_utils.cpp:170-184; kernels lib/cuda/render_utils_kernel.cu), JIT-built at
 *     import by lib/dvgo.py:14-19 and lib/grid.py:12-17;
 *   - the PyTorch library ops the reference strings between them on the hot path:
 *     F.grid_sample (lib/grid.py:124), torch_scatter.segment_coo (lib/dmpigo.py:382-386,
 *     lib/dvgo.py:415-419), the rgbnet nn.Sequential (lib/dmpigo.py:112-120) and the conv2d /
 *     leaky_relu / interpolate / cat chain of SFTNet (lib/sr_esrnet.py:112-182,446-465).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + explicit sizes, no torch types; the caller allocates every
 *     output (torch.empty on the host side) and owns all memory; the library keeps no state.
 *   - every entry point returns an int: 0 = success, otherwise a hipError_t value (launch errors are
 *     checked with hipGetLastError, which the reference never does -- render_utils_kernel.cu:93 etc.)
 *     or K4_ERR_* below for argument errors.
 *   - the last argument is the HIP stream to launch on (`hipStream_t` passed as void*); the
 *     reference launches on the legacy default stream (render_utils_kernel.cu:93,230,283...).
 *   - all floating point is fp32; index tensors are int64 as in the reference.
 *   - thread-safe / re-entrant: no mutable globals (experiment knobs are environment variables read once at load time;
 *     per-device facts such as the CU count are cached in write-once tables).
 */
#ifndef K4NERF_H
#define K4NERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define K4_OK               0
#define K4_ERR_BAD_ARG      10001   /* null pointer / non-positive size / unsupported combination */
#define K4_ERR_UNSUPPORTED  10002   /* configuration not covered by the fused kernel (use the staged ops) */

#define K4_ABI_VERSION      14      /* 14: k4_sft_train_bwd_gx / k4_sft_train_bwd_rest, k4_rdb_train.aux_stream (the SFT layers' backward split into the chain's grad_x launch and the rest on a third stream), k4_grid_flag_corners / k4_masked_adam_upd_unflagged / k4_masked_adam_upd_sparse_cl_seeded (a grid's masked step in two exact parts); 13: launch tapes (k4_tape_*), k4_add_f32, k4_upsample2x_nhwc / _bwd_nhwc, k4_side_wait_main / k4_main_wait_side, k4_stream_create_overlapping / k4_streams_overlap, K4_CONV_SMALL, k4_rdb_train.no_join / defer_side, k4_sft_train_bwd_side / _main / k4_sft_train_reduce, k4_nhwc_window_to_planar, k4_rgbnet_input_mpi, k4_grid_sample_3d_backward_cl_scatter / _sweep, k4_masked_adam_upd_sparse_cl, k4_joint_losses_fwd / _bwd; 12: round-5 experiments removed (k4_march_workspace_bytes_pre, k4_march_pre_supported, K4_K0_BRICK4, k4_repack_k0_brick4, k4_k0_brick4_floats: profiles/r05_split_path_brick_parts_removed.patch), k4_mlp_desc.arith K4_MLP_ARITH_B2; 11: k4_sft_train_bwd_ex, k4_sft_train_fwd_ex, k4_conv2d_wgrad_dbias_bf16x6_acc, k4_zero_f32, K4_EPI_LRELU_BWD, k4_rdb_train.gc_acc / gx0_add / dwdb_span / fused_lrelu / g5_from_gx0_add, k4_total_variation_add_grad dense_mode 2; 10: k4_train_select_mpi, k4_train_compact, k4_ndc_points_of (training forward with one read-back instead of four); 9: split shading path: k4_march_workspace_bytes_pre, k4_march_pre_supported, K4_K0_BRICK4 + k4_repack_k0_brick4 / k4_k0_brick4_floats; 8: k4_conv3x3_p16_sft_multi, k4_conv_sft_epilogue_bytes, k4_rdb_train_fwd / k4_rdb_train_bwd; 7: pre-split decoder activations: k4_conv3x3_p16_multi, k4_conv_weight_p16_bytes, k4_sft_nhwc_p16_multi, k4_absmax_slice; 6: k4_conv2d_sft_nhwc_bf16x6_multi removed; k4_conv2d_wgrad_dbias_bf16x6, k4_pack_conv_weight_bf16x6_multi, k4_lrelu_bwd, k4_grid_sample_3d_backward_cl, k4_touched_voxels; 5: k4_build_live_mask, k4_sft_train_*, K4_ARITH_F16X3 / k4_conv_weight_f16x3_bytes, no tile_queue, round-1 bf16x3 entry points removed; 4: marcher training entry points (k4_rgbnet_*, k4_distortion_loss); 2: SR / optimizer / ray-generation entry points, k4_mlp_desc.arith; 3: larger marcher workspace (bundle order), k4_sft_nhwc_multi arith, fused conv + SFT entry */
int k4_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Scene description shared by the fused marchers.
 * ------------------------------------------------------------------------------------------- */
#define K4_K0_CHANNEL_MAJOR 0   /* k0 = [C][X][Y][Z]   (checkpoint layout, `k0.grid` [1,C,X,Y,Z]) */
#define K4_K0_CHANNEL_LAST  1   /* k0 = [X][Y][Z][CP]  (load-time repack, CP = C rounded up to 4)  */

typedef struct k4_grid_desc {
    const float*   density;          /* [X][Y][Z] fp32, Z fastest (`density.grid` [1,1,X,Y,Z], lib/grid.py:115) */
    const float*   k0;               /* feature / colour grid, see k0_layout                              */
    const float*   act_shift;        /* MPI only: [act_depth] per-plane bias (`act_shift.grid`, lib/dmpigo.py:48-58); NULL for DVGO */
    const uint8_t* mask;             /* [MX][MY][MZ] bool bytes (`mask_cache.mask`, lib/grid.py:290)    */
    int32_t dims[3];                 /* X, Y, Z of density / k0                                          */
    int32_t k0_ch;                   /* C                                                                */
    int32_t k0_cpad;                 /* CP (channel-last only; multiple of 4)                            */
    int32_t k0_layout;               /* K4_K0_*                                                          */
    int32_t act_depth;               /* MPI: mpi_depth                                                   */
    int32_t mask_dims[3];
    float   xyz_min[3], xyz_max[3];
    float   xyz2ijk_scale[3], xyz2ijk_shift[3];   /* lib/grid.py:291-293 */
    const uint32_t* occ_summary;     /* optional (NULL = none): coarse occupancy summary of `mask` built by
                                        k4_build_occupancy_summary(); lets the geometry kernel skip 16-sample groups of a ray
                                        that cannot touch an occupied voxel.  Results are identical with and without it. */
    int32_t depth_split;             /* MPI, ABI 12: 0, or a multiple of 64 below n_samples: the geometry stage runs DEPTH-ORDERED -- samples
                                        [0, depth_split) in a first launch, then [depth_split, n_samples) only for the rays whose transmittance is
                                        still >= 1e-3 (Alphas2Weights' early stop, render_utils_kernel.cu:597-600: the reference evaluates density
                                        behind the stop and throws it away).  Results are identical with and without it; it pays on scenes of opaque
                                        surfaces (k4_mpi_depth_split_stats proposes a value from the density grid).  Ignored by k4_march_dvgo_fwd. */
} k4_grid_desc;

/* Where would a depth split pay?  Per (x, y) column of an MPI density grid the transmittance of a ray running along z (alpha from
 * density + act_shift, `interval`, as lib/dmpigo.py:316-317) is followed to the plane where it falls below 1e-3; out[b] (b = 1..7) = the share of
 * the grid's alpha > fast_color_thres voxels that lie at or behind plane b * Z / 8 in a column that stopped in front of that plane (what a
 * split at sample b * n_samples / 8 lets the second launch skip, to first order), out[0] = the share of such columns among the columns that
 * hold any such voxel, out[8..15] unused.  Load-time, one launch; the caller reads 16 floats back once per density version. */
int k4_mpi_depth_split_stats(const k4_grid_desc* grid, float interval, float fast_color_thres, float* out16, void* stream);

/* Coarse occupancy summary of a MaskGrid (load-time, like the k0 repack).  Base: per cell (cx, cy) of 8x8 (x, y) voxels (K4_OCC_CELL) and z word w,
 * bit z of the word = OR of mask[x][y][32 w + z] over the cell.  Stored (ABI 12) as FOUR tables [t][ceil(MX/8)][ceil(MY/8)][ceil(MZ/32)] of 64-bit
 * windows: window (cx, cy, w) of table t = (word w | word w+1 << 32) ORed over the cells (cx .. cx + (t & 1), cy .. cy + (t >> 1)) -- the box a group
 * of samples can touch (<= 2 x 2 cells, <= 32 planes) is tested with ONE 8-byte fetch (round 5: eight 4-byte fetches, 31 % of the geometry kernel's
 * wave time).  A group of consecutive samples of a ray touches only voxels inside the per-axis index interval of its two end samples (the index map is
 * monotone along a ray); when every summary bit of that box is clear no sample of the group can pass MaskGrid.forward (lib/grid.py:295-304) and the
 * group is skipped. */
#define K4_OCC_CELL  8
#define K4_OCC_SHIFT 3
int64_t k4_occupancy_summary_bytes(int32_t mx, int32_t my, int32_t mz);
int k4_build_occupancy_summary(const uint8_t* mask, int32_t mx, int32_t my, int32_t mz, uint32_t* out, void* stream);

/* "Live" mask of the fused marchers (load-time, cached by the host per density / act_shift / mask version and per
 * (interval, threshold)): out_mask[v] = mask[v] && (some density cell that a sample rounding to mask voxel v can lie in has
 * raw2alpha(max of its 8 corner densities + the largest act_shift plane value it can see + act_shift_scalar) > fast_color_thres,
 * evaluated with head room for every fp32 rounding of the kernels).  Trilinear interpolation never exceeds its corner maximum, so a
 * sample that out_mask drops is one the reference drops either at MaskGrid.forward (lib/dmpigo.py:308-313, lib/dvgo.py:345-350)
 * or at the alpha > fast_color_thres test that follows it (lib/dmpigo.py:316-323, lib/dvgo.py:353-360): passing out_mask (and its
 * k4_build_occupancy_summary) as k4_grid_desc.mask / .occ_summary leaves every output of k4_march_*_fwd bit-identical, while the
 * density stage stops running on samples that cannot survive it.  (The `mask-pass` counter of out_counters then counts out_mask.)
 *   grid      : density, dims, act_shift / act_depth (MPI; NULL for DVGO), mask, mask_dims, xyz_min/max, xyz2ijk_* are read
 *   workspace : k4_live_mask_workspace_bytes(X, Y, Z) bytes of device scratch (one byte per density cell)
 *   out_mask  : [MX][MY][MZ] bytes.   fast_color_thres must be > 0 (with 0 the reference keeps every sample). */
int64_t k4_live_mask_workspace_bytes(int32_t x, int32_t y, int32_t z);
int k4_build_live_mask(const k4_grid_desc* grid, float act_shift_scalar, float interval, float fast_color_thres,
                       uint8_t* workspace, uint8_t* out_mask, void* stream);

/* Colour MLP `Sequential(Linear, ReLU, [Sequential(Linear, ReLU)] x n_hidden, Linear)`
 * (lib/dmpigo.py:112-120, lib/dvgo.py:116-124), repacked by the host into ONE contiguous fp32 buffer in
 * v_mfma_f32_32x32x2_f32 operand order (k4_mlp_packed_floats() floats; NB = width/32, K1P = dim0+1 rounded
 * up to even -- the extra input is a constant 1 carrying the bias; row(r,h) = (r&3)+8*(r>>2)+4*h is the
 * C/D register->row map of the instruction; l = lane 0..63):
 *     W1A [NB][K1P/2][64]        = W1ext[mb*32+(l&31)][2*kk+(l>>5)],   W1ext = [W1 | b1 | 0]
 *     n_hidden==1:  W2A [NB][NB][16][64] = W2[mb2*32+(l&31)][mb*32+row(r,l>>5)] ;  B2A [NB][64] = l<32 ? b2[mb2*32+l] : 0
 *     WOT [NB][16][2][4]         = Wout[c][mb*32+row(r,h)]  (c = 0..2, 3rd padded with 0)
 *     BO  [4]
 * followed by TWO split sections (bf16 split w = t0 + t1 + t2, each term RNE of the running remainder: exact with 3 terms, 16
 * significant bits with the 2 leading ones; v_mfma_f32_32x32x16_bf16 operand order, 16-byte units = 8 bf16; KB1 = ceil(K1P/16)),
 * first with (NT1, NT2) = (3, 3) for K4_MLP_ARITH_B3, then with (NT1, NT2) = (k4_mlp_b2_layer1_terms(), 2) for the default arithmetic:
 *     W1S [NB][KB1][NT1 terms][64] = W1ext[mb*32+(l&31)][kb*16 + 8*(l>>5) + e]                      e = 0..7
 *     n_hidden==1:  W2S [NB][width/16][NT2][64] = W2[mb2*32+(l&31)][(kb>>1)*32 + (e&3) + 8*(2*(kb&1)+(e>>2)) + 4*(l>>5)]
 *                   B2S [NB][2][16] fp32      = b2[mb2*32 + row(r,h)]
 *     WOT, BO as above
 * width == 0 means "no rgbnet": rgb = sigmoid(k0) with k0_ch == 3 (lib/dvgo.py:377-379). */
typedef struct k4_mlp_desc {
    const float* packed;
    int32_t dim0;                    /* input width, must equal the feature count implied below          */
    int32_t width;                   /* 0 | 32 | 64 | 128                                                */
    int32_t n_hidden;                /* rgbnet_depth - 2: 0 or 1 in the fused path                       */
    int32_t viewbase_pe;             /* #frequencies 2^0..2^(n-1) on viewdirs                            */
    int32_t spatial_pe;              /* MPI only: #frequencies on the normalised position                */
    int32_t k0_skip;                 /* DVGO rgbnet_direct=False: 3 (first 3 k0 channels are added to the logits, lib/dvgo.py:385-386,412), else 0 */
    int32_t arith;                   /* K4_MLP_ARITH_*: matrix-pipe arithmetic of the rgbnet                */
} k4_mlp_desc;
#define K4_MLP_ARITH_DEFAULT 0      /* "b2" (round 6): activations and weights of both rgbnet layers on 2-term bf16 splits (16 significant bits,
                                       3 products a1 w0 + a0 w1 + a0 w0 on v_mfma_f32_32x32x16_bf16, fp32 accumulation); every width incl. 128.
                                       ~130 dB per sample against fp64 (fp32 itself: ~150); >= 100 dB per frame against the fp32 oracle is
                                       asserted (tests/test_march_gpu.py, bench.py parity_vs_oracle)                                        */
#define K4_MLP_ARITH_FP32    1      /* v_mfma_f32_32x32x2_f32: bit-exact fp32 FMA chains (2.7x the matrix-pipe time of B3)                        */
#define K4_MLP_ARITH_B3      2      /* exact 3-term bf16 splits, 6 products (fp32-equivalent; the default of rounds 2-5); width <= 64            */
int64_t k4_mlp_packed_floats(int32_t dim0, int32_t width, int32_t n_hidden);   /* <0: unsupported shape */
int k4_mlp_b2_layer1_terms(void);    /* NT1 of the default section: 2 (3 in A/B builds with -DK4_B2_L1_TERMS=3: layer 1 exact) */

/* ---------------------------------------------------------------------------------------------
 * Fused marchers: two launches (geometry + shading, csrc/k4_march.hip) replace everything inside
 * DirectMPIGO.forward (lib/dmpigo.py:292-427) / DirectVoxGO.forward (lib/dvgo.py:327-448) for the four
 * keys the render loop consumes (run_sr.py:107): rgb_marched (== rgb_feature, they alias in eval), depth,
 * alphainv_last.  No host synchronisation, no per-sample tensor.
 *
 *   rays_o, rays_d, viewdirs : [n_rays][3]
 *   img_w   : 0, or the image width when the n_rays rays are ONE full image in row-major pixel
 *             order (n_rays = H*img_w); only changes the ray->wavefront tiling (8x8 pixel tiles,
 *             XCD-banded), never the results.
 *   workspace : device scratch of >= k4_march_workspace_bytes(n_rays, img_w, max_steps) bytes (the compacted
 *             {ray,step,weight} records between the two kernels; worst-case sized, sparsely touched);
 *             max_steps = n_samples for MPI.  Owned by the caller, reusable across calls on one stream.
 *   out_rgb [n_rays][3], out_depth [n_rays], out_alphainv [n_rays]
 *   out_counters : NULL or uint64[8] (ABI 12; [4] before) = {in-bbox samples, mask-pass samples, alpha-pass samples,
 *                  shaded samples, alpha-pass samples BEHIND their ray's T<1e-3 stop (density evaluated, then dropped by the
 *                  transmittance scan: what a depth-ordered geometry stage could skip), 0, 0, 0}, ACCUMULATED
 *                  (caller zeroes) -- the counts SURVEY 8(d)'s algorithmic-bytes formula needs.
 * ------------------------------------------------------------------------------------------- */
int64_t k4_march_workspace_bytes(int64_t n_rays, int32_t img_w, int32_t max_steps);

int k4_march_mpi_fwd(const float* rays_o, const float* rays_d, const float* viewdirs,
                     int64_t n_rays, int32_t img_w,
                     const k4_grid_desc* grid, const k4_mlp_desc* mlp,
                     int32_t n_samples,      /* int((mpi_depth-1)/stepsize)+1, lib/dmpigo.py:278   */
                     float interval,         /* stepsize * voxel_size_ratio, lib/dmpigo.py:306     */
                     float fast_color_thres, float bg,
                     void* workspace, int64_t workspace_bytes,
                     float* out_rgb, float* out_depth, float* out_alphainv,
                     uint64_t* out_counters, void* stream);

int k4_march_dvgo_fwd(const float* rays_o, const float* rays_d, const float* viewdirs,
                      int64_t n_rays, int32_t img_w,
                      const k4_grid_desc* grid, const k4_mlp_desc* mlp,
                      float near, float far,   /* the reference overrides far with 1e9, lib/dvgo.py:307: pass what the kernel should use */
                      float stepdist,          /* stepsize * voxel_size, lib/dvgo.py:310             */
                      int32_t max_steps,       /* upper bound of samples on one ray: ceil(bbox diagonal / stepdist) + 2 */
                      int32_t depth_n_samples, /* int((max_world_size-1)/stepsize)+1, lib/dvgo.py:311 */
                      float act_shift,         /* scalar buffer, lib/dvgo.py:46                      */
                      float interval,          /* stepsize * voxel_size_ratio, lib/dvgo.py:341       */
                      float fast_color_thres, float bg,
                      void* workspace, int64_t workspace_bytes,
                      float* out_rgb, float* out_depth, float* out_alphainv,
                      uint64_t* out_counters, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Staged ops: one per reference native function, for callers that need the per-sample tensors
 * (training-compatible forward, the `render_utils_cuda` shim).  Same math as the fused kernels.
 * ------------------------------------------------------------------------------------------- */
/* sample_ndc_pts_on_rays (render_utils.cpp:87-98): pts [n_rays][n_samples][3], mask_outbbox [n_rays][n_samples] */
int k4_sample_ndc_pts_on_rays(const float* rays_o, const float* rays_d, const float* xyz_min, const float* xyz_max,
                              int64_t n_rays, int32_t n_samples, float* out_pts, uint8_t* out_mask_outbbox, void* stream);
/* Training forward of DirectMPIGO without host round trips per filter (round 5; lib/dmpigo.py:300-333 filters the sample list by
 * bounding box + mask cache, by alpha > fast_color_thres and by weight > fast_color_thres: three boolean-mask indexings = device-to-host
 * synchronisations).  k4_train_select_mpi takes the three decisions for all n_rays x n_samples samples in one launch, in the arithmetic of
 * the staged ops above (sampler, maskcache_lookup, grid_sample of the density and act_shift grids, raw2alpha with shift 0, the sequential
 * transmittance product of alpha2weight incl. its T < 1e-3 stop):
 *   steps2 [n_rays][n_samples] int16 : per ray, the steps of its alpha-passing samples, ascending (first cnt2[ray] entries)
 *   keep3  [n_rays][n_samples] uint8 : per such sample, 1 when its weight passes too
 *   cnt2, cnt3 [n_rays] int64        : the counts
 * The caller forms the inclusive cumsums of cnt2 / cnt3, reads the two totals back (one synchronisation, to size the tensors), and
 * k4_train_compact writes ray_id / step_id [total2] of the alpha-passing samples and idx3 [total3], the positions of the shaded ones in
 * that list; k4_ndc_points_of recomputes their points (render_utils_kernel.cu:260 on listed samples).  The differentiable ops then run on
 * those lists exactly as the reference runs them on its filtered tensors. */
int k4_train_select_mpi(const float* rays_o, const float* rays_d, const float* xyz_min, const float* xyz_max, int64_t n_rays, int32_t n_samples,
                        const uint8_t* mask, const float* xyz2ijk_scale, const float* xyz2ijk_shift, int32_t mi, int32_t mj, int32_t mk,
                        const float* density, int32_t x, int32_t y, int32_t z, const float* act_shift, int32_t act_depth,
                        float interval, float fast_color_thres,
                        int16_t* steps2, uint8_t* keep3, int64_t* cnt2, int64_t* cnt3, void* stream);
int k4_train_compact(const int16_t* steps2, const uint8_t* keep3, const int64_t* cnt2, const int64_t* cumsum2, const int64_t* cumsum3,
                     int64_t n_rays, int32_t n_samples, int64_t* ray_id, int64_t* step_id, int64_t* idx3, void* stream);
int k4_ndc_points_of(const float* rays_o, const float* rays_d, const int64_t* ray_id, const int64_t* step_id, int64_t n, int32_t n_samples,
                     float* pts, void* stream);
/* infer_t_minmax (render_utils.cpp:50-58) */
int k4_infer_t_minmax(const float* rays_o, const float* rays_d, const float* xyz_min, const float* xyz_max,
                      float near, float far, int64_t n_rays, float* out_t_min, float* out_t_max, void* stream);
/* infer_n_samples (render_utils.cpp:60-65) */
int k4_infer_n_samples(const float* rays_d, const float* t_min, const float* t_max, float stepdist,
                       int64_t n_rays, int64_t* out_n_samples, void* stream);
/* infer_ray_start_dir (render_utils.cpp:67-72) */
int k4_infer_ray_start_dir(const float* rays_o, const float* rays_d, const float* t_min, int64_t n_rays,
                           float* out_start, float* out_dir, void* stream);
/* sample_pts_on_rays (render_utils.cpp:74-85) in two phases because M = sum(N_steps) is data dependent
 * (the reference syncs with .item(), render_utils_kernel.cu:212):
 *   count: t_min, t_max, N_steps per ray;  the host takes cumsum(N_steps) and M;
 *   fill : ray_id/step_id/pts/mask for all M points from the inclusive cumsum. */
int k4_sample_pts_on_rays_count(const float* rays_o, const float* rays_d, const float* xyz_min, const float* xyz_max,
                                float near, float far, float stepdist, int64_t n_rays,
                                int64_t* out_n_steps, float* out_t_min, float* out_t_max, void* stream);
int k4_sample_pts_on_rays_fill(const float* rays_o, const float* rays_d, const float* xyz_min, const float* xyz_max,
                               const float* t_min, const int64_t* n_steps_cumsum, float stepdist,
                               int64_t n_rays, int64_t total_len,
                               float* out_pts, uint8_t* out_mask_outbbox, int64_t* out_ray_id, int64_t* out_step_id,
                               void* stream);
/* maskcache_lookup (render_utils.cpp:109-118) */
int k4_maskcache_lookup(const uint8_t* world, const float* xyz, const float* xyz2ijk_scale, const float* xyz2ijk_shift,
                        int32_t sz_i, int32_t sz_j, int32_t sz_k, int64_t n_pts, uint8_t* out, void* stream);
/* raw2alpha / raw2alpha_backward (render_utils.cpp:120-135); interval_pp != NULL -> the _nonuni variants (:125-140) */
int k4_raw2alpha(const float* density, float shift, float interval, const float* interval_pp, int64_t n_pts,
                 float* out_exp, float* out_alpha, void* stream);
int k4_raw2alpha_backward(const float* exp_d, const float* grad_back, float interval, const float* interval_pp,
                          int64_t n_pts, float* out_grad, void* stream);
/* alpha2weight (render_utils.cpp:142-149): weight/T are fully written (0 / 1 past the early stop) */
int k4_alpha2weight(const float* alpha, const int64_t* ray_id, int64_t n_pts, int64_t n_rays,
                    float* out_weight, float* out_T, float* out_alphainv_last,
                    int64_t* out_i_start, int64_t* out_i_end, void* stream);
/* alpha2weight_backward (render_utils.cpp:151-167) */
int k4_alpha2weight_backward(const float* alpha, const float* weight, const float* T, const float* alphainv_last,
                             const int64_t* i_start, const int64_t* i_end, int64_t n_rays, int64_t n_pts,
                             const float* grad_weights, const float* grad_last, float* out_grad, void* stream);
/* DenseGrid.forward = F.grid_sample(bilinear, align_corners=True, zero padding) on [C][X][Y][Z] (lib/grid.py:117-128):
 * out [n_pts][C] */
int k4_grid_sample_3d(const float* grid, int32_t channels, int32_t X, int32_t Y, int32_t Z,
                      const float* xyz, const float* xyz_min, const float* xyz_max, int64_t n_pts,
                      float* out, void* stream);
/* segment_coo(src, index, out=zeros, reduce='sum') for a SORTED index (lib/dmpigo.py:382-386): out [n_seg][C] */
int k4_segment_sum(const float* src, const int64_t* index, int64_t n_pts, int32_t channels, int64_t n_seg,
                   float* out, void* stream);
/* Backward of the two library ops above, as autograd runs them in the reference's training step:
 * grad_grid [C][X][Y][Z] += scatter of grad_out [n_pts][C] with the forward's trilinear weights (grid_sampler_3d_backward,
 * grad wrt input; fp32 hardware atomics, caller zero-fills or accumulates); grad_src [n_pts][C] = grad_out[index[i]]. */
int k4_grid_sample_3d_backward(const float* grad_out, int32_t channels, int32_t X, int32_t Y, int32_t Z,
                               const float* xyz, const float* xyz_min, const float* xyz_max, int64_t n_pts,
                               float* grad_grid, void* stream);
/* The same gradient for channels > 1 through a channel-last scratch image (lanes = (sample, channel): a corner's contributions are
 * consecutive floats, which is what the atomic units are fast at -- 2x on ray-coherent batches, ~10x on random points) and a sweep
 * that moves the touched voxels into grad_grid [C][X][Y][Z] (+=).  `workspace` (k4_grid_sample_3d_backward_workspace_bytes, 16-byte
 * aligned) must be ALL ZERO on entry and is all zero again on return: allocate and clear it once, not per call. */
int64_t k4_grid_sample_3d_backward_workspace_bytes(int32_t channels, int32_t X, int32_t Y, int32_t Z);      /* < 0: use the plain entry */
int k4_grid_sample_3d_backward_cl(const float* grad_out, int32_t channels, int32_t X, int32_t Y, int32_t Z,
                                  const float* xyz, const float* xyz_min, const float* xyz_max, int64_t n_pts,
                                  float* grad_grid, void* workspace, void* stream);
/* Its two halves: the scatter alone leaves the touched voxels' sums in the workspace (scratch[voxel][channels] + one flag byte per voxel behind it);
 * the sweep moves them into grad_grid and clears the workspace.  Between the two a caller may consume the sums where they lie:
 * k4_masked_adam_upd_sparse_cl (below, optimizer section) is MaskedAdam's masked update over exactly those voxels. */
int k4_grid_sample_3d_backward_cl_scatter(const float* grad_out, int32_t channels, int32_t X, int32_t Y, int32_t Z,
                                          const float* xyz, const float* xyz_min, const float* xyz_max, int64_t n_pts,
                                          void* workspace, void* stream);
int k4_grid_sample_3d_backward_cl_sweep(int32_t channels, int32_t X, int32_t Y, int32_t Z, void* workspace, float* grad_grid, void* stream);
int k4_segment_sum_backward(const float* grad_out, const int64_t* index, int64_t n_pts, int32_t channels,
                            float* grad_src, void* stream);
/* get_rays_of_a_view (lib/dvgo.py:516-582: get_rays + viewdirs + ndc_rays with near = 1) in one launch.  K_dev: [3][3]
 * intrinsics, c2w_dev: [3][4] (or the top of a [4][4]) camera-to-world, both fp32 ON THE DEVICE; focal = K[0][0] as a
 * host value (the reference passes it as a Python scalar); mode_center: 1 = pixel centres (+0.5), 0 = 'lefttop'.
 * Outputs [H*W][3] each.  k4_to8b: utils.to8b (lib/utils.py:19), n floats -> n bytes. */
int k4_get_rays_of_a_view(int32_t H, int32_t W, const float* K_dev, const float* c2w_dev, int32_t ndc,
                          int32_t inverse_y, int32_t flip_x, int32_t flip_y, int32_t mode_center, float focal,
                          float* rays_o, float* rays_d, float* viewdirs, void* stream);
int k4_to8b(const float* x, int64_t n, uint8_t* out, void* stream);
/* The interior of a decoded window into the frame (SFTNet.tile_process, lib/sr_esrnet.py:508-524: output_tile[..., crop] -> output[..., tile]) in ONE pass:
 * dst[c * dst_plane_stride + y * dst_row_stride + x] = src[((oy + y) * src_w + ox + x) * channels + c], 0 <= y < th, 0 <= x < tw, c < channels (<= 4);
 * src: the window's NHWC result [*][src_w][channels], dst: planes of the [1, 3, 4H, 4W] frame (pre-offset to the tile) or of a gather buffer. */
int k4_nhwc_window_to_planar(const float* src, int32_t src_w, int32_t channels, int32_t oy, int32_t ox, int32_t th, int32_t tw,
                             float* dst, int64_t dst_plane_stride, int64_t dst_row_stride, void* stream);
/* load-time repack of `k0.grid` [C][X][Y][Z] -> [X][Y][Z][CP] (zero padded channels) */
int k4_repack_k0(const float* k0_cmajor, int32_t channels, int32_t cpad, int64_t n_voxels, float* out, void* stream);

/* Occupancy / resolution maintenance of the training loop (lib/dmpigo.py:189-226, lib/dvgo.py:200-233):
 *   k4_resample_trilinear : DenseGrid.scale_volume_grid (lib/grid.py:130-135) = F.interpolate(trilinear, align_corners=True) of a
 *                           [C][X][Y][Z] grid to [C][X2][Y2][Z2]
 *   k4_alpha_maxpool3_gt  : (F.max_pool3d(alpha, 3, padding=1, stride=1) > thres) as one byte per voxel -- the occupancy refresh of
 *                           update_occupancy_cache / scale_volume_grid */
int k4_resample_trilinear(const float* in, int32_t channels, int32_t x, int32_t y, int32_t z,
                          float* out, int32_t x2, int32_t y2, int32_t z2, void* stream);
int k4_alpha_maxpool3_gt(const float* alpha, int32_t x, int32_t y, int32_t z, float thres, uint8_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Super-resolution decoder (SFTNet "VC-Decoder", lib/sr_esrnet.py:400-465): NHWC fp32 activations, fp32 MFMA
 * implicit GEMM (csrc/k4_sr.hip).  Replaces the conv2d / leaky_relu / torch.cat / F.interpolate chain of
 * SFTLayer (:112-123), ResidualDenseBlock_SFT (:149-158), RRDB_SFT (:176-182) and SFTNet.forward (:446-465).
 * ------------------------------------------------------------------------------------------- */
#define K4_EPI_LRELU       2u    /* y = y > 0 ? y : slope*y                        (after bias / modulation)   */
#define K4_EPI_RES         4u    /* y = y*res_scale + res[pix][co]                 (after the activation)      */
#define K4_EPI_MODULATE    8u    /* SFTLayer tail: the GEMM yields 2*cout channels [scale | shift];
                                    y[co] = mod_x[pix][co]*(scale[co]+1) + shift[co]   (lib/sr_esrnet.py:123)   */
#define K4_EPI_LRELU_BWD 256u    /* k4_conv2d_nhwc_bf16x6(_multi), 3x3 layers with cout % 32 == 0: after everything else the LAST 32 output channels are
                                    multiplied by (mod_x[pix][co] > 0 ? 1 : slope), mod_x = [H][W][mod_stride] addressed with the OUTPUT's channel index:
                                    LeakyReLU backward (k4_lrelu_bwd) of the gradient slice a dgrad accumulation has just completed, in its epilogue */
#define K4_CONV_SMALL    512u    /* k4_conv2d_nhwc_bf16x6, plain 3x3 layers (no K4_ARITH_* / K4_PRE_UPSAMPLE2X), one window: when the image is small (one-row
                                    workgroups <= 4 per CU: the 64x64 training patch) the layer runs on the K-split kernel -- a workgroup is one row x 32
                                    pixels x 32 output channels, its four waves split the input-channel chunks.  Same products; the fp32 additions are
                                    ordered differently than in the row kernels (not bit-identical to them).  Training graph only. */
#define K4_W_TAPS_AS_COUT 32u    /* k4_conv2d_nhwc_bf16x6 only, 3x3 with cout <= 3: w_split holds the 1x1 layer [9*cout -> 32][cin]
                                    (n = tap*cout + co) in the bf16x6 layout; the kernel sums the 9 taps from LDS        */
#define K4_PRE_UPSAMPLE2X 16u    /* the input is read through a nearest x2 upsample (lib/sr_esrnet.py:461-463)  */
#define K4_ARITH_2TERM    64u    /* k4_conv2d_nhwc_bf16x6(_multi), plain 3x3 layers: use the two leading split terms of both operands only -- 3 of the
                                    6 products (a1 b0 + a0 b1 + a0 b0), ~2^-16 relative per product: the decoder's opt-in 'bf16x3' arithmetic on the
                                    default kernel (1x1 layers, K4_W_TAPS_AS_COUT and the fused-SFT entry ignore / reject it) */
#define K4_ARITH_F16X3   128u    /* k4_conv2d_nhwc_bf16x6(_multi), plain 3x3 layers with cout > 3: 2-term splits in FP16 (22 significant bits per operand), 3
                                    products on v_mfma_f32_32x32x16_f16, ~2^-21 relative per product; w_split is then the k4_conv_weight_f16x3_bytes()
                                    buffer: [ceil(cin/16)][hi|lo][9][2 channel groups][32*NT][8] fp16 of w[co][ci] * 2^a[co] * 2^b[ci/16] (a: largest
                                    |w| of the output channel -> [2^13, 2^14); b >= 0: largest scaled |w| of the chunk -> [2^13, 2^14)), followed by
                                    [32*NT] fp32 2^-a[co] and [ceil(cin/16)] int32 b (padded to 16 bytes).  Activations are scaled per staged chunk
                                    inside the kernel. */

/* stride-1 "same" (zero padded) 3x3 or 1x1 convolution, NHWC:
 *   x        : [H_in][W_in][cin_stride], channels [0,cin) are read (pre-offset the pointer for a slice);
 *              H_in,W_in = H,W or H/2,W/2 with K4_PRE_UPSAMPLE2X
 *   w_packed : k4_conv_weight_floats() floats = [ceil(cin/8)][ksize*ksize][8][32*NT] (input-channel chunk, tap,
 *              channel in chunk, output channel; zero padded), NT = ceil(N/32), N = cout (2*cout with MODULATE)
 *   bias     : [32*NT] (zero padded)
 *   y        : [H][W][cout_stride], channels [0,cout) are written (pre-offset the pointer for a slice)
 *   res      : [H][W][res_stride] or NULL;  mod_x : [H][W][mod_stride] or NULL (may alias y)
 * Supported N tiles: 3x3: N <= 64;  1x1: N <= 128. */
int64_t k4_conv_weight_floats(int32_t cout, int32_t cin, int32_t ksize);
int k4_conv2d_nhwc(const float* x, int32_t cin, int32_t cin_stride,
                   const float* w_packed, const float* bias, int32_t ksize,
                   float* y, int32_t cout, int32_t cout_stride,
                   int32_t H, int32_t W, uint32_t flags, float slope,
                   const float* res, int32_t res_stride, float res_scale,
                   const float* mod_x, int32_t mod_stride, void* stream);

/* 3-term split ("bf16x6": the strictly fp32-equivalent decoder arithmetic; the decoder's default is "f16x3", K4_ARITH_F16X3 below): same contract as k4_conv2d_nhwc, fp32-equivalent results.
 * x = x0 + x1 + x2 exactly (bf16 terms), 6 of the 9 partial products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation;
 * the dropped terms are <= 2^-23 |x w| per product.  w_split : k4_conv_weight_bf16x6_bytes() bytes =
 * [ceil(cin/16)][3 terms][ksize*ksize][2 channel groups][32*NT][8] bf16 (zero padded, term t = RNE_bf16 of the remainder
 * after terms < t).  Replaces the same nn.Conv2d calls of lib/sr_esrnet.py:446-465. */
int64_t k4_conv_weight_bf16x6_bytes(int32_t cout, int32_t cin, int32_t ksize);
int64_t k4_conv_weight_f16x3_bytes(int32_t cout, int32_t cin, int32_t ksize);      /* K4_ARITH_F16X3 operand; < 0: layer shape not covered */
int k4_conv2d_nhwc_bf16x6(const float* x, int32_t cin, int32_t cin_stride,
                          const void* w_split, const float* bias, int32_t ksize,
                          float* y, int32_t cout, int32_t cout_stride,
                          int32_t H, int32_t W, uint32_t flags, float slope,
                          const float* res, int32_t res_stride, float res_scale,
                          const float* mod_x, int32_t mod_stride, void* stream);

/* Grouped launches: the windows (tiles) of SFTNet.tile_process (lib/sr_esrnet.py:482-526) are independent images sharing every
 * weight; `_multi` runs one layer of up to K4_MAX_JOBS windows as ONE launch (their workgroups fill the chip together).  `jobs` is
 * a HOST array; everything not in the job struct is shared and has the meaning documented for the single-window entry points,
 * which are the n_jobs == 1 case of these. */
#define K4_MAX_JOBS 8
typedef struct k4_conv_job { const float* x; float* y; const float* res; const float* mod_x; int32_t H, W; } k4_conv_job;
int k4_conv2d_nhwc_bf16x6_multi(const k4_conv_job* jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                                const void* w_split, const float* bias, int32_t ksize, int32_t cout, int32_t cout_stride,
                                uint32_t flags, float slope, int32_t res_stride, float res_scale, int32_t mod_stride,
                                void* stream);
typedef struct k4_sft_job { const float* cond; const float* x; float* y; const float* res; int64_t n_pix; } k4_sft_job;
#define K4_SFT_ARITH_FP32   0      /* v_mfma_f32_32x32x2_f32: exact fp32 FMA chains (what k4_sft_nhwc computes)                    */
#define K4_SFT_ARITH_BF16X6 1      /* exact 3-term bf16 splits, 6 partial products on v_mfma_f32_32x32x16_bf16 (fp32-equivalent)    */
int k4_sft_nhwc_multi(const k4_sft_job* jobs, int32_t n_jobs, int32_t cond_stride, const float* w_packed, int32_t x_stride,
                      int32_t y_stride, int32_t channels, float slope, int32_t res_stride, float res_scale, int32_t arith, void* stream);

/* ---- PRE-SPLIT activations ("p16"; the decoder's 'f16x3p' arithmetic = K4_ARITH_F16X3's three fp16 products with the split done ONCE by
 * the producing layer) -----------------------------------------------------------------------------------------------------------------------
 * A p16 tensor is a channel slice (offset and width multiples of 16 channels; 16-byte aligned rows) of an NHWC image of 4-byte elements.
 * Per pixel and 16-channel chunk (64 bytes) it holds four 16-byte units
 *     [hi ch 0-7][hi ch 8-15][lo ch 0-7][lo ch 8-15],   hi = RNE_fp16(v 2^E), lo = RNE_fp16(v 2^E - hi)
 * under ONE exponent E per tensor, fixed by the caller before the tensor is written (22 significant bits while |v| 2^E >= 2^-3, an
 * absolute error <= 2^-25 2^-E below).  Producers: k4_sft_nhwc_p16_multi and k4_conv3x3_p16_multi with out_scale = 2^E; each ORs 1 into
 * overflow[job] when a value of that window does not fit fp16 (|v| 2^E > 65504, or non-finite): the caller then re-evaluates the window on
 * the fp32-activation entry points (SFTNet does).  Consumer: k4_conv3x3_p16_multi (3x3, stride 1, zero padding; cin % 16 == 0,
 * cout % 32 == 0; flags from {K4_EPI_LRELU (0 <= slope <= 1), K4_EPI_RES, K4_PRE_UPSAMPLE2X}); out_scale == 0 writes plain fp32.
 *   w_p16 : k4_conv_weight_p16_bytes(cout, cin) bytes = [cin/16][cout/32][hi|lo][9 taps][2 channel groups][32 co][8] fp16 of
 *           w[co][ci][tap] 2^a[co] 2^-E[ci] (E[ci] = exponent of the p16 tensor input channel ci belongs to; a[co] brings the largest scaled
 *           magnitude of output channel co into [2^13, 2^14)), followed by [cout] floats 2^-a[co]; 16-byte aligned.  bias: [cout] floats,
 *           16-byte aligned.  The result does not depend on which other windows share the launch.
 *   K4_PRE_UPSAMPLE2X: the input image is (H/2) x (W/2) and is read through a nearest x2 upsampling (lib/sr_esrnet.py:461-463).  The entry point
 *           evaluates the layer per output PHASE (py, px) = (y & 1, x & 1) as a 2 x 2 convolution of the input: the 3 x 3 taps that land on the
 *           same input pixel are added by the packer (fp32) -- 2.25x fewer matrix instructions, the same sums up to one rounding of a weight sum.
 *           w_p16 then is k4_conv_weight_p16_up2x_bytes(cout, cin) bytes = [cin/16][cout/32][phase py*2+px][hi|lo][tap a*2+b][2][32][8] fp16 +
 *           [cout] floats 2^-a[co]; tap (a, b) of phase (py, px) multiplies input pixel (Y - 1 + py + a, X - 1 + px + b), output pixel (2Y + py, 2X + px).
 * k4_absmax_slice: *out_bits = max(*out_bits, bits of the largest |x| of the channel slice) (atomicMax; zero it first) -- what a
 * calibration pass on the fp32 entry points uses to choose the exponents. */
int64_t k4_conv_weight_p16_bytes(int32_t cout, int32_t cin);
int64_t k4_conv_weight_p16_up2x_bytes(int32_t cout, int32_t cin);
int k4_conv3x3_p16_multi(const k4_conv_job* jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                         const void* w_p16, const float* bias, int32_t cout, int32_t cout_stride,
                         uint32_t flags, float slope, int32_t res_stride, float res_scale,
                         float out_scale, uint32_t* overflow, void* stream);
int k4_sft_nhwc_p16_multi(const k4_sft_job* jobs, int32_t n_jobs, int32_t cond_stride, const float* w_packed, int32_t x_stride,
                          int32_t y_stride, int32_t channels, float slope, float out_scale, uint32_t* overflow, void* stream);

/* k4_conv3x3_p16_multi whose result v goes through the SFTLayer that CONSUMES it (lib/sr_esrnet.py:112-123; in a dense block sft1 after conv4,
 * lib/sr_esrnet.py:154-155, and the next block's sft0 after conv5, :156 + :150) in the epilogue instead of a launch of its own:
 *     y  (fp32, optional: jobs[g].y may be NULL) = v = the layer's result with its flags (K4_EPI_LRELU / K4_EPI_RES)
 *     y2 (p16 under out_scale = 2^E)             = v * (scale(cond) + 1) + shift(cond)
 * cond: [H W][cond_stride] fp32 (32 channels read); scale / shift = conv1x1(lrelu_{sft_slope}(conv1x1(cond))) as 32 -> 32 -> cout stacks in this
 * kernel's arithmetic (fp16 hi/lo splits, three products, fp32 accumulation).  cond_scale = 2^Ec, a power of two with |cond| 2^Ec < 2^10 over
 * the range the operand was packed for; a condition value with |cond| 2^Ec > 65504 (or a stored value beyond fp16) ORs 1 into overflow[job].
 *   w_sfe: k4_conv_sft_epilogue_bytes(cout) bytes = per 32-channel output block nb, 17408 bytes:
 *     A1 [path scale|shift][kb 2][hi|lo][64 lanes][8] fp16 : W0_path[m = lane & 31][k = 16 kb + 8 (lane >> 5) + e] 2^a1[m] 2^-Ec
 *     A2 [path][kb][hi|lo][64][8] fp16 : W1_path[32 nb + (lane & 31)][j(kb, lane >> 5, e)] 2^a2[c] 2^-Eh[j],
 *                                        j(kb, h, e) = (e & 3) + 8 (2 kb + (e >> 2)) + 4 h   (GEMM 1's accumulator-register order)
 *     tables [us1 | b1 | us2 | b2][path][half][16] fp32, entry i of half h <-> row (i & 3) + 8 (i >> 2) + 4 h:
 *            us1 = 2^-a1 2^Eh, b1 = bias0 2^Eh (hidden neuron), us2 = 2^-a2, b2 = bias1 (output channel 32 nb + row)
 *   with 2^Eh[j] chosen from the bound sum_k |W0[j][k]| 2^(10 - Ec) + |bias0[j]| <= 2^(9 - Eh[j]): the hidden activations cannot leave fp16
 *   while the condition passes its check.  Same window independence as k4_conv3x3_p16_multi. */
typedef struct k4_conv_sft_job { const float* cond; void* y2; } k4_conv_sft_job;
int64_t k4_conv_sft_epilogue_bytes(int32_t channels);
int k4_conv3x3_p16_sft_multi(const k4_conv_job* jobs, const k4_conv_sft_job* sft_jobs, int32_t n_jobs, int32_t cin, int32_t cin_stride,
                             const void* w_p16, const float* bias, int32_t cout, int32_t cout_stride,
                             uint32_t flags, float slope, int32_t res_stride, float res_scale,
                             int32_t cond_stride, float cond_scale, const void* w_sfe, float sft_slope, int32_t y2_stride,
                             float out_scale, uint32_t* overflow, void* stream);
int k4_absmax_slice(const float* x, int64_t n_pix, int32_t stride, int32_t channels, uint32_t* out_bits, void* stream);

/* Fused SFTLayer (lib/sr_esrnet.py:112-123): y[p][c] = x[p][c]*(scale(cond)[p][c]+1) + shift(cond)[p][c] (then
 * *res_scale + res if res != NULL), scale/shift = conv1x1(lrelu(conv1x1(cond))) evaluated in one launch, the hidden
 * activations stay in registers.  cond: [n_pix][cond_stride] (32 channels, 16-B aligned rows); channels = 32 or 64;
 * w_packed: k4_sft_weight_floats(channels) floats = WA [2][17][64] | WS [C/32][17][64] | WH [C/32][17][64] in
 * v_mfma_f32_32x32x2_f32 operand order (k-step 16 carries the bias), followed by the split-bf16 section of the same weights that
 * K4_SFT_ARITH_BF16X6 reads (layout: csrc/k4_sr.hip, k4_sft_b6_kernel; host packer sr_esrnet.pack_sft); y may alias x. */
int64_t k4_sft_weight_floats(int32_t channels);
int k4_sft_nhwc(const float* cond, int32_t cond_stride, const float* w_packed,
                const float* x, int32_t x_stride, float* y, int32_t y_stride, int32_t channels,
                int64_t n_pix, float slope, const float* res, int32_t res_stride, float res_scale, void* stream);

/* ---- decoder backward (run_sr.py:869-1014 back-propagates through SFTNet) ----------------------------------
 * dgrad needs no entry point of its own: dX = conv(dY, W') with W'[ci][co][2-dy][2-dx] = W[co][ci][dy][dx] runs on
 * k4_conv2d_nhwc_bf16x6 with the host-packed W' (3x3: any number of 32-channel output blocks).
 *   k4_conv2d_wgrad_bf16x6 : dW[co][ci][dy][dx] = sum_p dY[p][co] * X[p + (dy-pad, dx-pad)][ci]  (zero padded), MFMA GEMM over the
 *                            pixels with exact 3-term bf16 splits of both operands; `dw` ([cout][cin][k][k], PyTorch layout) is
 *                            OVERWRITTEN: zeroed on the stream, then the split-K partial sums are added with fp32 atomics
 *   k4_conv2d_bias_grad    : dbias[co] = sum_p dY[p][co]   (dbias is overwritten; pixel slabs are summed with fp32 atomics) */
int k4_conv2d_wgrad_bf16x6(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                           int32_t ksize, int32_t H, int32_t W, float* dw, void* stream);
int k4_conv2d_bias_grad(const float* gy, int32_t cout, int32_t gy_stride, int64_t n_pix, float* dbias, void* stream);
/* Both gradients of a biased layer in one zero-fill + ONE launch (the workgroups of tap 0 / input block 0 also sum dY): dw_db is one
 * buffer [cout*cin*k*k floats of dW | cout floats of dbias], overwritten. */
int k4_conv2d_wgrad_dbias_bf16x6(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                                 int32_t ksize, int32_t H, int32_t W, float* dw_db, void* stream);
/* ... ADDED to dw_db, which the caller has zeroed (k4_zero_f32): several layers' buffers zeroed by one launch */
int k4_conv2d_wgrad_dbias_bf16x6_acc(const float* x, int32_t cin, int32_t x_stride, const float* gy, int32_t cout, int32_t gy_stride,
                                     int32_t ksize, int32_t H, int32_t W, float* dw_db, void* stream);
int k4_zero_f32(float* p, int64_t n, void* stream);
/* Device-side weight packing for the training loop (every optimizer step changes every weight: 2 x 260 packings per iteration).
 * Writes the `w_split` operand of k4_conv2d_nhwc_bf16x6 for the nn.Conv2d weight w [cout][cin][k][k], bit-identical to the host packer:
 *   form 0: the layer as stored (k4_conv_weight_bf16x6_bytes(cout, cin, k) bytes; bias_out [ceil(cout/32)*32] = bias, zero padded)
 *   form 1: the dgrad operand W'[ci][co][2-dy][2-dx] = W[co][ci][dy][dx], a layer cout -> cin (k4_conv_weight_bf16x6_bytes(cin, cout, k)
 *           bytes; bias_out [ceil(cin/32)*32] = 0)
 *   form 2: K4_W_TAPS_AS_COUT (3x3, cout <= 3): the 1x1 layer [9*cout -> 32][cin] (k4_conv_weight_bf16x6_bytes(9*cout, cin, 1) bytes;
 *           bias_out [32] = bias)
 *   form 3: form 2 of the dgrad operand (3x3, cin <= 3: a layer cout -> cin with <= 3 outputs; k4_conv_weight_bf16x6_bytes(9*cin, cout, 1)
 *           bytes; bias_out [32] = 0).   bias == NULL reads as zeros. */
int k4_pack_conv_weight_bf16x6(const float* w, const float* bias, int32_t cout, int32_t cin, int32_t ksize, int32_t form,
                               void* w_split, float* bias_out, void* stream);
/* The same for many layers: ceil(n_jobs / K4_PACK_MULTI_MAX) launches instead of n_jobs (a training iteration re-packs every layer). */
#define K4_PACK_MULTI_MAX 64
typedef struct k4_pack_job {
    const float* w; const float* bias;       /* as k4_pack_conv_weight_bf16x6 */
    void* w_split; float* bias_out;
    int32_t cout, cin, ksize, form;
} k4_pack_job;
int k4_pack_conv_weight_bf16x6_multi(const k4_pack_job* jobs, int32_t n_jobs, void* stream);
/* out[p][c] = grad[p][c] * (y[p][c] > 0 ? 1 : slope) for c < channels: LeakyReLU backward from the layer's OUTPUT y (same sign as its
 * input); rows of `*_stride` floats (channel slices of wider images); out may be grad (in place). */
int k4_lrelu_bwd(const float* grad, int32_t g_stride, const float* y, int32_t y_stride, int64_t n_pix, int32_t channels, float slope,
                 float* out, int32_t out_stride, void* stream);

/* A whole ResidualDenseBlock_SFT (lib/sr_esrnet.py:126-158) of the training graph issued by ONE call: the same launches, in the same order, as the
 * host issues through k4_sft_train_fwd / k4_conv2d_nhwc_bf16x6 / k4_conv2d_wgrad_dbias_bf16x6 / k4_lrelu_bwd / k4_sft_train_bwd (7 forward, 19 backward) --
 * the joint training iteration is paced by the host's ~900 Python-to-C calls (profiles/r04_train_wgrad_side_stream_slower.md), this removes ~390 of them.
 * Images are NHWC fp32; g = 32 growth channels, nf = 32 | 64; bw = nf + 4 g.
 *   forward : buf[:, 0:nf] = sft0(t, c); buf[:, nf+(k-1)g : nf+kg] = lrelu(conv_k(buf[:, 0:nf+(k-1)g])), k = 1..3; x4 = lrelu(conv4(buf[:, 0:nf+3g]));
 *             buf[:, nf+3g : bw] = sft1(x4, c); out = 0.2 conv5(buf) + t.
 *   backward: g5 = 0.2 grad_out (caller); G = dgrad5(g5); the SFT / LeakyReLU / dgrad chain accumulates into the channel prefixes of G as the host form
 *             does (lib/sr_train.py K4RDB); gx0 / gc0 / gc1 = gradients of t through sft0 and of the condition through both SFT layers; dwdb[k] =
 *             [dW | dbias] of conv k+1; gsft0 / gsft1 = the eight weight / bias gradients of each SFT layer.  side_stream != NULL: every weight-gradient
 *             launch goes to that stream, forked behind the launch that finishes the gradient slice it reads and joined before the call returns control
 *             of `stream` (the wgrads depend on nothing the chain produces later; a 64x64 patch's launches leave most of the chip idle).
 * w_fwd / b_fwd: forward operands of conv1..conv5 (k4_pack_conv_weight_bf16x6 form 0); w_bwd / b_bwd: dgrad operands (form 1, zero bias). */
typedef struct k4_rdb_train {
    int32_t H, W, nf, g;
    const float* t; const float* c;
    float* buf; float* x4; float* out;
    const void* w_fwd[5]; const float* b_fwd[5];
    const float* sft0[8]; const float* sft1[8];         /* w0s b0s w1s b1s w0h b0h w1h b1h */
    const void* w_bwd[5]; const float* b_bwd[5];
    const float* g5;
    float* G; float* gx4; float* gx0; float* gc0; float* gc1;
    float* dwdb[5];
    float* gsft0[8]; float* gsft1[8];
    float* ws0; int64_t ws0_bytes; float* ws1; int64_t ws1_bytes;
    void* side_stream;
    /* optional (all NULL / 0 = the behaviour above): */
    float* gc_acc;                  /* both SFT layers ADD their condition gradient into this [n_pix][32] buffer (gc0 / gc1 unused): every SFT layer of the
                                       decoder reads the same condition map, the sum over layers needs no kernels of its own */
    const float* gx0_add;           /* gx0 = gradient of t through sft0 + gx0_add [n_pix][nf] (the block's skip connection: grad_out) */
    float* dwdb_span; int64_t dwdb_span_floats;   /* the five dwdb buffers lie in this one span: ONE zero-fill for all of them */
    int32_t fused_lrelu;            /* != 0: the four k4_lrelu_bwd launches run inside the epilogues of the launches in front of them (K4_EPI_LRELU_BWD, grad_x_lrelu):
                                       same values */
    int32_t g5_from_gx0_add;        /* != 0: g5 (a caller's [n_pix][nf] buffer) is WRITTEN here as 0.2 * gx0_add (= grad_out) */
    int32_t no_join;                /* ABI 13, != 0: k4_rdb_train_bwd does NOT make `stream` wait for side_stream before it returns -- the caller joins once,
                                       behind the last block (k4_main_wait_side), and keeps every buffer of the descriptor alive until then: with a join per
                                       block the chain waited ~40 us at every block for the weight gradient it had forked last */
    int32_t defer_side;             /* ABI 13, != 0 (with side_stream): the block's side-stream launches (zero-fill, five weight gradients, two SFT reductions) are issued at
                                       the END of the block behind ONE fork instead of one fork per launch -- an event record on `stream` in front of every dgrad launch cost
                                       ~7 us of the chain's time each.  Use with no_join (the weight gradients of a block then run beside the next block's chain). */
    void* aux_stream;               /* ABI 14, != NULL (with defer_side, no_join, gc_acc): the chain runs only k4_sft_train_bwd_gx for the block's two SFT layers; the rest of
                                       their backward (k4_sft_train_bwd_rest: condition gradient into gc_acc, partial sums) and the two reductions are issued on aux_stream
                                       at the end of the block, behind the same event as the side stream's launches.  The caller joins aux_stream (k4_main_wait_side) before
                                       the first reader of gc_acc and before the optimizer; the descriptor's buffers stay alive until then. */
    float* g5_next;                 /* ABI 14 (aux_stream form): sft0's grad_x launch also writes 0.2 * gx0 here -- the g5 of the block that receives gx0 as its grad_out */
    const float* gx0_add2; float* gx0_sum2;   /* ABI 14 (aux_stream form): ... and gx0_sum2 = gx0 + gx0_add2 (the RRDB's input gradient: last block's gx0 + the skip connection's) */
    int32_t g5_given;               /* ABI 14, != 0 with g5_from_gx0_add: g5 already holds 0.2 * gx0_add (the producer of gx0_add wrote it: g5_next / grad_x_scaled) */
    int32_t aux_wgrad;              /* ABI 14 (aux_stream form), != 0: conv1's weight gradient is issued on aux_stream (behind the SFT layers' deferred launches) instead of
                                       side_stream: the two streams carry about the same time per block */
} k4_rdb_train;
int k4_rdb_train_fwd(const k4_rdb_train* p, void* stream);
int k4_rdb_train_bwd(const k4_rdb_train* p, void* stream);

/* ---- training-step streaming kernels --------------------------------------------------------
 * Replace the reference extension `adam_upd_cuda` (lib/cuda/adam_upd.cpp:10-67 -> adam_upd_kernel.cu:60-133) that
 * MaskedAdam.step calls (lib/masked_adam.py:39-71).  fp32, n contiguous elements, updated in place; `step` >= 1 is
 * the 1-based step count the bias correction uses (step_size = lr*sqrt(1-beta2^step)/(1-beta1^step), .cu:71).
 *   k4_adam_upd            : every element                                      (adam_upd_kernel.cu:8-23)
 *   k4_masked_adam_upd     : only elements with grad != 0 (moments untouched)   (adam_upd_kernel.cu:25-41)
 *   k4_adam_upd_with_perlr : every element, step scaled by perlr[i]             (adam_upd_kernel.cu:43-58)      */
int k4_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                float beta1, float beta2, float lr, float eps, void* stream);
int k4_masked_adam_upd(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t step,
                       float beta1, float beta2, float lr, float eps, void* stream);
int k4_adam_upd_with_perlr(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* perlr,
                           int64_t n, int32_t step, float beta1, float beta2, float lr, float eps, void* stream);

/* k4_masked_adam_upd of a grid [channels][X][Y][Z] whose gradient is still the channel-last scratch image of
 * k4_grid_sample_3d_backward_cl_scatter (`workspace`): the touched voxels' non-zero sums are the gradient's only non-zero elements, every one of
 * them gets the masked update (same arithmetic per element), the workspace is all zero again on return.  For iterations in which nothing else
 * contributes to the grid's gradient (run_sr.py:1005-1014 after tv_before: no total variation) this is MaskedAdam.step
 * (lib/masked_adam.py:58-71, skip_zero_grad) without the dense gradient tensor. */
/* ABI 14 -- the masked step of a multi-channel grid in two exact parts (Adam is elementwise): k4_grid_flag_corners marks the voxels a scatter of `xyz` will touch
 * (the lookup's forward knows them: flags [X*Y*Z] bytes, the caller's, all-zero before the iteration's first lookup); k4_masked_adam_upd_unflagged steps every
 * unflagged voxel of [C][nvox] tensors with gradient `grad` (the dense TV term written ahead: K4_ERR_UNSUPPORTED unless nvox % 4 == 0 and the tensors are 16-byte
 * aligned); k4_masked_adam_upd_sparse_cl_seeded steps the flagged ones after the backward pass with gradient seed + the scatter's sums (the sweep's operand order)
 * and leaves `workspace` and `flags` all-zero.  Same `step` for both.  max_workgroups > 0 caps the first part's launch (grid-stride): it is meant to run beside
 * latency-bound kernels of other streams, which a pass at the full HBM rate slows by more than it saves.  Reference: lib/masked_adam.py:39-71 after run_sr.py:1005-1011. */
int k4_grid_flag_corners(int32_t X, int32_t Y, int32_t Z, const float* xyz, const float* xyz_min, const float* xyz_max, int64_t n, uint8_t* flags, void* stream);
int k4_masked_adam_upd_unflagged(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t channels, int64_t n_vox, const uint8_t* flags,
                                 int32_t step, float beta1, float beta2, float lr, float eps, int32_t max_workgroups, void* stream);
int k4_masked_adam_upd_sparse_cl_seeded(float* param, float* exp_avg, float* exp_avg_sq, void* workspace, const float* seed, uint8_t* flags,
                                        int32_t channels, int32_t X, int32_t Y, int32_t Z, int32_t step, float beta1, float beta2, float lr, float eps, void* stream);
int k4_masked_adam_upd_sparse_cl(float* param, float* exp_avg, float* exp_avg_sq, void* workspace, int32_t channels, int32_t X, int32_t Y, int32_t Z,
                                 int32_t step, float beta1, float beta2, float lr, float eps, void* stream);

/* The same update for MANY tensors in as few launches as possible (the decoder's optimizer step, run_sr.py:665-667,1014: 458 parameter
 * tensors = 458 launches of k4_adam_upd otherwise).  `jobs` is a HOST array; hyper-parameters and the step count are shared (one
 * param group whose tensors have been stepped together); masked != 0 selects the masked_adam_upd arithmetic.  Per element identical to
 * k4_adam_upd / k4_masked_adam_upd. */
#define K4_ADAM_MULTI_MAX 64       /* tensors per launch */
typedef struct k4_adam_job { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t n; } k4_adam_job;
int k4_adam_upd_multi(const k4_adam_job* jobs, int32_t n_jobs, int32_t masked, int32_t step, float beta1, float beta2, float lr, float eps,
                      void* stream);

/* total_variation_cuda.total_variation_add_grad (lib/cuda/total_variation.cpp:16-20 ->
 * total_variation_kernel.cu:13-66), called by DenseGrid.total_variation_add_grad (lib/grid.py:137-140).
 * param/grad: [1, C, sz_i, sz_j, sz_k] contiguous fp32, n = C*sz_i*sz_j*sz_k; grad += sum over the 6 neighbours of
 * (w_axis/6)*clamp(param - param_nb, -1, 1) with wx on the k (fastest) axis, wy on j, wz on i -- the reference's
 * naming; dense_mode == 0 restricts the update to elements whose grad is non-zero; dense_mode == 2 (no reference counterpart) WRITES the dense term:
 * grad = term, grad not read -- the term computed before the backward pass into the buffer the lookups' backward then accumulates into. */
int k4_total_variation_add_grad(const float* param, float* grad, float wx, float wy, float wz, int64_t sz_i,
                                int64_t sz_j, int64_t sz_k, int64_t n, int32_t dense_mode, void* stream);

/* ---- marcher training step: colour MLP with its backward, distortion loss (SURVEY.md 8f rank 1 "MLP bwd", 3.4) --------------
 * k4_rgbnet_fwd / k4_rgbnet_bwd replace, in the training graph, the `rgbnet` nn.Sequential + torch.sigmoid that the reference
 * evaluates and differentiates with PyTorch autograd over library GEMMs (lib/dmpigo.py:112-120,375-379; lib/dvgo.py:116-124,
 * 407-412).  Weights are the nn.Linear tensors AS STORED (w1 [width][dim0], w2 [width][width], w3 [3][width], row-major; b*),
 * nothing is repacked.  width in {32, 64, 128}, n_hidden in {0, 1} (rgbnet_depth 2 | 3), 1 <= dim0 <= 64; other shapes
 * return K4_ERR_UNSUPPORTED (the host keeps them on the nn.Module).
 *   fwd: rgb[n][3] = sigmoid(W3 relu(W2 relu(W1 x + b1) + b2) + b3 (+ add[n][3]));  h1 / h2 [n][width] (post-ReLU, NULL = not
 *        saved) are what the backward needs.  `add` = k0_diffuse of rgbnet_direct=False (lib/dvgo.py:412), NULL otherwise.
 *   bwd: grad_x [n][dim0] (NULL = not wanted), grad_logit [n][3] (= gradient of `add`; NULL = not wanted), gw* / gb* are
 *        OVERWRITTEN with the sums over the n samples.  Exact fp32 FMA chains, no atomics: per-workgroup partial sums land in
 *        `workspace` (k4_rgbnet_bwd_workspace_bytes) and are added in workgroup order. */
int k4_rgbnet_fwd(const float* x, int64_t n_pts, int32_t dim0, int32_t width, int32_t n_hidden,
                  const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                  const float* add, float* h1, float* h2, float* rgb, void* stream);
int64_t k4_rgbnet_bwd_workspace_bytes(int64_t n_pts, int32_t dim0, int32_t width, int32_t n_hidden);      /* <0: unsupported shape */
int k4_rgbnet_bwd(const float* x, int64_t n_pts, int32_t dim0, int32_t width, int32_t n_hidden,
                  const float* w1, const float* w2, const float* w3, const float* h1, const float* h2,
                  const float* rgb, const float* grad_rgb, float* grad_x, float* grad_logit,
                  float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3,
                  float* workspace, int64_t workspace_bytes, void* stream);
/* The colour MLP's input of DirectMPIGO's training forward (lib/dmpigo.py:360-374) in one launch instead of 16 PyTorch ops:
 *   x[i] = [ vox_emb[i] (channels) | pe_spa (3) | sin (3 P) | cos (3 P) of pe_spa[d] * posfreq[f] | viewdirs[ray_id[i]] (3) | sin (3 V) | cos (3 V) of viewdirs[d] * viewfreq[f] ],
 *   pe_spa[j] = ((ray_pts[i][2-j] - xyz_min[2-j]) / (xyz_max[2-j] - xyz_min[2-j])) * 2 - 1 -- the reference's op sequence, one rounding per op.
 * dim0 = channels + 3 + 6 P + 3 + 6 V.  Only vox_emb carries a gradient (grad_x[:, :channels]). */
int k4_rgbnet_input_mpi(const float* vox_emb, int32_t channels, const float* ray_pts, const float* viewdirs, const int64_t* ray_id, int64_t n_pts,
                        const float* xyz_min, const float* xyz_max, const float* posfreq, int32_t n_posfreq, const float* viewfreq, int32_t n_viewfreq,
                        float* x, int32_t dim0, void* stream);
/* The elementwise loss terms of the joint training iteration (run_sr.py:877-995) in one launch each way (as tensor-library ops: ~20 launches of a few
 * microseconds forward and as many backward):
 *   terms[0] photo   = weight_main * mean |rgb_feature - target|                                              (run_sr.py:877-881)
 *   terms[1] l1      = mean |rgb_sr - rgb_hr|,  rgb_hr[c][p] = target_4x[p][c]                                 (:925)
 *   terms[2] psnr_sr = -10 log10 mean (clamp(rgb_sr, 0, 1) - rgb_hr)^2                                         (:930; a metric, no gradient)
 *   terms[3] entropy = -mean(p log p + (1 - p) log(1 - p)) * weight_entropy_last, p = clamp(alphainv_last, 1e-6, 1 - 1e-6)   (:962-964; alphainv_last NULL: off)
 *   terms[4] rgbper  = weight_rgbper * sum_m |raw_rgb[m] - target[ray_id[m]]|^2 weights[m] / n_rays          (:993-995, weights a constant; raw_rgb NULL: off)
 *   total[0] = photo + l1 (+ entropy) (+ rgbper); the distortion term (k4_distortion_loss) is added by the caller.
 * rgb_sr element (channel c, pixel p) lies at c * sr_cstride + p * sr_pstride (NCHW: n_hr, 1; the decoder's NHWC result: 1, 3); grad_rgb_sr has the same
 * layout.  `acc`: 8 doubles, ALL ZERO on entry and again on return (allocate and clear once).  k4_joint_losses_bwd: gradients of total[0] times the device
 * scalar grad_total w.r.t. rgb_feature, rgb_sr, alphainv_last, raw_rgb (a NULL output is skipped). */
typedef struct k4_joint_losses {
    const float* rgb_feature; const float* target; int64_t n_rays;                 /* [n_rays][3] each */
    const float* rgb_sr; const float* target_4x; int64_t n_hr;                     /* n_hr pixels; target_4x [n_hr][3] */
    int64_t sr_cstride, sr_pstride;
    const float* alphainv_last;                                                    /* [n_rays] or NULL */
    const float* raw_rgb; const float* weights; const int64_t* ray_id; int64_t n_pts;   /* [n_pts][3], [n_pts], [n_pts]; raw_rgb NULL: term off */
    float weight_main, weight_entropy_last, weight_rgbper;
} k4_joint_losses;
int k4_joint_losses_fwd(const k4_joint_losses* d, double* acc, float* terms, float* total, void* stream);
int k4_joint_losses_bwd(const k4_joint_losses* d, const float* grad_total, float* grad_rgb_feature, float* grad_rgb_sr, float* grad_alphainv_last,
                        float* grad_raw_rgb, void* stream);
/* Distortion loss of the joint training step: run_sr.py:976-988 calls `flatten_eff_distloss(w, s, 1/n_max, ray_id)` of the
 * third-party package torch_efficient_distloss (not vendored in the reference tree).  Its published form is evaluated per
 * ray over samples sorted by s (ray_id ascending, as the marcher emits them):
 *   ray_loss[r] = sum_i [ interval/3 * w_i^2 + 2 w_i (s_i P_i - Q_i) ],   P / Q = exclusive prefix sums of w / w*s along the ray
 *   grad_w[i]   = d(sum_r ray_loss[r]) / d w_i
 * The package's value is sum(ray_loss) / (ray_id.max() + 1): that division stays on the host. */
int k4_distortion_loss(const float* w, const float* s, const int64_t* ray_id, int64_t n_pts, int64_t n_rays, float interval,
                       float* ray_loss, float* grad_w, void* stream);

/* Touched voxels (any of the `channels` planes non-zero) of a grid gradient [channels][n_vox] as a compact index list in arbitrary
 * order: what the data-parallel exchange of the joint step sends instead of the dense tensor.  *counter (device) = the total number of
 * touched voxels, also when it exceeds `cap` (then only `cap` indices were written: retry with a larger list). */
int k4_touched_voxels(const float* grad, int32_t channels, int64_t n_vox, int32_t* idx_out, int64_t cap, int64_t* counter, void* stream);

/* SFTLayer of the VC-Decoder in the training graph (lib/sr_esrnet.py:112-123 under autograd, run_sr.py:869-1014):
 *   y = x * (scale + 1) + shift,   scale = W1s lrelu(W0s c + b0s) + b1s,   shift = W1h lrelu(W0h c + b0h) + b1h
 * x / y / grad_y / grad_x: [n_pix][stride] rows of `channels` (32 | 64) floats; cond / grad_cond: [n_pix][32]; weights are the nn.Conv2d
 * tensors as stored (w0* [32][32], w1* [channels][32], row-major).  Forward: one launch, nothing saved (the backward recomputes the
 * hidden activations).  Backward: grad_x, grad_cond and the eight weight / bias gradients (OVERWRITTEN) in two launches; exact fp32
 * FMA chains, no atomics: per-workgroup partial sums go to `workspace` (k4_sft_train_bwd_workspace_bytes) and are added in
 * workgroup order.  Replaces ~13 + ~30 launches of the convolution Functions + elementwise glue per layer. */
int k4_sft_train_fwd(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, int64_t n_pix, int32_t channels,
                     const float* w0s, const float* b0s, const float* w1s, const float* b1s,
                     const float* w0h, const float* b0h, const float* w1h, const float* b1h,
                     float slope, float* y, int32_t y_stride, void* stream);
/* ... followed by the RRDB's skip connection in the store: y = sft(x) * res_scale + res ([n_pix][res_stride] rows; NULL = plain), two roundings as the
 * reference's two ops (lib/sr_esrnet.py:181).  Its backward: k4_sft_train_bwd_ex with grad_y_scale = res_scale (the skip's own gradient is grad_y itself). */
int k4_sft_train_fwd_ex(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, int64_t n_pix, int32_t channels,
                        const float* w0s, const float* b0s, const float* w1s, const float* b1s,
                        const float* w0h, const float* b0h, const float* w1h, const float* b1h,
                        float slope, float* y, int32_t y_stride, const float* res, int32_t res_stride, float res_scale, void* stream);
int64_t k4_sft_train_bwd_workspace_bytes(int64_t n_pix, int32_t channels);       /* < 0: unsupported channel count */
int k4_sft_train_bwd(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                     int64_t n_pix, int32_t channels,
                     const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                     float slope, float* grad_x, float* grad_cond,
                     float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                     float* workspace, int64_t workspace_bytes, void* stream);
/* The same with the two sums the training graph puts right behind it folded into the stores: grad_x = the layer's gradient + grad_x_add
 * ([n_pix][gxa_stride] rows; NULL = none) and, with accumulate_grad_cond != 0, grad_cond += the layer's gradient (the caller zeroes it once per
 * backward pass: every SFT layer of the decoder reads the same condition map).  grad_x_lrelu != 0: x is the OUTPUT of a LeakyReLU(slope) and
 * grad_x is the gradient in front of it: grad_x *= (x > 0 ? 1 : slope) (k4_lrelu_bwd folded in; before grad_x_add).  grad_y_scale: grad_y is multiplied by it as it is
 * read (1 = as is). */
int k4_sft_train_bwd_ex(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                        int64_t n_pix, int32_t channels,
                        const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                        float slope, float* grad_x, float* grad_cond,
                        float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                        float* workspace, int64_t workspace_bytes,
                        const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale, void* stream);

/* ---- launch tapes: the decoder's training pass as ONE native call (csrc/k4_tape.hip; SURVEY.md 8f rank 3, run_sr.py:869-1014) ------------------
 * The reference's joint step is one autograd graph over cuDNN calls; here SFTNet's forward + backward on a 64x64 patch is ~450 launches of 4-30 us
 * whose issue cost paced the iteration.  Between k4_tape_begin and k4_tape_end every RECORDABLE entry point the calling thread invokes runs as usual
 * AND is appended to the tape with a copy of its arguments (host structs and job arrays included); k4_tape_replay issues the recorded calls again
 * from C++, in order.  A call whose `stream` argument was `main_stream` runs on the replaying stream, a call placed on another stream (side-stream
 * weight gradients) stays there.  Recordable: k4_conv2d_nhwc_bf16x6, k4_conv2d_wgrad_bf16x6, k4_conv2d_wgrad_dbias_bf16x6(_acc), k4_conv2d_bias_grad,
 * k4_zero_f32, k4_pack_conv_weight_bf16x6(_multi), k4_lrelu_bwd, k4_sft_train_fwd(_ex), k4_sft_train_bwd(_ex), k4_rdb_train_fwd / _bwd (one entry
 * each) and the five entry points below.  The caller keeps every buffer a tape names alive and in place; one recording per thread at a time
 * (k4_tape_begin returns NULL otherwise). */
typedef struct k4_tape k4_tape;
k4_tape* k4_tape_begin(void* main_stream);
int k4_tape_end(k4_tape* tape);
int64_t k4_tape_length(const k4_tape* tape);                   /* recorded calls; -1 for NULL */
int k4_tape_replay(const k4_tape* tape, void* stream);
void k4_tape_free(k4_tape* tape);
/* The elementwise steps of SFTNet's training graph between the fused blocks (PyTorch ops and the autograd engine's gradient sums upstream):
 *   k4_add_f32             : out[i] = a[i] + b[i]                        (16-byte aligned; out may alias a or b)
 *   k4_upsample2x_nhwc     : y[2Y+py][2X+px][c] = x[Y][X][c]             (F.interpolate(scale_factor=2, mode='nearest'), lib/sr_esrnet.py:461-463)
 *   k4_upsample2x_bwd_nhwc : grad_x[Y][X][c] = (gy[2Y][2X] + gy[2Y][2X+1]) + (gy[2Y+1][2X] + gy[2Y+1][2X+1])      (channels % 4 == 0)
 *   k4_side_wait_main      : everything queued on `stream` so far completes before what `side` gets next (fork); k4_main_wait_side: the join. */
int k4_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
int k4_upsample2x_nhwc(const float* x, int32_t H, int32_t W, int32_t channels, float* y, void* stream);
int k4_upsample2x_bwd_nhwc(const float* grad_y, int32_t H, int32_t W, int32_t channels, float* grad_x, void* stream);
/* A stream (hipStreamNonBlocking; the device's least priority with low_priority != 0) whose kernels demonstrably run BESIDE those of main_stream and of every
 * stream in `others`: the runtime multiplexes streams onto a few hardware queues, and a second stream that shares the main stream's queue serialises with it
 * (csrc/k4_train.hip).  Up to 12 candidates are probed (a 200 us spin kernel on the one stream, a time stamp on the candidate); the stream lives until the process
 * ends.  NULL: creation failed.  k4_streams_overlap: the probe alone (1 / 0, < 0 on error). */
void* k4_stream_create_overlapping(void* main_stream, void* const* others, int32_t n_others, int32_t low_priority);
int k4_streams_overlap(void* a, void* b);
int k4_side_wait_main(void* side, void* stream);
int k4_main_wait_side(void* side, void* stream);

/* The two halves of k4_sft_train_bwd_ex as calls of their own (a caller that batches its side-stream work issues several layers' reductions behind ONE fork):
 * k4_sft_train_bwd_main = grad_x / grad_cond + the per-workgroup partial sums in `workspace`; k4_sft_train_reduce = the partials of a k4_sft_train_bwd_main
 * call on (n_pix, channels) summed in workgroup order into the eight parameter gradients (overwritten), on `stream`. */
int k4_sft_train_bwd_main(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                          int64_t n_pix, int32_t channels,
                          const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                          float slope, float* grad_x, float* grad_cond, float* workspace, int64_t workspace_bytes,
                          const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale, void* stream);
int k4_sft_train_reduce(const float* workspace, int64_t n_pix, int32_t channels,
                        float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h, void* stream);
/* ABI 14 -- the layer's backward as TWO launches for a caller with a third stream (the chain of the decoder's backward pass reads only grad_x):
 *   k4_sft_train_bwd_gx   : grad_x = grad_y * grad_y_scale * (scale(cond) + 1) [LeakyReLU mask from x when grad_x_lrelu] [+ grad_x_add] -- k4_sft_train_bwd_main's
 *                           grad_x bit for bit; x is read only when grad_x_lrelu != 0 (NULL otherwise).  Optional by-products, [n_pix][channels] each, that the
 *                           chain's next launches would otherwise compute in launches of their own: grad_x_scaled = grad_x * scaled_by (k4_rdb_train.g5 of the next
 *                           dense block), sum2 = grad_x + add2 (both or neither) -- one rounding each
 *   k4_sft_train_bwd_rest : everything else of k4_sft_train_bwd_main (grad_cond written or, accumulate_grad_cond != 0, added to; the partial sums in `workspace`
 *                           for k4_sft_train_reduce), same values.  Ordered by the caller behind grad_y's producer and in front of grad_cond's first reader.
 * Reference: autograd through SFTLayer.forward, lib/sr_esrnet.py:112-123. */
int k4_sft_train_bwd_gx(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                        int64_t n_pix, int32_t channels, const float* w0s, const float* b0s, const float* w1s, const float* b1s,
                        float slope, float* grad_x, const float* grad_x_add, int32_t gxa_stride, int32_t grad_x_lrelu, float grad_y_scale,
                        float* grad_x_scaled, float scaled_by, const float* add2, float* sum2, void* stream);
int k4_sft_train_bwd_rest(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                          int64_t n_pix, int32_t channels,
                          const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                          float slope, float* grad_cond, float* workspace, int64_t workspace_bytes, int32_t accumulate_grad_cond, float grad_y_scale, void* stream);
/* k4_sft_train_bwd_ex whose reduction of the per-workgroup partial sums (the eight parameter gradients; nothing on the caller's chain reads them) is forked
 * to side_stream (NULL = `stream`: k4_sft_train_bwd_ex): the caller joins side_stream before the gradients are read and keeps `workspace` untouched until then. */
int k4_sft_train_bwd_side(const float* x, int32_t x_stride, const float* cond, int32_t cond_stride, const float* grad_y, int32_t gy_stride,
                          int64_t n_pix, int32_t channels,
                          const float* w0s, const float* b0s, const float* w1s, const float* b1s, const float* w0h, const float* b0h, const float* w1h,
                          float slope, float* grad_x, float* grad_cond,
                          float* gw0s, float* gb0s, float* gw1s, float* gb1s, float* gw0h, float* gb0h, float* gw1h, float* gb1h,
                          float* workspace, int64_t workspace_bytes,
                          const float* grad_x_add, int32_t gxa_stride, int32_t accumulate_grad_cond, int32_t grad_x_lrelu, float grad_y_scale,
                          void* side_stream, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* K4NERF_H */
