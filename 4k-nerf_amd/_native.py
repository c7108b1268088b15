"""ctypes binding of lib4k_hip.so (C ABI: include/k4nerf.h).

The library is the product: there is NO fallback.  ``lib()`` raises if the shared object is missing
(run ``python -c "import __graft_entry__ as g; g.build()"``) and every wrapper raises on a non-zero
return code, on CPU tensors and on non-contiguous / wrong-dtype tensors -- the same guards the
reference's C++ entry points apply with TORCH_CHECK (lib/cuda/render_utils.cpp:46-48).
"""
import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('K4_LIB') or os.path.join(_PKG, 'lib4k_hip.so')      # K4_LIB: a variant build (A/B experiments, tools/)
K4_ABI_VERSION = 14
# True: the data-path collectives (tile all-gather, gradient exchange) are issued even on a process group of ONE rank -- the RCCL smoke test
# on a single GPU (tests/test_rccl_gpu.py: communicator + the production collective calls on device buffers); never set in production
FORCE_COLLECTIVES = False
K4_ERR_UNSUPPORTED = 10002

K0_CHANNEL_MAJOR, K0_CHANNEL_LAST = 0, 1


class GridDesc(C.Structure):
    _fields_ = [('density', C.c_void_p), ('k0', C.c_void_p), ('act_shift', C.c_void_p), ('mask', C.c_void_p),
                ('dims', C.c_int32 * 3), ('k0_ch', C.c_int32), ('k0_cpad', C.c_int32), ('k0_layout', C.c_int32),
                ('act_depth', C.c_int32), ('mask_dims', C.c_int32 * 3),
                ('xyz_min', C.c_float * 3), ('xyz_max', C.c_float * 3),
                ('xyz2ijk_scale', C.c_float * 3), ('xyz2ijk_shift', C.c_float * 3), ('occ_summary', C.c_void_p), ('depth_split', C.c_int32)]


class MlpDesc(C.Structure):
    _fields_ = [('packed', C.c_void_p), ('dim0', C.c_int32), ('width', C.c_int32), ('n_hidden', C.c_int32),
                ('viewbase_pe', C.c_int32), ('spatial_pe', C.c_int32), ('k0_skip', C.c_int32), ('arith', C.c_int32)]


class ConvJob(C.Structure):          # k4_conv_job
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('res', C.c_void_p), ('mod_x', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32)]


class ConvSftJob(C.Structure):       # k4_conv_sft_job
    _fields_ = [('cond', C.c_void_p), ('y2', C.c_void_p)]


class JointLosses(C.Structure):      # k4_joint_losses
    _fields_ = [('rgb_feature', C.c_void_p), ('target', C.c_void_p), ('n_rays', C.c_int64),
                ('rgb_sr', C.c_void_p), ('target_4x', C.c_void_p), ('n_hr', C.c_int64), ('sr_cstride', C.c_int64), ('sr_pstride', C.c_int64),
                ('alphainv_last', C.c_void_p),
                ('raw_rgb', C.c_void_p), ('weights', C.c_void_p), ('ray_id', C.c_void_p), ('n_pts', C.c_int64),
                ('weight_main', C.c_float), ('weight_entropy_last', C.c_float), ('weight_rgbper', C.c_float)]


class RdbTrain(C.Structure):         # k4_rdb_train
    _fields_ = [('H', C.c_int32), ('W', C.c_int32), ('nf', C.c_int32), ('g', C.c_int32),
                ('t', C.c_void_p), ('c', C.c_void_p), ('buf', C.c_void_p), ('x4', C.c_void_p), ('out', C.c_void_p),
                ('w_fwd', C.c_void_p * 5), ('b_fwd', C.c_void_p * 5), ('sft0', C.c_void_p * 8), ('sft1', C.c_void_p * 8),
                ('w_bwd', C.c_void_p * 5), ('b_bwd', C.c_void_p * 5), ('g5', C.c_void_p),
                ('G', C.c_void_p), ('gx4', C.c_void_p), ('gx0', C.c_void_p), ('gc0', C.c_void_p), ('gc1', C.c_void_p),
                ('dwdb', C.c_void_p * 5), ('gsft0', C.c_void_p * 8), ('gsft1', C.c_void_p * 8),
                ('ws0', C.c_void_p), ('ws0_bytes', C.c_int64), ('ws1', C.c_void_p), ('ws1_bytes', C.c_int64), ('side_stream', C.c_void_p),
                ('gc_acc', C.c_void_p), ('gx0_add', C.c_void_p), ('dwdb_span', C.c_void_p), ('dwdb_span_floats', C.c_int64),
                ('fused_lrelu', C.c_int32), ('g5_from_gx0_add', C.c_int32), ('no_join', C.c_int32), ('defer_side', C.c_int32),
                ('aux_stream', C.c_void_p), ('g5_next', C.c_void_p), ('gx0_add2', C.c_void_p), ('gx0_sum2', C.c_void_p),
                ('g5_given', C.c_int32), ('aux_wgrad', C.c_int32)]


class AdamJob(C.Structure):          # k4_adam_job
    _fields_ = [('param', C.c_void_p), ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p), ('n', C.c_int64)]


class PackJob(C.Structure):          # k4_pack_job
    _fields_ = [('w', C.c_void_p), ('bias', C.c_void_p), ('w_split', C.c_void_p), ('bias_out', C.c_void_p),
                ('cout', C.c_int32), ('cin', C.c_int32), ('ksize', C.c_int32), ('form', C.c_int32)]


class SftJob(C.Structure):           # k4_sft_job
    _fields_ = [('cond', C.c_void_p), ('x', C.c_void_p), ('y', C.c_void_p), ('res', C.c_void_p), ('n_pix', C.c_int64)]


K4_MAX_JOBS = 8
_P, _I32, _I64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_SIGS = {
    'k4_abi_version': [],
    'k4_march_mpi_fwd': [_P, _P, _P, _I64, _I32, C.POINTER(GridDesc), C.POINTER(MlpDesc), _I32, _F, _F, _F,
                         _P, _I64, _P, _P, _P, _P, _P],
    'k4_march_dvgo_fwd': [_P, _P, _P, _I64, _I32, C.POINTER(GridDesc), C.POINTER(MlpDesc), _F, _F, _F, _I32, _I32,
                          _F, _F, _F, _F, _P, _I64, _P, _P, _P, _P, _P],
    'k4_sample_ndc_pts_on_rays': [_P, _P, _P, _P, _I64, _I32, _P, _P, _P],
    'k4_infer_t_minmax': [_P, _P, _P, _P, _F, _F, _I64, _P, _P, _P],
    'k4_infer_n_samples': [_P, _P, _P, _F, _I64, _P, _P],
    'k4_infer_ray_start_dir': [_P, _P, _P, _I64, _P, _P, _P],
    'k4_sample_pts_on_rays_count': [_P, _P, _P, _P, _F, _F, _F, _I64, _P, _P, _P, _P],
    'k4_sample_pts_on_rays_fill': [_P, _P, _P, _P, _P, _P, _F, _I64, _I64, _P, _P, _P, _P, _P],
    'k4_maskcache_lookup': [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _P, _P],
    'k4_raw2alpha': [_P, _F, _F, _P, _I64, _P, _P, _P],
    'k4_raw2alpha_backward': [_P, _P, _F, _P, _I64, _P, _P],
    'k4_alpha2weight': [_P, _P, _I64, _I64, _P, _P, _P, _P, _P, _P],
    'k4_alpha2weight_backward': [_P, _P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P],
    'k4_grid_sample_3d': [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _P],
    'k4_segment_sum': [_P, _P, _I64, _I32, _I64, _P, _P],
    'k4_grid_sample_3d_backward': [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _P],
    'k4_touched_voxels': [_P, _I32, _I64, _P, _I64, _P, _P],
    'k4_grid_sample_3d_backward_cl': [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _P, _P],
    'k4_grid_sample_3d_backward_cl_scatter': [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _P],
    'k4_grid_sample_3d_backward_cl_sweep': [_I32, _I32, _I32, _I32, _P, _P, _P],
    'k4_segment_sum_backward': [_P, _P, _I64, _I32, _P, _P],
    'k4_get_rays_of_a_view': [_I32, _I32, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _P, _P, _P, _P],
    'k4_to8b': [_P, _I64, _P, _P],
    'k4_repack_k0': [_P, _I32, _I32, _I64, _P, _P],
    'k4_train_select_mpi': [_P, _P, _P, _P, _I64, _I32, _P, _P, _P, _I32, _I32, _I32, _P, _I32, _I32, _I32, _P, _I32, _F, _F, _P, _P, _P, _P, _P],
    'k4_train_compact': [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, _P],
    'k4_ndc_points_of': [_P, _P, _P, _P, _I64, _I32, _P, _P],
}
_lib = None


def lib():
    """Load lib4k_hip.so once; raise loudly when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the HIP extension is not built (python -c "import __graft_entry__ as g; '
                f'g.build()").  There is no CPU/PyTorch fallback for the 4K-NeRF hot path.')
        l = C.CDLL(LIB_PATH)
        missing = [name for name in list(_SIGS) + list(_EXTRA_SIGS) if not hasattr(l, name)]
        if missing:          # a stale library (built before an entry point was added) must not load silently
            raise RuntimeError(f'lib4k_hip.so lacks {missing}: rebuild (python -c "import __graft_entry__ as g; g.build()")')
        for name, args in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name, (args, res) in _EXTRA_SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        if l.k4_abi_version() != K4_ABI_VERSION:
            raise RuntimeError('lib4k_hip.so ABI version mismatch: rebuild')
        _lib = l
    return _lib


# entry points with a non-default return type / added after ABI v1 (all REQUIRED: lib() refuses a library lacking any)
_EXTRA_SIGS = {
    'k4_conv2d_nhwc': ([_P, _I32, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, C.c_uint32, _F,
                        _P, _I32, _F, _P, _I32, _P], C.c_int),
    'k4_conv_weight_floats': ([_I32, _I32, _I32], C.c_int64),
    'k4_sft_weight_floats': ([_I32], C.c_int64),
    'k4_sft_nhwc': ([_P, _I32, _P, _P, _I32, _P, _I32, _I32, _I64, _F, _P, _I32, _F, _P], C.c_int),
    'k4_adam_upd': ([_P, _P, _P, _P, _I64, _I32, _F, _F, _F, _F, _P], C.c_int),
    'k4_masked_adam_upd': ([_P, _P, _P, _P, _I64, _I32, _F, _F, _F, _F, _P], C.c_int),
    'k4_masked_adam_upd_sparse_cl': ([_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _F, _P], C.c_int),
    'k4_grid_flag_corners': ([_I32, _I32, _I32, _P, _P, _P, _I64, _P, _P], C.c_int),
    'k4_masked_adam_upd_unflagged': ([_P, _P, _P, _P, _I32, _I64, _P, _I32, _F, _F, _F, _F, _I32, _P], C.c_int),
    'k4_masked_adam_upd_sparse_cl_seeded': ([_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _F, _F, _F, _F, _P], C.c_int),
    'k4_adam_upd_with_perlr': ([_P, _P, _P, _P, _P, _I64, _I32, _F, _F, _F, _F, _P], C.c_int),
    'k4_total_variation_add_grad': ([_P, _P, _F, _F, _F, _I64, _I64, _I64, _I64, _I32, _P], C.c_int),
    'k4_conv_weight_bf16x6_bytes': ([_I32, _I32, _I32], C.c_int64),
    'k4_conv_weight_f16x3_bytes': ([_I32, _I32, _I32], C.c_int64),
    'k4_conv2d_nhwc_bf16x6': ([_P, _I32, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, C.c_uint32, _F,
                               _P, _I32, _F, _P, _I32, _P], C.c_int),
    'k4_march_workspace_bytes': ([_I64, _I32, _I32], C.c_int64),
    'k4_grid_sample_3d_backward_workspace_bytes': ([_I32, _I32, _I32, _I32], C.c_int64),
    'k4_mlp_packed_floats': ([_I32, _I32, _I32], C.c_int64),
    'k4_mlp_b2_layer1_terms': ([], C.c_int),
    'k4_mpi_depth_split_stats': ([C.POINTER(GridDesc), C.c_float, C.c_float, _P, _P], C.c_int),
    'k4_conv2d_nhwc_bf16x6_multi': ([C.POINTER(ConvJob), _I32, _I32, _I32, _P, _P, _I32, _I32, _I32, C.c_uint32, _F, _I32, _F, _I32, _P], C.c_int),
    'k4_sft_nhwc_multi': ([C.POINTER(SftJob), _I32, _I32, _P, _I32, _I32, _I32, _F, _I32, _F, _I32, _P], C.c_int),
    'k4_conv_weight_p16_bytes': ([_I32, _I32], C.c_int64),
    'k4_conv_weight_p16_up2x_bytes': ([_I32, _I32], C.c_int64),
    'k4_conv3x3_p16_multi': ([C.POINTER(ConvJob), _I32, _I32, _I32, _P, _P, _I32, _I32, C.c_uint32, _F, _I32, _F, _F, _P, _P], C.c_int),
    'k4_conv_sft_epilogue_bytes': ([_I32], C.c_int64),
    'k4_conv3x3_p16_sft_multi': ([C.POINTER(ConvJob), C.POINTER(ConvSftJob), _I32, _I32, _I32, _P, _P, _I32, _I32, C.c_uint32, _F, _I32, _F,
                                 _I32, _F, _P, _F, _I32, _F, _P, _P], C.c_int),
    'k4_sft_nhwc_p16_multi': ([C.POINTER(SftJob), _I32, _I32, _P, _I32, _I32, _I32, _F, _F, _P, _P], C.c_int),
    'k4_absmax_slice': ([_P, _I64, _I32, _I32, _P, _P], C.c_int),
    'k4_conv2d_wgrad_bf16x6': ([_P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P], C.c_int),
    'k4_conv2d_wgrad_dbias_bf16x6': ([_P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P], C.c_int),
    'k4_conv2d_wgrad_dbias_bf16x6_acc': ([_P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _P], C.c_int),
    'k4_zero_f32': ([_P, _I64, _P], C.c_int),
    'k4_pack_conv_weight_bf16x6_multi': ([C.POINTER(PackJob), _I32, _P], C.c_int),
    'k4_rdb_train_fwd': ([C.POINTER(RdbTrain), _P], C.c_int),
    'k4_rdb_train_bwd': ([C.POINTER(RdbTrain), _P], C.c_int),
    'k4_lrelu_bwd': ([_P, _I32, _P, _I32, _I64, _I32, _F, _P, _I32, _P], C.c_int),
    'k4_conv2d_bias_grad': ([_P, _I32, _I32, _I64, _P, _P], C.c_int),
    'k4_adam_upd_multi': ([C.POINTER(AdamJob), _I32, _I32, _I32, _F, _F, _F, _F, _P], C.c_int),
    'k4_pack_conv_weight_bf16x6': ([_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P], C.c_int),
    'k4_resample_trilinear': ([_P, _I32, _I32, _I32, _I32, _P, _I32, _I32, _I32, _P], C.c_int),
    'k4_alpha_maxpool3_gt': ([_P, _I32, _I32, _I32, _F, _P, _P], C.c_int),
    'k4_occupancy_summary_bytes': ([_I32, _I32, _I32], C.c_int64),
    'k4_build_occupancy_summary': ([_P, _I32, _I32, _I32, _P, _P], C.c_int),
    'k4_live_mask_workspace_bytes': ([_I32, _I32, _I32], C.c_int64),
    'k4_build_live_mask': ([C.POINTER(GridDesc), _F, _F, _F, _P, _P, _P], C.c_int),
    'k4_rgbnet_fwd': ([_P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    'k4_rgbnet_bwd_workspace_bytes': ([_I64, _I32, _I32, _I32], C.c_int64),
    'k4_rgbnet_bwd': ([_P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'k4_sft_train_fwd': ([_P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I32, _P], C.c_int),
    'k4_sft_train_fwd_ex': ([_P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I32, _P, _I32, _F, _P], C.c_int),
    'k4_sft_train_bwd_workspace_bytes': ([_I64, _I32], C.c_int64),
    'k4_sft_train_bwd': ([_P, _I32, _P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P,
                          _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P], C.c_int),
    'k4_sft_train_bwd_ex': ([_P, _I32, _P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P,
                             _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _F, _P], C.c_int),
    'k4_sft_train_bwd_side': ([_P, _I32, _P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P,
                               _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _F, _P, _P], C.c_int),
    'k4_sft_train_bwd_main': ([_P, _I32, _P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _F, _P], C.c_int),
    'k4_sft_train_reduce': ([_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    'k4_sft_train_bwd_gx': ([_P, _I32, _P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _F, _P, _P, _I32, _I32, _F, _P, _F, _P, _P, _P], C.c_int),
    'k4_sft_train_bwd_rest': ([_P, _I32, _P, _I32, _P, _I32, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I64, _I32, _F, _P], C.c_int),
    'k4_rgbnet_input_mpi': ([_P, _I32, _P, _P, _P, _I64, _P, _P, _P, _I32, _P, _I32, _P, _I32, _P], C.c_int),
    'k4_joint_losses_fwd': ([_P, _P, _P, _P, _P], C.c_int),
    'k4_joint_losses_bwd': ([_P, _P, _P, _P, _P, _P, _P], C.c_int),
    'k4_distortion_loss': ([_P, _P, _P, _I64, _I64, _F, _P, _P, _P], C.c_int),
    'k4_nhwc_window_to_planar': ([_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _P], C.c_int),
    'k4_tape_begin': ([_P], C.c_void_p),
    'k4_tape_end': ([_P], C.c_int),
    'k4_tape_length': ([_P], C.c_int64),
    'k4_tape_replay': ([_P, _P], C.c_int),
    'k4_tape_free': ([_P], None),
    'k4_add_f32': ([_P, _P, _P, _I64, _P], C.c_int),
    'k4_upsample2x_nhwc': ([_P, _I32, _I32, _I32, _P, _P], C.c_int),
    'k4_upsample2x_bwd_nhwc': ([_P, _I32, _I32, _I32, _P, _P], C.c_int),
    'k4_stream_create_overlapping': ([_P, C.POINTER(C.c_void_p), _I32, _I32], C.c_void_p),
    'k4_streams_overlap': ([_P, _P], C.c_int),
    'k4_side_wait_main': ([_P, _P], C.c_int),
    'k4_main_wait_side': ([_P, _P], C.c_int),
}


class K4Error(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        if rc == 10001:
            msg = 'K4_ERR_BAD_ARG'
        elif rc == K4_ERR_UNSUPPORTED:
            msg = 'K4_ERR_UNSUPPORTED'
        else:
            msg = f'hipError {rc}'
        raise K4Error(f'{what} failed: {msg}')


def ptr(t):
    """Device pointer of a CUDA(HIP), contiguous tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise K4Error('tensor must be on the GPU (no CPU path exists for this op)')
    if not t.is_contiguous():
        raise K4Error('tensor must be contiguous')
    return C.c_void_p(t.data_ptr())


def f32(t):
    if t.dtype != torch.float32:
        raise K4Error(f'expected float32, got {t.dtype}')
    return ptr(t)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """hipStream_t of torch's current stream on the current device.  The raw accessors cost ~0.3 us; `torch.cuda.current_stream()`
    builds a Stream object through several Python layers (~9 us -- 8 ms of a 48 ms training iteration with ~900 launches)."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def vec3(t):
    v = [float(x) for x in t.detach().cpu().reshape(-1).tolist()]
    assert len(v) == 3
    return (C.c_float * 3)(*v)


_OVERLAP_STREAMS = {}           # (device index, main stream handle, group) -> {tag: torch.cuda.ExternalStream}


def overlapping_stream(device, tag, low_priority=False, group='training step', beside_main=True):
    """The side stream `tag` of the CURRENT stream on `device`: a stream verified to run beside the current stream and beside the streams the other tags of
    the same group already hold (include/k4nerf.h k4_stream_create_overlapping: HIP streams share a few hardware queues, and a "second stream" that
    lands on the main stream's queue serialises with it -- the joint training iteration took 9 ms or 21 ms depending on how many streams the process had
    created before).  One stream per (device, main stream, group, tag) for the life of the process.  beside_main=False: verified against the group's
    other streams only (a pool of worker streams that run while the main stream idles: four hardware queues cannot hold four workers AND the main stream)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(idx).cuda_stream
    mine = _OVERLAP_STREAMS.setdefault((idx, main, group), {})
    st = mine.get(tag)
    if st is None:
        others = [s.cuda_stream for s in mine.values()]
        first = main
        if not beside_main:                              # probe against the first worker instead of the main stream (none yet: any stream will do)
            first, others = (others[0], others[1:]) if others else (None, [])
        if first is None or torch.cuda.is_current_stream_capturing():
            # (no probe inside a hipGraph capture -- it allocates and synchronises: a plain pool stream, kept for this capture stream only)
            st = mine[tag] = torch.cuda.Stream(device=idx)
            return st
        arr = (C.c_void_p * max(1, len(others)))(*others)
        with torch.cuda.device(idx):
            raw = lib().k4_stream_create_overlapping(C.c_void_p(first), arr, len(others), int(bool(low_priority)))
        if not raw:
            raise K4Error('k4_stream_create_overlapping failed')
        st = mine[tag] = torch.cuda.ExternalStream(raw, device=idx)
    return st
