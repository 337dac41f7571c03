"""ctypes binding of libimvoxel_hip.so (the C-ABI declared in include/imvoxel.h).

The product path fails loudly when the library is missing or a call fails: there is no CPU
fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IVX_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libimvoxel_hip.so')      # (IVX_LIB_PATH: A/B builds of the same ABI, tools/)
_lib = None


class IvxError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('B', 'D', 'H', 'W', 'Cin', 'Cout', 'KD', 'KH', 'KW', 'sd', 'sh', 'sw', 'pd', 'ph', 'pw',
                 'relu', 'res_mode', 'res_h', 'res_w', 'wgt_layout', 'out_mode', 'res_after_act')] + [('post_scale', C.c_float),
                                                                  ('in_dtype', C.c_int32), ('out_dtype', C.c_int32), ('res_scale', C.c_float),
                                                                  ('wino_operands', C.c_int32)]


class AnchorHeadDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ('B', 'H', 'W', 'CH', 'num_anchors', 'num_classes', 'cls_off', 'reg_off', 'dir_off',
                 'nms_pre', 'max_num', 'use_rotate_nms', 'hw_transposed')] + \
               [(n, C.c_float) for n in ('score_thr', 'nms_thr', 'dir_offset', 'dir_limit_offset')]


class ModelCfg(C.Structure):
    """ivx_model_cfg (include/imvoxel.h)."""
    _fields_ = [('neck_type', C.c_int32), ('with_trunk', C.c_int32), ('fpn_channels', C.c_int32), ('neck_out_channels', C.c_int32),
                ('n_voxels', C.c_int32 * 3), ('voxel_size', C.c_float * 3), ('num_classes', C.c_int32), ('n_sizes', C.c_int32),
                ('n_rotations', C.c_int32), ('anchor_range', C.c_float * 6), ('anchor_sizes', C.c_float * 12),
                ('anchor_rotations', C.c_float * 4), ('nms_pre', C.c_int32), ('max_num', C.c_int32), ('use_rotate_nms', C.c_int32),
                ('score_thr', C.c_float), ('nms_thr', C.c_float), ('dir_offset', C.c_float), ('dir_limit_offset', C.c_float),
                ('winograd', C.c_int32), ('winograd_tile', C.c_int32),
                ('fast_n_blocks', C.c_int32 * 3), ('unet_channels', C.c_int32 * 4), ('unet_down_layers', C.c_int32 * 4),
                ('unet_up_layers', C.c_int32 * 3),
                ('head_type', C.c_int32), ('head_classes', C.c_int32), ('head_nms_pre', C.c_int32), ('head_use_rotate_nms', C.c_int32),
                ('head_score_thr', C.c_float), ('head_nms_thr', C.c_float), ('dcn_stages', C.c_int32 * 4), ('layout_head', C.c_int32),
                ('layout_linear_size', C.c_int32), ('wino_operands', C.c_int32), ('trunk_operands', C.c_int32), ('storage', C.c_int32)]


class PairIO(C.Structure):
    """ivx_pair_io: device-side scales / amax slots of a convolution on fp16-pair activations (ivx_conv_fwd_pio)."""
    _fields_ = [('in_scale', C.c_void_p), ('res_scale', C.c_void_p), ('res_dtype', C.c_int32), ('out_scale', C.c_void_p), ('amax_in', C.c_void_p),
                ('amax_res', C.c_void_p), ('amax_out', C.c_void_p), ('wbound', C.c_float), ('sbound', C.c_float)]


class BottleneckDesc(C.Structure):
    """ivx_bottleneck_desc."""
    _fields_ = [(n, C.c_int32) for n in ('B', 'H', 'W', 'P')]


class BottleneckIO(C.Structure):
    """ivx_bottleneck_io: scalar blocks of the fused bottleneck's input / output and the bound terms of its three layers."""
    _fields_ = [('in_scale', C.c_void_p), ('amax_in', C.c_void_p), ('out_scale', C.c_void_p), ('amax_out', C.c_void_p), ('wbound', C.c_float * 3),
                ('sbound', C.c_float * 3)]


class SampleMeta(C.Structure):
    """ivx_sample_meta."""
    _fields_ = [('intrinsic', C.c_float * 16), ('extrinsics', C.c_void_p), ('origin', C.c_float * 3), ('img_h', C.c_int32), ('img_w', C.c_int32),
                ('ori_h', C.c_int32)]


class IndoorTailDesc(C.Structure):
    """ivx_indoor_tail_desc."""
    _fields_ = [('B', C.c_int32), ('n_levels', C.c_int32), ('k', C.c_int32 * 4), ('n_classes', C.c_int32), ('n_reg', C.c_int32),
                ('use_rotate_nms', C.c_int32), ('max_num', C.c_int32), ('score_thr', C.c_float), ('nms_thr', C.c_float)]


class TraceRec(C.Structure):
    """ivx_trace_rec."""
    _fields_ = [('step', C.c_int32), ('stage', C.c_int32), ('is3d', C.c_int32), ('ms', C.c_float), ('start_ms', C.c_float), ('flops', C.c_double),
                ('bytes', C.c_double), ('name', C.c_char * 48)]


EXPORTS = ['ivx_bottleneck_proj_supported', 'ivx_bottleneck_proj_pack', 'ivx_bottleneck_proj_fwd_pio', 'ivx_anchor_head_decode', 'ivx_fcos3d_head_decode', 'ivx_nms_rotated_bev', 'ivx_nms_aligned3d', 'ivx_bottleneck_supported', 'ivx_bottleneck_fwd_pio', 'ivx_amax_f32', 'ivx_stem_pool_filter_bytes', 'ivx_stem_pool_pack_filters', 'ivx_stem_pool_out_dims', 'ivx_stem_pool_fwd_pair', 'ivx_conv_winograd_set_variant', 'ivx_ubench_mfma', 'ivx_ubench_copy', 'ivx_version', 'ivx_last_error', 'ivx_conv_out_dims', 'ivx_conv_fwd', 'ivx_conv_fwd_naive', 'ivx_conv_set_tile_override', 'ivx_conv_set_epilogue_mode', 'ivx_conv_set_plan_mode', 'ivx_topk_set_mode', 'ivx_conv_workspace_bytes', 'ivx_conv_fwd_ws', 'ivx_conv_set_halo_mode', 'ivx_bf16_pair_split', 'ivx_f16_pair_split', 'ivx_conv_pair_supported', 'ivx_conv_pio_workspace_bytes', 'ivx_conv_fwd_pio', 'ivx_conv_fwd_pio_naive', 'ivx_pair_pack_filters', 'ivx_nchw_to_nhwc_amax', 'ivx_maxpool2d_fwd_pair', 'ivx_f16_pair_merge', 'ivx_model_calibrate_fp8', 'ivx_model_calibrate_fp8_ex', 'ivx_amax_bf16', 'ivx_conv_winograd_output_blocks', 'ivx_conv_winograd_issued_fraction', 'ivx_conv_winograd_output_amax', 'ivx_conv_winograd_input_amax', 'ivx_conv_winograd_fused_supported', 'ivx_conv_winograd_fused_blocks', 'ivx_conv_winograd_gemm_output_amax',
           'ivx_conv_winograd_supported', 'ivx_conv_winograd_weight_elems', 'ivx_conv_winograd_weights', 'ivx_conv_winograd_workspace_bytes',
           'ivx_conv_winograd_input', 'ivx_conv_winograd_gemm', 'ivx_conv_winograd_output', 'ivx_conv_winograd_fwd',
           'ivx_maxpool2d_fwd', 'ivx_maxpool2d_fwd_bf16', 'ivx_maxpool2d_fwd_fp8', 'ivx_global_avgpool_fwd', 'ivx_upsample_trilinear2x_fwd', 'ivx_dcn_im2col_fwd', 'ivx_dcn_im2col_fwd_pair', 'ivx_nchw_to_nhwc', 'ivx_image_s2d_bf16', 'ivx_nhwc_to_nchw', 'ivx_backproject_mean_fwd', 'ivx_backproject_mean_fwd_amax', 'ivx_backproject_amax_blocks', 'ivx_backproject_mean_fwd_bf16', 'ivx_upsample_trilinear2x_fwd_bf16', 'ivx_backproject_sum_fwd', 'ivx_volume_normalize_fwd',
           'ivx_anchor_head_workspace_bytes', 'ivx_anchor_head_get_bboxes', 'ivx_fcos_head_workspace_bytes',
           'ivx_fcos_head_level_candidates', 'ivx_nms_workspace_bytes',
           'ivx_nms_bev', 'ivx_boxes_overlap_bev', 'ivx_aligned_3d_nms', 'ivx_aligned_3d_nms_workspace_bytes', 'ivx_aligned_3d_nms_ws', 'ivx_multiclass_nms_workspace_bytes', 'ivx_multiclass_nms_bev',
           'ivx_create', 'ivx_destroy', 'ivx_weights_load', 'ivx_weights_finalize', 'ivx_model_workspace_bytes', 'ivx_model_forward',
           'ivx_backbone_fpn_workspace_bytes', 'ivx_backbone_fpn_fwd', 'ivx_neck3d_workspace_bytes', 'ivx_neck3d_out_dims',
           'ivx_neck3d_kitti_fwd', 'ivx_neck3d_nuscenes_fwd', 'ivx_neck3d_levels', 'ivx_neck3d_fast_fwd', 'ivx_neck3d_unet_fwd',
           'ivx_model_forward_levels', 'ivx_model_anchors', 'ivx_compute_projection', 'ivx_voxel_new_origin', 'ivx_fold_batchnorm',
           'ivx_model_trace', 'ivx_model_trace_count', 'ivx_model_trace_read',
           'ivx_model_detect_workspace_bytes', 'ivx_model_max_detections', 'ivx_model_detect', 'ivx_layout_head_decode', 'ivx_layout_extrinsics',
           'ivx_indoor_tail_workspace_bytes', 'ivx_indoor_tail_get_bboxes',
           'ivx_kitti_image_box_overlap', 'ivx_kitti_compute_statistics', 'ivx_kitti_collect_scores', 'ivx_kitti_fused_statistics']


def lib():
    """Load (once) and return the library.  Raises IvxError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IvxError(f'{LIB_PATH} is missing: build it with `python -m imvoxelnet_amd._build` '
                       '(or __graft_entry__.build()).  There is no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.ivx_version.restype = C.c_int
    L.ivx_last_error.restype = C.c_char_p
    L.ivx_conv_out_dims.argtypes = [C.POINTER(ConvDesc), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    for name in ('ivx_conv_fwd', 'ivx_conv_fwd_naive'):
        getattr(L, name).argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]
    L.ivx_conv_set_tile_override.argtypes = [C.c_int]
    L.ivx_conv_set_halo_mode.argtypes = [C.c_int]
    L.ivx_conv_set_epilogue_mode.argtypes = [C.c_int]
    L.ivx_conv_set_plan_mode.argtypes = [C.c_int]
    L.ivx_conv_winograd_output_blocks.argtypes = [C.POINTER(ConvDesc), i32]
    L.ivx_conv_winograd_output_blocks.restype = i32
    L.ivx_conv_winograd_output_amax.argtypes = [C.POINTER(ConvDesc), i32, vp, vp, vp, vp, vp, i64, vp, vp]
    L.ivx_conv_winograd_input_amax.argtypes = [C.POINTER(ConvDesc), i32, vp, vp, i64, vp, i32, vp]
    L.ivx_conv_winograd_fused_supported.argtypes = [C.POINTER(ConvDesc), i32]
    L.ivx_conv_winograd_fused_blocks.argtypes = [C.POINTER(ConvDesc), i32]
    L.ivx_conv_winograd_fused_blocks.restype = i32
    L.ivx_conv_winograd_gemm_output_amax.argtypes = [C.POINTER(ConvDesc), i32, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.ivx_bf16_pair_split.argtypes = [vp, i64, vp, vp]
    L.ivx_f16_pair_split.argtypes = [vp, i64, f32, vp, vp]
    L.ivx_conv_pair_supported.argtypes = [C.POINTER(ConvDesc)]
    L.ivx_conv_pio_workspace_bytes.argtypes = [C.POINTER(ConvDesc), C.POINTER(PairIO)]
    L.ivx_conv_pio_workspace_bytes.restype = i64
    L.ivx_amax_f32.argtypes = [vp, i64, vp, vp]
    L.ivx_stem_pool_filter_bytes.restype = i64
    L.ivx_stem_pool_filter_bytes.argtypes = []
    L.ivx_stem_pool_pack_filters.argtypes = [vp, vp, vp, vp]
    L.ivx_stem_pool_out_dims.argtypes = [i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.ivx_stem_pool_fwd_pair.argtypes = [vp, i32, i32, i32, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp]
    L.ivx_bottleneck_supported.argtypes = [C.POINTER(BottleneckDesc)]
    L.ivx_bottleneck_fwd_pio.argtypes = [C.POINTER(BottleneckDesc), C.POINTER(BottleneckIO)] + [vp] * 12
    L.ivx_bottleneck_proj_supported.argtypes = [C.POINTER(BottleneckDesc), i32]
    L.ivx_bottleneck_proj_pack.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
    L.ivx_bottleneck_proj_fwd_pio.argtypes = [C.POINTER(BottleneckDesc), i32, C.POINTER(BottleneckIO), f32] + [vp] * 12
    L.ivx_conv_fwd_pio.argtypes = [C.POINTER(ConvDesc), C.POINTER(PairIO), vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.ivx_conv_fwd_pio_naive.argtypes = [C.POINTER(ConvDesc), C.POINTER(PairIO), vp, vp, vp, vp, vp, vp, vp]
    L.ivx_pair_pack_filters.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, C.POINTER(f32), C.POINTER(f32)]
    L.ivx_nchw_to_nhwc_amax.argtypes = [vp, i32, i32, i64, i32, vp, vp, vp]
    L.ivx_maxpool2d_fwd_pair.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp]
    L.ivx_f16_pair_merge.argtypes = [vp, i64, vp, vp, vp]
    L.ivx_conv_workspace_bytes.argtypes = [C.POINTER(ConvDesc)]
    L.ivx_conv_workspace_bytes.restype = i64
    L.ivx_conv_fwd_ws.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, i64, vp]
    D = C.POINTER(ConvDesc)
    L.ivx_conv_winograd_supported.argtypes = [D, i32]
    L.ivx_conv_winograd_issued_fraction.argtypes = [D]
    L.ivx_conv_winograd_issued_fraction.restype = C.c_float
    L.ivx_conv_winograd_weight_elems.argtypes = [D, i32]
    L.ivx_conv_winograd_weight_elems.restype = i64
    L.ivx_conv_winograd_weights.argtypes = [D, i32, vp, vp, vp]
    L.ivx_conv_winograd_workspace_bytes.argtypes = [D, i32]
    L.ivx_conv_winograd_workspace_bytes.restype = i64
    L.ivx_conv_winograd_input.argtypes = [D, i32, vp, vp, i64, vp]
    L.ivx_conv_winograd_gemm.argtypes = [D, i32, vp, vp, i64, vp]
    L.ivx_conv_winograd_output.argtypes = [D, i32, vp, vp, vp, vp, vp, i64, vp]
    L.ivx_conv_winograd_fwd.argtypes = [D, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.ivx_maxpool2d_fwd.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.ivx_global_avgpool_fwd.argtypes = [vp, i32, i64, i32, vp, vp]
    L.ivx_upsample_trilinear2x_fwd.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    L.ivx_dcn_im2col_fwd.argtypes = [vp, vp] + [i32] * 10 + [vp, vp]
    L.ivx_dcn_im2col_fwd_pair.argtypes = [vp, vp, vp] + [i32] * 10 + [vp, vp, vp, vp]
    L.ivx_nchw_to_nhwc.argtypes = [vp, i32, i32, i64, i32, vp, vp]
    L.ivx_nhwc_to_nchw.argtypes = [vp, i32, i64, i32, vp, vp]
    L.ivx_backproject_mean_fwd.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, C.POINTER(f32), i32, i32, i32,
                                           vp, vp, vp]
    L.ivx_backproject_sum_fwd.argtypes = L.ivx_backproject_mean_fwd.argtypes
    L.ivx_backproject_mean_fwd_bf16.argtypes = L.ivx_backproject_mean_fwd.argtypes
    L.ivx_upsample_trilinear2x_fwd_bf16.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    L.ivx_volume_normalize_fwd.argtypes = [vp, vp, i64, i32, vp, vp]
    L.ivx_anchor_head_workspace_bytes.argtypes = [C.POINTER(AnchorHeadDesc)]
    L.ivx_anchor_head_workspace_bytes.restype = i64
    L.ivx_anchor_head_get_bboxes.argtypes = [C.POINTER(AnchorHeadDesc), vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ivx_fcos_head_workspace_bytes.argtypes = [i32, i32, i32]
    L.ivx_fcos_head_workspace_bytes.restype = i64
    L.ivx_fcos_head_level_candidates.argtypes = [vp, vp, vp, vp, f32] + [i32] * 12 + [vp, i64, vp, vp, vp, vp]
    L.ivx_nms_workspace_bytes.argtypes = [i32]
    L.ivx_nms_workspace_bytes.restype = i64
    L.ivx_nms_bev.argtypes = [vp, i32, f32, i32, vp, i64, vp, vp, vp]
    L.ivx_boxes_overlap_bev.argtypes = [vp, i32, vp, i32, i32, vp, vp]
    L.ivx_aligned_3d_nms.argtypes = [vp, vp, vp, i32, f32, vp, vp, vp]
    L.ivx_aligned_3d_nms_workspace_bytes.argtypes = [i32]
    L.ivx_aligned_3d_nms_workspace_bytes.restype = i64
    L.ivx_aligned_3d_nms_ws.argtypes = [vp, vp, vp, i32, f32, vp, i64, vp, vp, vp]
    L.ivx_multiclass_nms_workspace_bytes.argtypes = [i32, i32]
    L.ivx_multiclass_nms_workspace_bytes.restype = i64
    L.ivx_multiclass_nms_bev.argtypes = [vp, vp, i32, i32, i32, f32, f32, i32, i32, vp, i64, vp, vp, vp, vp]
    L.ivx_create.argtypes = [C.POINTER(ModelCfg), C.POINTER(vp)]
    L.ivx_destroy.argtypes = [vp]
    L.ivx_weights_load.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32]
    L.ivx_weights_finalize.argtypes = [vp, vp]
    L.ivx_model_workspace_bytes.argtypes = [vp, i32, i32, i32, i32]
    L.ivx_model_workspace_bytes.restype = i64
    L.ivx_model_forward.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp]
    L.ivx_backbone_fpn_workspace_bytes.argtypes = [vp, i32, i32, i32]
    L.ivx_backbone_fpn_workspace_bytes.restype = i64
    L.ivx_backbone_fpn_fwd.argtypes = [vp, vp, i32, i32, i32, vp, vp, i64, vp]
    L.ivx_neck3d_workspace_bytes.argtypes = [vp, i32]
    L.ivx_neck3d_workspace_bytes.restype = i64
    L.ivx_neck3d_out_dims.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ivx_neck3d_kitti_fwd.argtypes = [vp, vp, i32, vp, vp, i64, vp]
    L.ivx_neck3d_nuscenes_fwd.argtypes = [vp, vp, i32, vp, vp, i64, vp]
    L.ivx_topk_set_mode.argtypes = [i32]
    L.ivx_image_s2d_bf16.argtypes = [vp, i32, i32, i32, vp, vp]
    L.ivx_neck3d_levels.argtypes = [vp, i32, vp]
    L.ivx_neck3d_fast_fwd.argtypes = [vp, vp, i32, vp, vp, i64, vp]
    L.ivx_neck3d_unet_fwd.argtypes = [vp, vp, i32, vp, vp, i64, vp]
    L.ivx_model_forward_levels.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp]
    L.ivx_model_anchors.argtypes = [vp, i32, i32, vp, i64]
    L.ivx_compute_projection.argtypes = [vp, vp, i32, C.c_double, vp]
    L.ivx_voxel_new_origin.argtypes = [vp, vp, vp, vp]
    L.ivx_fold_batchnorm.argtypes = [vp, vp, vp, vp, vp, f32, i32, vp, vp]
    L.ivx_model_calibrate_fp8.argtypes = [vp, vp, i32, i32, i32, f32, vp, i64, vp]
    L.ivx_model_calibrate_fp8_ex.argtypes = [vp, vp, i32, i32, i32, f32, i32, i32, vp, i64, vp]
    L.ivx_amax_bf16.argtypes = [vp, i64, vp, vp]
    L.ivx_model_trace.argtypes = [vp, i32]
    L.ivx_model_detect_workspace_bytes.argtypes = [vp, i32, i32, i32, i32]
    L.ivx_model_detect_workspace_bytes.restype = i64
    L.ivx_model_max_detections.argtypes = [vp, i32, i32, i32, i32]
    L.ivx_model_max_detections.restype = i32
    L.ivx_model_detect.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ivx_layout_head_decode.argtypes = [vp, vp, vp, vp]
    L.ivx_layout_extrinsics.argtypes = [vp, vp]
    L.ivx_indoor_tail_workspace_bytes.argtypes = [vp]
    L.ivx_indoor_tail_workspace_bytes.restype = i64
    L.ivx_indoor_tail_get_bboxes.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]
    L.ivx_conv_winograd_set_variant.argtypes = [i32, i32]
    L.ivx_ubench_mfma.argtypes = [i32, vp, i64, C.POINTER(C.c_double), vp]
    L.ivx_ubench_copy.argtypes = [vp, vp, i64, C.POINTER(C.c_double), vp]
    L.ivx_model_trace_count.argtypes = [vp]
    L.ivx_model_trace_count.restype = i32
    L.ivx_model_trace_read.argtypes = [vp, i32, C.POINTER(TraceRec)]
    # ctypes' default restype (c_int) is the int status every other entry point returns; the 64-bit queries must have been
    # declared explicitly above (a missing one would silently truncate)
    wide = [n for n in EXPORTS if n.endswith(('_workspace_bytes', '_weight_elems')) and getattr(L, n).restype is not C.c_int64]
    assert not wide, f'int64-returning entry points without an explicit restype: {wide}'
    _lib = L
    return L


def check(rc, what=''):
    if rc != 0:
        msg = lib().ivx_last_error().decode('utf-8', 'replace')
        if rc == -1:
            raise ValueError(f'{what}: {msg}')
        raise IvxError(f'{what}: status {rc}: {msg}')
