"""ctypes loader for libnbp_hip.so (the C ABI declared in include/nbp_hip.h).

The product path has no fallback: ``lib()`` raises if the shared object is missing or does
not export a declared symbol.  Building is explicit (``python -m nextbestpath_amd.build`` or
``__graft_entry__.build()``); importing this module never compiles anything.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnbp_hip.so")

_vp, _i, _ll, _f, _sz, _d = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_double

# name -> (restype, argtypes).  Kept in the order of include/nbp_hip.h.
SIGNATURES = {
    "nbp_abi_version": (_i, []),
    "nbp_device_info": (_i, [C.c_char_p, _i, C.POINTER(_i)]),
    "nbp_tuning_active": (_i, []),
    "nbp_tuning_report": (_i, [C.c_char_p, _i]),
    "nbp_tile_kernel_symbol": (_i, [_i, C.c_char_p, _i]),
    "nbp_packed_weights_bytes": (_sz, []),
    "nbp_pack_weights": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp, C.POINTER(_vp)]),
    "nbp_free_weights": (None, [_vp]),
    "nbp_forward_workspace_bytes": (_sz, [_i, _i]),
    "nbp_forward_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "nbp_forward_flops": (_d, [_i, _i]),
    "nbp_conv_igemm_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp,
                                _sz, _vp]),
    "nbp_conv_igemm_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "nbp_pack_conv_weight": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "nbp_conv_first_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "nbp_maxpool2_nhwc_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_psi_gate_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _ll, _vp, _vp]),
    "nbp_final_1x1_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "nbp_nchw_to_nhwc_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_nhwc_to_nchw_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_transform_points_f32": (_i, [_vp, _ll, _f, _f, _f, _vp, _vp]),
    "nbp_map_points_to_imgs_f32": (_i, [_vp, _i, _ll, _i, _i, _f, _f, _vp, _vp]),
    "nbp_point_position_i64": (_i, [_vp, _ll, _i, _i, _f, _f, _vp, _vp]),
    "nbp_map_accumulate_f32": (_i, [_vp, _ll, _vp, _f, _f, _f, C.POINTER(_f), _i, _f, _f, _i, _f, _f, _vp, _vp]),
    "nbp_coverage_count_planned_batch_f32": (_i, [_i, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nbp_unproject_append_shaded_batch_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _d, _vp, _f, _vp, _vp, _vp, _vp,
                                                   _vp, _vp, _sz, _vp]),
    "nbp_raster_zface_batch_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "nbp_replan_batch_f32": (_i, [_i, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nbp_step_maps_batch_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nbp_step_maps_f32": (_i, [_vp, _ll, _vp, _f, _f, _f, C.POINTER(_f), _i, _f, _f, _i, _f, _f, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "nbp_cloud_bins_bytes": (_sz, [C.POINTER(_f), C.POINTER(_f), _ll]),
    "nbp_cloud_bins_geometry": (_i, [C.POINTER(_f), C.POINTER(_f), _ll, C.POINTER(_i), C.POINTER(_f)]),
    "nbp_cloud_bins_init": (_i, [_vp, _sz, C.POINTER(_f), C.POINTER(_f), _ll, _vp]),
    "nbp_step_maps_binned_f32": (_i, [_vp, _i, _vp, _ll, _vp, _f, _f, _f, C.POINTER(_f), _i, _f, _f, _i, _f, _f, _vp, _i, _vp, _i, _vp, _vp,
                                      _vp]),
    "nbp_step_maps_binned_batch_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nbp_unproject_workspace_bytes": (_sz, [_i, _i, _i]),
    "nbp_unproject_append_f32": (_i, [_vp, _vp, C.POINTER(_f), _i, _i, _i, _f, _f, _d, C.c_uint, _vp, _vp, _vp, _ll, _vp, _sz,
                                      _vp]),
    "nbp_raster_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "nbp_raster_zbuf_f32": (_i, [_vp, _i, _vp, _i, C.POINTER(_f), _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "nbp_segments_hit_mesh_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "nbp_axis_ray_counts_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "nbp_carve_update_f32": (_i, [_vp, _i, _vp, _vp, C.POINTER(_f), _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "nbp_view_state_update_f32": (_i, [_vp, _i, _vp, _vp, _f, C.POINTER(_f), _i, _i, _i, _vp, _vp]),
    "nbp_view_gain_i32": (_i, [_vp, _i, _vp, _vp, C.POINTER(_f), C.POINTER(_f), _i, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    "nbp_carve_view_update_f32": (_i, [_vp, _i, _vp, _vp, C.POINTER(_f), _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp,
                                       C.POINTER(_f), _i, _i, _f, _vp, _vp, _vp, _vp]),
    "nbp_append_points_f32": (_i, [_vp, _ll, C.POINTER(_f), _i, _vp]),
    "nbp_perm_index_host": (C.c_uint, [C.c_uint, C.c_uint, C.c_uint]),
    "nbp_colreduce_workspace_bytes": (_sz, [_ll, _i]),
    "nbp_bn_train_forward_f32": (_i, [_vp, _ll, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nbp_bn_train_forward_amax_f32": (_i, [_vp, _ll, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nbp_bn_train_forward_stat4_f32": (_i, [_vp, _ll, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "nbp_bn_train_backward_stat_f32": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nbp_bn_backward_fuses": (_i, [_i]),
    "nbp_bn_train_backward_fused_f32": (_i, [_vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nbp_bn_train_backward_f32": (_i, [_vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nbp_colsum_f32": (_i, [_vp, _vp, _ll, _i, _vp, _vp, _sz, _vp]),
    "nbp_elementwise_f32": (_i, [_i, _vp, _vp, _ll, _vp, _vp]),
    "nbp_rowscale_f32": (_i, [_vp, _vp, _ll, _i, _vp, _vp]),
    "nbp_rowdot_f32": (_i, [_vp, _vp, _i, _ll, _i, _vp, _vp]),
    "nbp_rowscale_backward_f32": (_i, [_vp, _ll, _vp, _vp, _ll, _i, _vp, _vp, _vp]),
    "nbp_outer_f32": (_i, [_vp, _vp, _ll, _i, _vp, _vp]),
    "nbp_maxpool2_backward_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_sum2x2_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_slice_channels_f32": (_i, [_vp, _ll, _i, _i, _i, _vp, _vp]),
    "nbp_pad_channels_f32": (_i, [_vp, _ll, _i, _i, _vp, _vp]),
    "nbp_pack_conv_weight_padded": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nbp_pack_conv_weight_dgrad": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nbp_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "nbp_conv_wgrad_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "nbp_conv_wgrad_split_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nbp_gather_values_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_scatter_values_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "nbp_loss_f32": (_i, [_i, _vp, _vp, _ll, _f, _vp, _vp, _vp, _sz, _vp]),
    "nbp_fuse_obstacle_f32": (_i, [_vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "nbp_score_candidates_f32": (_i, [_vp, _i, _f, _f, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "nbp_edges_blocked_u8": (_i, [_vp, _i, _f, _f, _f, _f, _vp, _vp, _i, _vp, _vp]),
    "nbp_plan_search_host": (_i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _i, _f, _f,
                                  _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "nbp_coverage_workspace_bytes": (_sz, [C.POINTER(_f), C.POINTER(_f), _f, _ll]),
    "nbp_coverage_count_f32": (_i, [_vp, _i, _vp, _ll, _vp, _ll, C.c_uint, _f, C.POINTER(_f), C.POINTER(_f), _vp, _vp,
                                    _vp, _sz, _vp]),
}



class LayerTiming(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("flops", _d), ("ms", _f), ("tile", _i), ("split_k", _i), ("M", _ll),
                ("N", _i), ("K", _i)]


SIGNATURES["nbp_forward_timed_f32"] = (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp, C.POINTER(LayerTiming), _i,
                                            C.POINTER(_i)])
SIGNATURES["nbp_forward_timed_bf16"] = SIGNATURES["nbp_forward_timed_f32"]
SIGNATURES["nbp_pack_weights_bf16"] = SIGNATURES["nbp_pack_weights"]
SIGNATURES["nbp_packed_weights_bytes_bf16"] = SIGNATURES["nbp_packed_weights_bytes"]
SIGNATURES["nbp_pack_upconv_weight_bf16"] = (_i, [_vp, _i, _i, _vp, _vp])
SIGNATURES["nbp_forward_workspace_bytes_bf16"] = SIGNATURES["nbp_forward_workspace_bytes"]
SIGNATURES["nbp_forward_bf16"] = SIGNATURES["nbp_forward_f32"]
SIGNATURES["nbp_packed_weights_bytes_split"] = SIGNATURES["nbp_packed_weights_bytes"]
SIGNATURES["nbp_pack_weights_split"] = SIGNATURES["nbp_pack_weights"]
SIGNATURES["nbp_forward_workspace_bytes_split"] = SIGNATURES["nbp_forward_workspace_bytes"]
SIGNATURES["nbp_forward_split_f32"] = SIGNATURES["nbp_forward_f32"]
SIGNATURES["nbp_forward_timed_split_f32"] = SIGNATURES["nbp_forward_timed_f32"]
SIGNATURES["nbp_pack_conv_weight_split"] = (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_amax_f32"] = (_i, [_vp, _ll, _vp, _vp])
SIGNATURES["nbp_pack_upconv_weight_split"] = (_i, [_vp, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_upconv3x3_split_f32"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp])
SIGNATURES["nbp_conv_split_workspace_bytes"] = SIGNATURES["nbp_conv_igemm_workspace_bytes"]
SIGNATURES["nbp_conv_split_planned_workspace_bytes"] = (_sz, [_i, _i, _i, _i, _i, _i, _vp])
SIGNATURES["nbp_conv_split_planned_workspace_bytes_k"] = (_sz, [_i, _i, _i, _i, _i, _i, _i, _vp])
SIGNATURES["nbp_conv3x3_split_f32"] = (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp,
                                            _sz, _vp])
SIGNATURES["nbp_pack_upconv_weight_split_dgrad"] = (_i, [_vp, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_upconv_split_dgrad_workspace_bytes"] = (_sz, [_i, _i, _i, _i, _i])
SIGNATURES["nbp_upconv3x3_split_dgrad_f32"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_upconv_wgrad_split_workspace_bytes"] = (_sz, [_i, _i, _i, _i, _i])
SIGNATURES["nbp_upconv_wgrad_split_f32"] = (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_pack_conv_weight_split_prezeroed"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_pack_conv_weight_split_dgrad_known"] = (_i, [_vp, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_prepack_weights_split"] = (_i, [_vp, _i, _vp, _vp])
SIGNATURES["nbp_prepack_desc_bytes"] = (_i, [])
SIGNATURES["nbp_rowscale_amax_f32"] = (_i, [_vp, _vp, _ll, _i, _vp, _vp, _vp])
SIGNATURES["nbp_conv_first_linear_f32"] = (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
SIGNATURES["nbp_conv_first_wgrad_workspace_bytes"] = (_sz, [])
SIGNATURES["nbp_conv_first_wgrad_f32"] = (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_conv1x1_split_f32"] = (_i, [_vp, _i, _ll, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp])
SIGNATURES["nbp_pack_conv1x1_weight_split_dgrad"] = (_i, [_vp, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_pack_conv_weight_split_dgrad"] = (_i, [_vp, _i, _i, _i, _vp, _vp, _vp])
SIGNATURES["nbp_conv_bn_part_rows"] = (_i, [_i, _i, _i])
SIGNATURES["nbp_conv3x3_split_bn_f32"] = (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp,
                                               _sz, _vp, C.POINTER(_i), _vp])
SIGNATURES["nbp_upconv3x3_split_bn_f32"] = (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp,
                                                 C.POINTER(_i), _vp])
SIGNATURES["nbp_bn_train_forward_part4_f32"] = (_i, [_vp, _ll, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp])
SIGNATURES["nbp_conv_igemm_bf16"] = SIGNATURES["nbp_conv_igemm_f32"]
SIGNATURES["nbp_conv_igemm_bf16_workspace_bytes"] = SIGNATURES["nbp_conv_igemm_workspace_bytes"]
SIGNATURES["nbp_pack_conv_weight_bf16"] = SIGNATURES["nbp_pack_conv_weight"]
SIGNATURES["nbp_f32_to_bf16"] = (_i, [_vp, _ll, _vp, _vp])
SIGNATURES["nbp_bf16_to_f32"] = (_i, [_vp, _ll, _vp, _vp])
_fpp, _ipp = C.POINTER(_f), C.POINTER(_i)
SIGNATURES["nbp_scene_fill_workspace_bytes"] = (_sz, [_fpp, _ipp, _i, _ll, _d])
SIGNATURES["nbp_scene_fill_cells_f32"] = (_i, [_vp, _ll, _vp, _fpp, _ipp, _i, _d, _i, C.c_uint, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_scene_gather_f32"] = (_i, [_vp, _vp, _i, _i, _vp, _ll, _vp, _vp])
SIGNATURES["nbp_scene_coverage_workspace_bytes"] = (_sz, [_fpp, _ipp, _i, _d])
SIGNATURES["nbp_scene_coverage_f32"] = (_i, [_vp, _vp, _i, _vp, _vp, _i, _fpp, _ipp, _d, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_coverage_plan_bytes"] = (_sz, [_fpp, _fpp, _f, _i])
SIGNATURES["nbp_coverage_plan_workspace_bytes"] = (_sz, [_fpp, _fpp, _f, _i])
SIGNATURES["nbp_coverage_plan_build_f32"] = (_i, [_vp, _i, _f, _fpp, _fpp, _vp, _sz, _vp, _sz, _vp])
SIGNATURES["nbp_coverage_count_planned_f32"] = (_i, [_vp, _i, _f, _fpp, _fpp, _vp, _ll, _vp, _ll, C.c_uint, C.c_uint, _vp, _vp,
                                                     _vp])
SIGNATURES["nbp_points_in_fov_u8"] = (_i, [_vp, _i, _fpp, _i, _i, _i, _f, _f, _vp, _vp, _vp])
SIGNATURES["nbp_sample_points_f32"] = (_i, [_vp, _ll, _vp, _ll, C.c_uint, _vp, _vp, _vp])
SIGNATURES["nbp_raster_rgb_workspace_bytes"] = (_sz, [_i, _i, _i, _i])
SIGNATURES["nbp_raster_rgbz_f32"] = (_i, [_vp, _i, _vp, _i, _vp, _fpp, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_unproject_append_rgb_f32"] = (_i, [_vp, _vp, _vp, _fpp, _i, _i, _i, _f, _f, _d, C.c_uint, _vp, _vp, _vp, _vp, _ll,
                                                   _vp, _sz, _vp])
SIGNATURES["nbp_raster_zface_f32"] = (_i, [_vp, _i, _vp, _i, _fpp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_shade_image_f32"] = (_i, [_vp, _vp, _vp, _vp, _fpp, _i, _i, _i, _f, _f, _f, _vp, _vp])
SIGNATURES["nbp_unproject_append_shaded_f32"] = (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _fpp, _i, _i, _i, _f, _f, _d, C.c_uint, _f,
                                                      _vp, _vp, _vp, _vp, _ll, _vp, _sz, _vp])
SIGNATURES["nbp_mdb_open"] = (_i, [C.c_char_p, C.c_ulonglong, _i, C.POINTER(_vp)])
SIGNATURES["nbp_mdb_close"] = (_i, [_vp])
SIGNATURES["nbp_mdb_entries"] = (_ll, [_vp])
SIGNATURES["nbp_mdb_put"] = (_i, [_vp, C.c_char_p, _sz, C.c_char_p, _sz])
SIGNATURES["nbp_mdb_del"] = (_i, [_vp, C.c_char_p, _sz])
SIGNATURES["nbp_mdb_get"] = (_i, [_vp, C.c_char_p, _sz, _vp, _sz, C.POINTER(_sz)])
SIGNATURES["nbp_mdb_keys"] = (_i, [_vp, _vp, _sz, C.POINTER(_sz)])
SIGNATURES["nbp_mdb_stat"] = (_i, [_vp, C.POINTER(C.c_ulonglong)])
SIGNATURES["nbp_gate_mid_workspace_bytes"] = (_sz, [_ll, _i])
SIGNATURES["nbp_gate_mid_forward_f32"] = (_i, [_vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_gate_mid_backward_f32"] = (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp])
SIGNATURES["nbp_sum_n_f32"] = (_i, [_i, _vp, _vp, _ll, _i, _vp, _vp])
SIGNATURES["nbp_unproject_append_filed_f32"] = (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _fpp, _i, _i, _i, _f, _f, _d, C.c_uint, _f,
                                                     _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _i, _vp, _sz, _vp])
SIGNATURES["nbp_step_maps_prefiled_f32"] = SIGNATURES["nbp_step_maps_binned_f32"]
SIGNATURES["nbp_slice_obstacle_f32"] = (_i, [_vp, _vp, _i, _f, _f, _f, _i, _f, _f, _f, _vp, _vp])
SIGNATURES["nbp_slice_obstacle_fig_f32"] = (_i, [_vp, _vp, _i, _f, _f, _f, _i, _f, _f, _f, _f, _f, _f, _vp, _vp])

_lock = threading.Lock()
_lib = None


class NbpHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Returns the loaded library; raises (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NbpHipError(
                f"{LIB_PATH} is missing: build it with `python -m nextbestpath_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch ships its own libamdhip64; it has to be in the process first so that this library binds to the
        # SAME HIP runtime (device pointers and streams cross the boundary).  Loaded the other way round the
        # process ends up with two runtimes and the second one reports hipErrorNoDevice.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:  # pragma: no cover
                raise NbpHipError(f"libnbp_hip.so does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


_ERR = {-1: "NBP_E_ARG (bad argument)", -2: "NBP_E_WS (workspace too small)", -3: "NBP_E_SHAPE (unsupported shape)"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = _ERR.get(rc, f"hipError_t {rc}" if rc > 0 else f"error {rc}")
        raise NbpHipError(f"{what} failed: {msg}")


def ptr(t) -> int:
    """Device pointer of a contiguous torch tensor (0 for None)."""
    if t is None:
        return 0
    if not t.is_contiguous():
        raise ValueError("tensor passed to the HIP path must be contiguous")
    return t.data_ptr()


# ---- A/B switches of the host side: the same gate as the library's (csrc/nbp_tuning.cpp).  A switch keeps its default unless
# the process opted in with NBP_TUNING=1 AND sets the variable, so an inherited environment never changes results.
_knobs = {}
# switches that change the ARITHMETIC (results differ beyond reordering): a benchmark line measured with one of them set is
# not the headline configuration (bench.py refuses to call it `value`)
NUMERICS_KNOBS = {"NBP_CONV_PRECISION", "NBP_TRAIN_SPLIT", "NBP_TRAIN_WGRAD_SPLIT", "NBP_SPLIT_MAX_K", "NBP_SPLIT_MAX_K_SMALL", "NBP_GATE_PSI",
                  "NBP_CONV_HEAD", "NBP_BF16_PSI", "NBP_BF16_FUSE", "NBP_TRAIN_FUSE", "NBP_TRAIN_SPLIT_1X1", "NBP_TRAIN_CHAIN_BOUND",
                  "NBP_TRAIN_UP_DGRAD", "NBP_TRAIN_UP_WGRAD",
                  #  Conv1.conv.0 on the fp32 MFMA pipe instead of the split scheme -- it flipped a ReLU mask in
                  # test_full_network_training_step_vs_oracle; the fused AdamW rounds in another order than the foreach form
                  "NBP_TRAIN_FIRST_CONV", "NBP_TRAIN_FUSED_ADAMW",
                  # (round 6) the n-ary gradient sums add in another order than autograd's pairwise adds (the fused gate middle,
                  # NBP_TRAIN_GATE_FUSE, is bit-identical to the separate Functions: not listed)
                  "NBP_TRAIN_FANOUT"}
# bit-identical switches (NBP_TRAIN_PREPACK, NBP_STEP_OVERLAP, ...) are not listed here; effective_knobs() reports every switch
# that is off its default, numerics-affecting or not


def tuning_active() -> bool:
    return os.environ.get("NBP_TUNING") == "1"


def tune(name: str, default: str) -> str:
    """String-valued A/B switch `name`: `default` unless NBP_TUNING=1 and the variable is set.  Recorded for effective_knobs()."""
    v = default
    if tuning_active():
        e = os.environ.get(name)
        if e is not None and e != "":
            v = e
    _knobs[name] = (default, v)
    return v


def effective_knobs() -> dict:
    """{name: value} of every switch (host side and library) read so far whose value differs from its default."""
    out = {k: v for k, (d, v) in _knobs.items() if v != d}
    if _lib is not None:
        buf = C.create_string_buffer(4096)
        if _lib.nbp_tuning_report(buf, 4096) > 0:
            for item in buf.value.decode().split(","):
                if "=" in item:
                    k, v = item.split("=", 1)
                    out[k] = v
    return out


_raw_stream = None


def current_stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device (the fast private accessor: the public
    torch.cuda.current_stream() builds a Stream object, ~9 us per call and ~25 calls per exploration step)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
