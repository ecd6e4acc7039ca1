"""ctypes binding of libdh3d_hip.so (the C ABI declared in include/dh3d_hip.h).

The product path has no CPU fallback: if the shared library is missing or a kernel reports an
error, this module raises.  PyTorch tensors are used only as device-memory containers (data_ptr)
and for the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DH3D_HIP_LIB") or os.path.join(_HERE, "libdh3d_hip.so")  # (env: dev builds of the library)

DH3D_OK = 0
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

_lib = None

c_fp = ctypes.c_void_p  # device pointers travel as void*
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float
c_ll = ctypes.c_longlong


class Epilogue(ctypes.Structure):
    """struct dh3d_epilogue"""
    _fields_ = [("pre_bias", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
                ("act", ctypes.c_int)]


# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "dh3d_version": [],
    "dh3d_abi_version": [],
    "dh3d_arch": [],
    "dh3d_source_hash": [],
    "dh3d_status_string": [c_int],
    "dh3d_stage_copy": [c_fp, c_fp, c_size_t, c_int, c_fp],
    "dh3d_knn_bruteforce": [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_knn_bruteforce_xyz": [c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_flex_conv_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_flex_conv_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp,
                           c_fp, c_fp, c_fp],
    "dh3d_flex_pool_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_flex_pool_bwd": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_conv_pointset_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_conv_pointset_bwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp,
                               c_fp],
    "dh3d_flex_conv_fwd_f64": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_flex_conv_bwd_f64": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp,
                               c_fp, c_fp, c_fp],
    "dh3d_flex_pool_fwd_f64": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_flex_pool_bwd_f64": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_conv_pointset_fwd_f64": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_conv_pointset_bwd_f64": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp,
                                   c_fp],
    "dh3d_flex_conv_fwd_workspace_bytes": [c_int, c_int, c_int, c_int, c_int, c_int],
    "dh3d_flex_conv_fwd_ws": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp, c_fp,
                              c_size_t, c_fp],
    "dh3d_flex_conv_bwd_workspace_bytes": [c_int, c_int, c_int, c_int, c_int, c_int],
    "dh3d_flex_conv_bwd_ws": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp,
                              c_fp, c_fp, c_fp, c_size_t, c_fp],
    "dh3d_flex_pool_fwd_workspace_bytes": [c_int, c_int, c_int, c_int],
    "dh3d_flex_pool_fwd_ws": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_size_t, c_fp],
    "dh3d_flex_conv_pm_bwd_workspace_bytes": [c_int, c_int, c_int, c_int],
    "dh3d_flex_conv_pm_bwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp,
                              c_size_t, c_fp, c_fp, c_fp, c_fp],
    "dh3d_gemm_is_split": [c_int, c_int, c_int, c_int, c_int],
    "dh3d_gemm_tn_f32": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_gemm_nn_f32": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_gemm_tn_f32_batched": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_gemm_nn_f32_batched": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_bn_colstats": [c_fp, c_ll, c_int, c_fp, c_int, c_fp, c_fp, c_fp],
    "dh3d_bn_finalize": [c_fp, c_fp, c_fp, c_fp, c_fp, c_float, c_float, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_scale_shift_act": [c_fp, c_ll, c_int, c_fp, c_fp, c_int, c_fp, c_fp],
    "dh3d_pairwise_sqdist": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_flex_pool_pm_bwd": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_se_gate_fwd": [c_fp, c_fp, c_ll, c_fp, c_fp],
    "dh3d_se_gate_bwd": [c_fp, c_fp, c_fp, c_ll, c_fp, c_fp, c_fp],
    "dh3d_relu_fwd": [c_fp, c_ll, c_fp, c_fp],
    "dh3d_relu_bwd": [c_fp, c_fp, c_ll, c_fp, c_fp],
    "dh3d_pointset_sum_pm": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_scale_shift_act_res": [c_fp, c_ll, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp],
    "dh3d_row_logit_sigmoid": [c_fp, c_ll, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_bn_bwd_sums": [c_fp, c_fp, c_fp, c_fp, c_ll, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int,
                         c_fp, c_fp, c_fp, c_fp],
    "dh3d_bn_bwd_finalize": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp],
    "dh3d_bn_bwd_apply": [c_fp, c_fp, c_fp, c_fp, c_ll, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_int,
                          c_fp, c_fp],
    "dh3d_netvlad_assign_rows": [c_fp, c_ll, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_ll, c_fp],
    "dh3d_idw_weights": [c_fp, c_ll, c_fp, c_fp],
    "dh3d_context_gate_fwd": [c_fp, c_fp, c_ll, c_fp, c_fp],
    "dh3d_context_gate_bwd": [c_fp, c_fp, c_fp, c_ll, c_fp, c_fp, c_fp],
    "dh3d_netvlad_assign_rows_bwd": [c_fp, c_ll, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_l2norm_rows_bwd": [c_fp, c_fp, c_ll, c_int, c_float, c_fp, c_fp],
    "dh3d_transpose32": [c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_colsum_f32": [c_fp, c_ll, c_int, c_int, c_fp, c_fp],
    "dh3d_farthest_point_sample": [c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "dh3d_farthest_point_sample_mode": [c_int, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_fp],
    "dh3d_group_point_fwd": [c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "dh3d_group_point_bwd": [c_int, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "dh3d_three_nn": [c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_three_interpolate_fwd": [c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_three_interpolate_bwd": [c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_spatial_sort": [c_fp, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_knn_sorted": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_spatial_sort_cells": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp],
    "dh3d_knn_grid": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_fps_sorted": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_three_nn_sorted": [c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_fps_sorted_xyz": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_fps_sorted_ordered": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_fps_sorted_cloud": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_pack_weight": [c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_pack_flex_weight": [c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_flex_conv_pm_gather_fwd": [c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int,
                                     ctypes.POINTER(Epilogue), c_fp, c_fp],
    "dh3d_flex_conv_pm_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int,
                              ctypes.POINTER(Epilogue), c_fp, c_fp],
    "dh3d_flex_conv_pm_post_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp,
                                   c_fp, c_int, c_fp, c_fp],
    "dh3d_flex_conv_pm_tile_x6_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp,
                                   c_fp, c_int, c_fp, c_fp],
    "dh3d_pack_flex_weight_x3": [c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_flex_conv_pm_x6_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int,
                                 ctypes.POINTER(Epilogue), c_fp, c_fp],
    "dh3d_flex_conv_pm_x6_fwd_r": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int,
                                   ctypes.POINTER(Epilogue), c_int, c_fp, c_fp],
    "dh3d_upsample_linear_pm_x6_fwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int,
                                       ctypes.POINTER(Epilogue), c_fp, c_fp, c_fp],
    "dh3d_upsample_linear_l2cat_pm_x6_fwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int,
                                             ctypes.POINTER(Epilogue), c_fp, c_fp, c_float, c_fp, c_fp],
    "dh3d_upsample_linear_shortcut_pm_x6_fwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int,
                                                c_fp, c_int, ctypes.POINTER(Epilogue), ctypes.POINTER(Epilogue),
                                                c_fp, c_float, c_fp, c_fp],
    "dh3d_interp_combine_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp, c_fp,
                                c_float, c_fp, c_fp],
    "dh3d_local_tail_fused_fwd": [c_fp, c_fp, c_fp, c_fp, ctypes.POINTER(Epilogue), ctypes.POINTER(Epilogue), c_fp, c_fp,
                                  c_fp, c_fp, c_float, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_linear_slices_pm_x6_fwd": [c_fp, c_int, c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_interp_head_fwd": [c_fp, c_int, c_fp, c_fp, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp, c_float,
                             c_fp, c_fp],
    "dh3d_interp_head_sorted_fwd": [c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp,
                                    c_float, c_fp, c_fp],
    "dh3d_interp_head_sorted_fwd_dev": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, ctypes.POINTER(Epilogue),
                                        c_fp, c_fp, c_fp, c_fp],
    "dh3d_three_interpolate_bwd_sorted": [c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_bn_small_fwd": [c_fp, c_int, c_int, c_fp, c_fp, c_float, c_float, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_bn_small_bwd": [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_quadruplet_loss": [c_fp, c_int, c_int, c_int, c_int, c_float, c_float, c_fp, c_fp, c_fp],
    "dh3d_vlad_normalize_fwd": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_float, c_fp, c_fp, c_fp, c_fp],
    "dh3d_vlad_normalize_bwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_float, c_fp, c_fp, c_fp, c_fp],
    "dh3d_netvlad_commuted_fwd_stats": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_netvlad_commuted_fwd_assign": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp,
                                         c_fp, c_fp, c_fp],
    "dh3d_netvlad_commuted_bwd_sums": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                                       c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_netvlad_commuted_bwd_apply": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp,
                                        c_fp, c_fp, c_fp],
    "dh3d_interp_scatter_scaled": [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_bn_finalize_parts": [c_fp, c_int, c_fp, c_fp, c_fp, c_float, c_float, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp,
                               c_fp],
    "dh3d_bn_bwd_finalize_parts": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp],
    "dh3d_sigmoid_bwd": [c_fp, c_fp, c_fp, c_int, c_ll, c_fp, c_fp, c_fp],
    "dh3d_interp_bn_colstats": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_interp_bn_bwd_sums": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp,
                                c_fp, c_fp, c_fp, c_fp],
    "dh3d_interp_bn_bwd_apply": [c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp,
                                 c_fp, c_fp, c_fp, c_fp],
    "dh3d_linear_pm_x6_fwd": [c_fp, c_int, c_fp, c_int, c_fp, c_int, c_int, ctypes.POINTER(Epilogue), c_fp, c_fp, c_fp],
    "dh3d_se_res_pm_packed_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_se_res_pool_pm_packed_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp],
    "dh3d_se_res_pool_conv_pm_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp,
                                     ctypes.POINTER(Epilogue), c_int, c_fp, c_fp],
    "dh3d_se_res_pool_conv_tails_pm_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                           ctypes.POINTER(Epilogue), c_fp, c_fp, ctypes.POINTER(Epilogue), c_fp, c_fp,
                                           ctypes.POINTER(Epilogue), c_fp, c_fp],
    "dh3d_flex_pool_pm_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp],
    "dh3d_flex_avg_pm_fwd": [c_fp, c_fp, c_int, c_int, c_int, c_int, c_float, c_fp, c_fp],
    "dh3d_conv_pointset_pm_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, ctypes.POINTER(Epilogue),
                                  c_fp, c_fp],
    "dh3d_conv_pointset_pool_pm_fwd": [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, ctypes.POINTER(Epilogue),
                                       c_fp, c_fp, c_fp],
    "dh3d_linear_pm_fwd": [c_fp, c_int, c_fp, c_int, c_fp, c_int, c_int, ctypes.POINTER(Epilogue), c_fp, c_fp,
                           c_fp],
    "dh3d_se_res_pm_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_three_interpolate_idw_fwd": [c_int, c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_l2norm_concat_fwd": [c_fp, c_int, c_int, c_float, c_fp, c_int, c_fp, c_fp],
    "dh3d_mlp_head_pm_fwd": [c_fp, c_int, c_int, c_fp, c_int, ctypes.POINTER(Epilogue), c_fp, c_float, c_fp,
                             c_fp],
    "dh3d_pack_weight_x3": [c_fp, c_int, c_int, c_fp, c_fp],
    "dh3d_mlp_head_pm_x6_fwd": [c_fp, c_int, c_int, c_fp, c_int, ctypes.POINTER(Epilogue), c_fp, c_float, c_fp,
                                c_fp],
    "dh3d_netvlad_workspace_bytes": [c_int, c_int, c_int, c_int],
    "dh3d_netvlad_aggregate_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp,
                                   c_size_t, c_fp, c_fp],
    "dh3d_netvlad_head_workspace_bytes": [c_int, c_int, c_int],
    "dh3d_global_tail_fwd": [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp,
                             c_float, c_fp, c_fp, c_fp, c_fp, c_fp],
    "dh3d_netvlad_tail_workspace_bytes": [c_int, c_int, c_int, c_int],
    "dh3d_netvlad_tail_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_float,
                              c_fp, c_size_t, c_fp, c_fp],
    "dh3d_walk_plan_bytes": [c_int, c_int],
    "dh3d_walk_plan": [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_fp],
    "dh3d_global_walk_planned_fwd": [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, ctypes.POINTER(Epilogue),
                                     c_fp, c_float, c_fp, c_fp, c_fp, c_fp, c_int, c_fp],
    "dh3d_netvlad_tail_assign_fwd": [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int,
                                     c_float, c_fp, c_size_t, c_fp, c_fp],
    "dh3d_global_walk_fwd": [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, ctypes.POINTER(Epilogue), c_fp,
                             c_float, c_fp, c_fp, c_fp, c_fp, c_int, c_fp],
    "dh3d_netvlad_fused_workspace_bytes": [c_int, c_int, c_int, c_int, c_int],
    "dh3d_netvlad_fused_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                               c_int, c_int, c_float, c_fp, c_size_t, c_fp, c_fp],
    "dh3d_netvlad_head_fwd": [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_float, c_fp,
                              c_size_t, c_fp, c_fp],
}
_RESTYPES = {
    "dh3d_arch": ctypes.c_char_p,
    "dh3d_source_hash": ctypes.c_char_p,
    "dh3d_status_string": ctypes.c_char_p,
    "dh3d_netvlad_workspace_bytes": c_size_t,
    "dh3d_netvlad_head_workspace_bytes": c_size_t,
    "dh3d_netvlad_fused_workspace_bytes": c_size_t,
    "dh3d_netvlad_tail_workspace_bytes": c_size_t,
    "dh3d_walk_plan_bytes": c_size_t,
    "dh3d_flex_conv_fwd_workspace_bytes": c_size_t,
    "dh3d_flex_conv_bwd_workspace_bytes": c_size_t,
    "dh3d_flex_pool_fwd_workspace_bytes": c_size_t,
    "dh3d_flex_conv_pm_bwd_workspace_bytes": c_size_t,
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
ABI_VERSION = 4  # include/dh3d_hip.h DH3D_ABI_VERSION: the semantics (not just the signatures) this file was written against


def tree_source_hash():
    """The hash dh3d_source_hash() must return for a library built from the sources of THIS tree (csrc/Makefile SRC_HASH:
    sha256 over basename + newline + content of csrc/*.hip, csrc/*.h, csrc/Makefile and include/dh3d_hip.h in the
    Makefile's $(sort ...) order, first 16 hex digits).  None when the sources are not there (an installed binary)."""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    rel = [os.path.basename(f) for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))]
    rel += ["../../include/dh3d_hip.h", "Makefile"]
    if not os.path.isfile(os.path.join(csrc, "Makefile")):
        return None
    h = hashlib.sha256()
    for r in sorted(set(rel)):   # (make's $(sort) is a plain byte-wise sort, as is Python's on ASCII names)
        h.update((os.path.basename(r) + "\n").encode())
        with open(os.path.join(csrc, r), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def lib():
    """Load libdh3d_hip.so (once).  Raises if it was not built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                "dh3d_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dh3d_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the header and the library disagree
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        if handle.dh3d_abi_version() != ABI_VERSION:
            raise RuntimeError("dh3d_amd: %s implements ABI version %d, this binding was written against %d "
                               "(include/dh3d_hip.h DH3D_ABI_VERSION): rebuild with `make -C dh3d_amd/csrc`"
                               % (LIB_PATH, handle.dh3d_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream  # (plain ints / None: argtypes convert them, no ctypes object per argument)


def ptr(t):
    return t.data_ptr() if t is not None else None


def check(status, what):
    if status != DH3D_OK:
        msg = lib().dh3d_status_string(status).decode()
        if status in (1, 2):
            raise ValueError("%s: %s" % (what, msg))
        raise RuntimeError("%s: %s" % (what, msg))


def make_epilogue(pre_bias=None, scale=None, shift=None, act=ACT_NONE):
    ep = Epilogue()
    ep.pre_bias = pre_bias.data_ptr() if pre_bias is not None else None
    ep.scale = scale.data_ptr() if scale is not None else None
    ep.shift = shift.data_ptr() if shift is not None else None
    ep.act = act
    return ep


def require_cuda_f32(t, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s must be a torch.Tensor" % name)
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have rank %d, got shape %s" % (name, ndim, tuple(t.shape)))
    return t.contiguous()


def require_cuda_float(t, name, ndim=None, like=None):
    """float32 or float64 (the six reference-layout flex operators accept both, as the reference's registrations do);
    `like`: a tensor whose dtype this one has to share."""
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s must be a torch.Tensor" % name)
    if t.dtype not in (torch.float32, torch.float64):
        raise ValueError("%s must be float32 or float64, got %s" % (name, t.dtype))
    if like is not None and t.dtype != like.dtype:
        raise ValueError("%s must have the dtype of the features (%s), got %s" % (name, like.dtype, t.dtype))
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have rank %d, got shape %s" % (name, ndim, tuple(t.shape)))
    return t.contiguous()


def require_cuda_i32(t, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s must be a torch.Tensor" % name)
    if t.dtype != torch.int32:
        raise ValueError("%s must be int32, got %s" % (name, t.dtype))
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have rank %d, got shape %s" % (name, ndim, tuple(t.shape)))
    return t.contiguous()
