"""ctypes loader for libp2pb_hip.so (the C ABI declared in include/p2pb_hip.h).

There is NO CPU fallback: if the library is missing, was not built for gfx950, or a tensor is not on
a HIP device, the call raises. (tests/ check this; the CPU oracle under oracle/ is never imported here.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P2PB_LIB_PATH") or os.path.join(_HERE, "libp2pb_hip.so")  # override: kernel experiments

# every symbol include/p2pb_hip.h declares (tests/test_abi.py checks the two lists agree)
SYMBOLS = [
    "p2pb_version", "p2pb_target_arch", "p2pb_set_split_terms", "p2pb_set_split_terms_thread", "p2pb_get_split_terms", "p2pb_set_deterministic", "p2pb_get_deterministic", "p2pb_voxel_coords", "p2pb_avg_voxelize_ws_bytes",
    "p2pb_avg_voxelize_forward", "p2pb_avg_voxelize_backward", "p2pb_avg_voxelize_cl_gather_split", "p2pb_conv3d_presplit",
    "p2pb_conv3d_k3_forward_compact_pre", "p2pb_trilinear_devoxelize_forward",
    "p2pb_trilinear_devoxelize_backward", "p2pb_ball_query", "p2pb_grouping_forward", "p2pb_grouping_backward", "p2pb_group_concat", "p2pb_group_sub", "p2pb_three_interpolate_add", "p2pb_group_sub_stats_floats", "p2pb_group_sub_stats_slots", "p2pb_group_sub_stats",
    "p2pb_gather_features_forward", "p2pb_gather_features_backward", "p2pb_furthest_point_sampling",
    "p2pb_fps_coop_ws_bytes", "p2pb_furthest_point_sampling_coop", "p2pb_point_face_dist", "p2pb_face_point_dist", "p2pb_knn_points_ws_bytes", "p2pb_knn_points", "p2pb_three_nn_interpolate_forward", "p2pb_three_nn_interpolate_backward", "p2pb_three_nn", "p2pb_three_nn_cells", "p2pb_three_nn_cells_ws_bytes",
    "p2pb_three_interpolate", "p2pb_chamfer_forward",
    "p2pb_chamfer_backward", "p2pb_approxmatch_forward", "p2pb_matchcost_forward", "p2pb_matchcost_backward",
    "p2pb_auction_forward", "p2pb_auction_backward", "p2pb_conv3d_k3_packed_floats", "p2pb_conv3d_k3_pack_weights",
    "p2pb_conv3d_k3_split_packed_bytes", "p2pb_conv3d_k3_pack_weights_split",
    "p2pb_conv3d_k3_stats_floats", "p2pb_conv3d_k3_forward", "p2pb_conv3d_k3_forward_ex",
    "p2pb_conv3d_k3_far_field", "p2pb_conv3d_k3_far_field_gn", "p2pb_pvconv_tail", "p2pb_minmax_act_pool_gn", "p2pb_conv3d_active_lists", "p2pb_conv3d_k3_forward_compact", "p2pb_conv3d_brick_lists", "p2pb_conv3d_k3_forward_sparse", "p2pb_gn_affine_params", "p2pb_se_gate_affine",
    "p2pb_trilinear_devoxelize_affine", "p2pb_avg_voxelize_cl_forward", "p2pb_voxel_sort", "p2pb_avg_voxelize_cl_gather", "p2pb_trilinear_devoxelize_cl_affine", "p2pb_pointwise_packed_floats", "p2pb_pointwise_pack_weights",
    "p2pb_pointwise_stats_floats", "p2pb_pointwise_conv_forward", "p2pb_affine_act", "p2pb_affine_act_max",
    "p2pb_pointwise_split_packed_bytes", "p2pb_pointwise_pack_weights_split", "p2pb_pointwise_pool_supported", "p2pb_pointwise_minmax_floats", "p2pb_pointwise_conv_pool_forward",
    "p2pb_minmax_act", "p2pb_linear_attention_forward", "p2pb_linear_attention_backward",
    "p2pb_approxmatch_temp_floats", "p2pb_approxmatch_forward_ws", "p2pb_pointwise_conv_pool_gather", "p2pb_chamfer_ws_bytes", "p2pb_chamfer_forward_ws", "p2pb_radius_count", "p2pb_radius_fill", "p2pb_merge_accumulate", "p2pb_merge_finish", "p2pb_gn_affine_params_ex", "p2pb_norm_act_backward", "p2pb_norm_act_backward_ex", "p2pb_grouping_backward_pitched", "p2pb_three_nn_interpolate_backward_pitched", "p2pb_affine_act_train", "p2pb_conv3d_k3_wgrad_ws_floats", "p2pb_conv3d_k3_wgrad", "p2pb_conv3d_k3_wgrad_occ_ws_floats", "p2pb_conv3d_k3_wgrad_occ", "p2pb_pointwise_wgrad_ws_floats", "p2pb_pointwise_wgrad",
    "p2pb_debug_pointwise_form", "p2pb_linear_rows", "p2pb_gn_finisher_arm", "p2pb_gn_finisher_armed", "p2pb_gn_finisher_disarm", "p2pb_fps_grid_ws_bytes", "p2pb_furthest_point_sampling_grid",
    "p2pb_optim_entry_bytes", "p2pb_optim_chunk", "p2pb_optim_clip_adam_step",
    "p2pb_conv3d_k3_pack_weights_split_adjoint", "p2pb_pointwise_pack_weights_adjoint", "p2pb_pointwise_pack_weights_split_adjoint",
    "p2pb_conv3d_k3_pack_weights_split_amax", "p2pb_pointwise_pack_weights_split_amax",
    "p2pb_se_gate_forward", "p2pb_se_gate_backward", "p2pb_row_max_forward", "p2pb_row_max_backward",
]

ABI_VERSION = 7  # include/p2pb_hip.h P2PB_ABI_VERSION this binding was written against (tests/test_abi.py compares the two)

_lib = None


class P2PBError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P2PBError(
                f"{LIB_PATH} not found: build it with `python -m p2p_bridge_amd.build` "
                "(p2p_bridge_amd has no CPU / eager fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.p2pb_target_arch.restype = ctypes.c_char_p
        _lib.p2pb_avg_voxelize_ws_bytes.restype = ctypes.c_size_t
        _lib.p2pb_knn_points_ws_bytes.restype = ctypes.c_size_t
        _lib.p2pb_fps_grid_ws_bytes.restype = ctypes.c_size_t
        _lib.p2pb_three_nn_cells_ws_bytes.restype = ctypes.c_size_t
        _lib.p2pb_fps_coop_ws_bytes.restype = ctypes.c_size_t
        _lib.p2pb_conv3d_k3_packed_floats.restype = ctypes.c_size_t
        _lib.p2pb_conv3d_k3_split_packed_bytes.restype = ctypes.c_size_t
        _lib.p2pb_conv3d_k3_stats_floats.restype = ctypes.c_size_t
        _lib.p2pb_pointwise_packed_floats.restype = ctypes.c_size_t
        _lib.p2pb_pointwise_stats_floats.restype = ctypes.c_size_t
        _lib.p2pb_pointwise_minmax_floats.restype = ctypes.c_size_t
        _lib.p2pb_group_sub_stats_floats.restype = ctypes.c_size_t
        _lib.p2pb_pointwise_split_packed_bytes.restype = ctypes.c_size_t
        _lib.p2pb_conv3d_k3_wgrad_ws_floats.restype = ctypes.c_size_t
        _lib.p2pb_conv3d_k3_wgrad_occ_ws_floats.restype = ctypes.c_size_t
        _lib.p2pb_chamfer_ws_bytes.restype = ctypes.c_size_t
        _lib.p2pb_approxmatch_temp_floats.restype = ctypes.c_size_t
        _lib.p2pb_pointwise_wgrad_ws_floats.restype = ctypes.c_size_t
        _lib.p2pb_optim_entry_bytes.restype = ctypes.c_size_t
        have = _lib.p2pb_version() if hasattr(_lib, "p2pb_version") else None
        if have != ABI_VERSION:
            bad, _lib = _lib, None
            del bad
            raise P2PBError(f"{LIB_PATH} is ABI version {have}, this package binds version {ABI_VERSION} (include/p2pb_hip.h): "
                            "rebuild it with `python -m p2p_bridge_amd.build`")
        for s in SYMBOLS:
            getattr(_lib, s)  # AttributeError here = stale library
        # products per split operand pair (include/p2pb_hip.h; fused.conv_math / set_conv_math)
        _lib.p2pb_set_split_terms(6 if os.environ.get("P2PB_CONV_MATH") in ("bf16x6", "fp32") else 16)
    return _lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def check(t, dtype, name):
    """Same preconditions as the reference's CHECK_CUDA / CHECK_CONTIGUOUS / CHECK_IS_* (PN2/utils.hpp:7-18)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {'float' if dtype == torch.float32 else 'int'} tensor")


def call(fn_name, *args):
    rc = getattr(lib(), fn_name)(*args)
    if rc != 0:
        raise P2PBError(f"{fn_name} failed with code {rc}")
