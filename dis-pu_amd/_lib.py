"""ctypes binding of libdispu_hip.so (the C ABI declared in include/dispu_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, the op raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdispu_hip.so")

ABI_VERSION = 5       # include/dispu_hip.h: dispu_version()
ARITH_PLAIN = 0
ARITH_CONTRACT = 1
ARITH_PINNED_EXP = 2   # OR-able, approx_match only (bit-reproducible exp; parity mode)

_vp, _i, _sz, _l = C.c_void_p, C.c_int, C.c_size_t, C.c_long

# name -> (restype, argtypes); must list every symbol of include/dispu_hip.h (tests check this)
SIGNATURES = {
    "dispu_version": (_i, []),
    "dispu_error_string": (C.c_char_p, [_i]),
    "dispu_event_record": (_i, [_vp, _vp]),
    "dispu_stream_wait_event": (_i, [_vp, _vp]),
    "dispu_memset_async": (_i, [_vp, _i, _sz, _vp]),
    "dispu_fps_scratch_bytes": (_sz, [_i, _i, _i]),
    "dispu_fps": (_i, [_i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "dispu_fps_ws": (_i, [_i, _i, _i, _vp, _vp, _sz, _vp, _i, _vp]),
    "dispu_prob_sample": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_selection_sort": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_gather_point": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_gather_point_grad": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_query_ball": (_i, [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_group_point": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_group_point_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_knn_point": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_knn_feat": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_knn_feat_strided": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "dispu_knn_xyz": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_knn_feat_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dispu_knn_feat_strided_ws": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "dispu_knn_xyz_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "dispu_knn_xyz_ws": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "dispu_three_nn": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_three_interpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_three_interpolate_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_nn_distance": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_nn_distance_grad": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dispu_approx_match_scratch_bytes": (_sz, [_i, _i, _i]),
    "dispu_approx_match": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_approx_match_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "dispu_match_cost_scratch_bytes": (_sz, [_i, _i, _i]),
    "dispu_match_cost": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_match_cost_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_match_cost_grad_scratch_bytes": (_sz, [_i, _i, _i]),
    "dispu_match_cost_grad": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_match_cost_grad_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "dispu_linear": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp]),
    "dispu_linear_bn": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _vp, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l,
                             _vp]),
    "dispu_linear_tile": (_i, [_i, _i, _i]),
    "dispu_linear_tile2": (_i, [_i, _i, _i, _i, _i]),
    "dispu_debug_linear_tile": (None, [_i]),
    "dispu_sa_fused": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dispu_edge_conv_fused": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dispu_group_center": (_i, [_l, _i, _i, _vp, _vp, _vp]),
    "dispu_pool_nsample": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_idw_weights": (_i, [_l, _vp, _vp, _vp]),
    "dispu_l2_normalize_rows": (_i, [_l, _i, _vp, _vp, _vp]),
    "dispu_scale_add": (_i, [_l, _vp, C.c_float, _vp, _vp, _vp]),
    "dispu_edge_feature": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _l, _vp]),
    "dispu_row_mean_max": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "dispu_repulsion": (_i, [_l, _i, _i, _i, C.c_float, _vp, _vp, _vp, _vp]),
    "dispu_linear_small_k": (_i, [_l, _i, _i, _vp, _l, _vp, _vp, _i, _vp, _l, _vp]),
    "dispu_linear_small_n": (_i, [_l, _i, _i, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _l, _vp]),
    "dispu_edge_dense_conv": (_i, [_i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_stem_block": (_i, [_i, _i, _i, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp,
                              _l, _vp]),
    "dispu_edge_dense_conv_valu": (_i, [_i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_dup_grid": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_ps_prep": (_i, [_l, _i, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_gather_sub_relu": (_i, [_l, _i, _i, _i, _vp, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_skip_max": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_weight_net": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dispu_ps_point_matmul": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp]),
    "dispu_ps_local": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dispu_knn_patch": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_normalize_patches": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_denormalize_patches": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dispu_attention": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, C.c_float, _vp, _l, _vp]),
    "dispu_attention_fwd_lse": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, C.c_float, _vp, _l, _vp, _vp]),
    "dispu_attention_bwd": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, C.c_float, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _l,
                                 _vp, _vp]),
    "dispu_attention_project": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, C.c_float, _vp, _vp, _i, _vp, _l, _vp]),
    "dispu_softmax_rows": (_i, [_l, _i, C.c_float, _vp, _l, _vp]),
    "dispu_linear_tn_bf16_stream_scratch_floats": (_l, [_i, _i, _i]),
    "dispu_linear_tn_bf16_stream": (_i, [_i, _i, _i, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp, _vp, _l, _vp]),
    "dispu_bf16_pack": (_i, [_i, _i, _vp, _l, _i, _vp, _vp]),
    "dispu_linear_bf16_stream": (_i, [_i, _i, _i, _vp, _l, _i, _vp, _l, _vp, _i, _vp, _l, _i, _i, _l, _vp]),
    "dispu_linear_bf16": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp]),
    "dispu_linear_tn_bf16_scratch_floats": (_l, [_i, _i, _i, _i]),
    "dispu_linear_tn_bf16": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _l, _vp]),
    "dispu_bf16x3_split_weights": (_i, [_i, _i, _vp, _l, _vp, _vp]),
    "dispu_linear_bf16x3": (_i, [_i, _i, _i, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_debug_x3_kernel": (None, [_i]),
    "dispu_linear_tn_scratch_floats": (_l, [_i, _i, _i, _i]),
    "dispu_linear_tn": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _l, _vp]),
    "dispu_ps_local_grad": (_i, [_l, _i] + [_vp] * 3 + [_l] + [_vp] * 14),
    "dispu_tn_defer": (_i, [_vp]),
    "dispu_tn_reduce_grouped": (_i, [_i, _vp, _vp, _vp]),
    "dispu_act_bias_grad_scratch_floats": (_l, [_l, _i]),
    "dispu_act_bias_grad": (_i, [_l, _i, _vp, _l, _vp, _l, _i, _vp, _l, _vp, _i, _vp, _l, _vp]),
    "dispu_max_k": (_i, [_l, _i, _i, _vp, _l, _vp, _l, _vp]),
    "dispu_max_k_grad": (_i, [_l, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _vp]),
    "dispu_max_k_grad_tail": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_edge_feature_grad": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _l, _vp]),
    "dispu_dup_sum_grad": (_i, [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_group": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_group_grad": (_i, [_l, _i, _i, _i, _vp, _vp, _l, _vp, _vp, _l, _vp]),
    "dispu_ps_point_matmul_grad": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _vp]),
    "dispu_softmax_rows_grad": (_i, [_l, _i, C.c_float, _vp, _l, _vp, _l, _vp]),
    "dispu_bn_scratch_bytes": (_l, [_l, _i]),
    "dispu_bn_train": (_i, [_l, _i, _vp, _l, _vp, _vp, C.c_float, C.c_float, _i, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_bn_train_grad": (_i, [_l, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_sigmoid_offset": (_i, [_l, _vp, _vp, _vp, _vp]),
    "dispu_sigmoid_offset_grad": (_i, [_l, _vp, _vp, _vp, _vp, _vp]),
    "dispu_repulsion_grad": (_i, [_l, _i, _i, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "dispu_add3": (_i, [_l, _vp, _vp, _vp, _vp, _vp]),
    "dispu_linear_splitk_finish": (_i, [_l, _i, _i, _vp, _l, _vp, _i, _vp, _l, _vp]),
    "dispu_fill_rows": (_i, [_i, _i, _vp, C.c_float, _vp, _vp]),
    "dispu_mlp_chain": (_i, [_l, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _l, _vp, _l, _vp]),
    "dispu_linear_masked": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _vp]),
    "dispu_linear_bf16_masked": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _vp]),
    "dispu_mlp_chain_sum3": (_i, [_l, _i, _i, _i, _i, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _l, _vp, _l, _vp]),
    "dispu_mlp_chain_dup": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _l,
                                 _vp, _l, _vp]),
    "dispu_mlp_chain_stash": (_i, [_l, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _l,
                                   _i, _vp, _l, _vp, _l, _vp]),
    "dispu_mlp_chain_grad": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l,
                                  _vp, _vp, _vp, _l, _vp, _vp, _vp, _l, _vp]),
    "dispu_mask3": (_i, [_l, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _vp, _l, _vp]),
    "dispu_ps_wnet_scratch_bytes": (_l, [_l]),
    "dispu_ps_wnet_bn_stats": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_ps_wnet_grad": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp]),
    "dispu_knn_invert": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dispu_ps_conv0_gather_grad": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_prep_grad": (_i, [_l, _i, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _vp]),
    "dispu_ps_skip_max_grad": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _l, _vp, _l, _vp, _l, _vp, _vp, _l, _i, _vp]),
    "dispu_ps_point_matmul_grad_relu": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _vp]),
    "dispu_edge_dense_conv_grad_scratch_floats": (_l, [_i, _i]),
    "dispu_edge_dense_conv_grad_partials": (_i, [_i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_edge_dense_conv_grad_reduce": (_i, [_i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dispu_edge_dense_conv_grad": (_i, [_i, _i, _i, _vp, _l, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _l, _vp]),
    "dispu_repulsion_loss_grad": (_i, [_l, _i, _i, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "dispu_transpose_batched": (_i, [_i, _vp, _vp, _vp, _vp]),
    "dispu_linear_bf16s": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp, _i, _vp, _l, _l, _vp, _l, _l, _i, _vp]),
    "dispu_linear_tn_bf16s": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _vp, _vp, _l, _i, _vp]),
    "dispu_ps_gather_sub_relu_bf16": (_i, [_l, _i, _i, _i, _vp, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_ps_point_matmul_grad_relu_s": (_i, [_l, _i, _i, _i, _vp, _l, _vp, _vp, _l, _vp, _l, _vp, _i, _vp]),
    "dispu_ps_conv0_gather_grad_s": (_i, [_l, _i, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp]),
    "dispu_chamfer_loss_grad": (_i, [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    "dispu_pu_loss_finalize": (_i, [_vp, _vp, _l, C.c_float, C.c_float, _vp, _vp]),
    "dispu_augment": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dispu_adam": (_i, [_l, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
}

class TnReduceDesc(C.Structure):
    """include/dispu_hip.h: dispu_tn_reduce_desc (72 bytes)."""
    _fields_ = [("part", _vp), ("out", _vp), ("dbias", _vp), ("ldo", _l), ("stride", _l), ("K", _i), ("N", _i), ("splits", _i),
                ("rows_p", _i), ("accumulate", _i), ("bias_accumulate", _i), ("assoc", _i), ("reserved", _i)]


_LIB = None


class DispuError(RuntimeError):
    pass


def lib():
    """Load libdispu_hip.so (after torch, so both share one HIP runtime).  Raises if absent."""
    global _LIB
    if _LIB is None:
        import torch  # noqa: F401  -- must come first: the runtime torch loaded is reused by soname
        if not os.path.exists(LIB_PATH):
            raise DispuError(
                "libdispu_hip.so not found at %s -- build it with `python dis-pu_amd/build.py` "
                "(there is no CPU fallback for the dis-pu_amd ops)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.dispu_version() < ABI_VERSION:
            raise DispuError("libdispu_hip.so at %s has ABI version %d, this package needs >= %d -- rebuild it with "
                             "`python dis-pu_amd/build.py`" % (LIB_PATH, l.dispu_version(), ABI_VERSION))
        _LIB = l
    return _LIB


# ---- launch tape -----------------------------------------------------------------------------------------------------------------
# The training step is ~130 launches of 5 - 150 us; issued from Python each costs ~10 us of interpreter + ctypes argument conversion
# (1.4 ms per step at 8 patches, as long as the GPU's own critical chain).  A Tape records the launch sequence of one step -- every C
# call with its arguments already converted to ctypes objects, plus the stream / event operations between them -- and replays it as a
# flat loop of foreign calls (~1.5 us each).  Unlike a hipGraph replay it keeps the eager step's stream assignment and submission order.
class Tape(object):
    def __init__(self):
        self.calls = []          # (foreign function, ctypes args, label)
        self.keep = []           # objects the recorded raw handles belong to (events, streams, tensors)

    def replay(self):
        for fn, args, what in self.calls:
            rc = fn(*args)
            if rc:
                check(rc, what + " (tape replay)")

    def __len__(self):
        return len(self.calls)


_TAPE = None                     # the Tape being recorded, if any
_NO_TAPE = ("dispu_version", "dispu_error_string", "dispu_linear_tile", "dispu_linear_tile2")


class _TapeLib(object):
    """the library seen through a recorder: calls execute as usual; while a Tape is being recorded, calls that launch work
    (int-returning, not a size query) are appended to it with their converted arguments."""

    def __init__(self, real):
        self._real = real
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real = getattr(self._real, name)
            if name in _NO_TAPE or real.restype is not _i or "scratch" in name:
                fn = real
            else:
                argtypes = real.argtypes

                def fn(*args, _real=real, _types=argtypes, _name=name):
                    if _TAPE is None:
                        return _real(*args)
                    cargs = tuple(a if isinstance(a, C._SimpleCData) else t(a) for a, t in zip(args, _types))
                    rc = _real(*cargs)
                    _TAPE.calls.append((_real, cargs, _name))
                    return rc
            self._cache[name] = fn
        return fn


_TAPELIB = None


def tape_lib():
    global _TAPELIB
    if _TAPELIB is None:
        _TAPELIB = _TapeLib(lib())
    return _TAPELIB


def tape_begin():
    global _TAPE
    _TAPE = Tape()
    return _TAPE


def tape_end():
    global _TAPE
    t, _TAPE = _TAPE, None
    return t


def taping():
    return _TAPE


def check(code, what):
    if code != 0:
        msg = lib().dispu_error_string(code)
        raise DispuError("%s failed: hip error %d (%s)" % (what, code, msg.decode() if msg else "?"))


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
