"""ctypes binding of libtgpose_hip.so (the C ABI declared in include/tgpose.h).

There is NO fallback: if the library is missing or a call fails, this raises.  The product path
never routes through the CPU oracle or through torch ops for the hot layers.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtgpose_hip.so")

c_int, c_i64, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class GemmArgs(ctypes.Structure):
    """struct tgp_gemm_args (include/tgpose.h)"""
    _fields_ = [
        ("A", c_vp), ("lda", c_int),
        ("W", c_vp), ("ldw", c_int),
        ("C", c_vp), ("ldc", c_int),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("bias", c_vp),
        ("rowbias", c_vp), ("ldrb", c_int),
        ("rows_per_obj", c_int),
        ("res1", c_vp), ("ldr1", c_int),
        ("res2", c_vp), ("ldr2", c_int),
        ("scale", c_vp),
        ("shift", c_vp),
        ("act", c_int),
        ("slope", c_f32),
        ("colmax_keys", c_vp), ("ldcm", c_int),
        ("slope_vec", c_vp),
        ("cm_cols", c_int), ("c_col0", c_int),
        ("batch", c_int),
        ("batch_stride_a", c_i64), ("batch_stride_w", c_i64), ("batch_stride_c", c_i64),
        ("batch_stride_vec", c_i64), ("batch_stride_colmax", c_i64),
        ("W_split", c_vp), ("ldws", c_int), ("w_split_kind", c_int),
        ("a_scale", c_vp), ("c_scale", c_vp), ("ksplit_chunk", c_int),
        ("gres1", c_vp), ("ldg1", c_int), ("gidx1", c_vp),
        ("gres2", c_vp), ("ldg2", c_int), ("gidx2", c_vp),
        ("epilogue", c_int), ("pred", c_vp),
        ("row_base", c_int),
        ("A_planes", c_vp), ("a_kt", c_int), ("a_amax", c_vp),
        ("W_planes", c_vp), ("w_kt", c_int),
        ("C_planes", c_vp), ("c_kt", c_int), ("cp_col0", c_int), ("c_amax", c_vp),
        ("pp_config", c_int),
        ("range_flag", c_vp),
        ("a_keys", c_int), ("a_wrap", c_int),
        ("C_sigmoid", c_vp),
    ]


# name -> (restype, argtypes); every symbol include/tgpose.h declares
class ConvMaxFusedArgs(ctypes.Structure):
    """struct tgp_conv_max_fused_args (include/tgpose.h)"""
    _fields_ = [
        ("fine", c_vp), ("ldf", c_int), ("K", c_int),
        ("wa_planes", c_vp),
        ("p1", c_vp), ("ldp1", c_int), ("p1_rows", c_int), ("idx1", c_vp),
        ("p2", c_vp), ("ldp2", c_int), ("p2_rows", c_int), ("idx2", c_vp),
        ("bias", c_vp), ("scale", c_vp), ("shift", c_vp), ("slope", ctypes.c_float),
        ("keys", c_vp), ("ldk", c_int),
        ("M", c_int), ("rows_per_obj", c_int), ("C", c_int),
        ("overflow", c_vp),
        ("fine_planes", c_vp), ("fine_kt", c_int), ("fine_amax", c_vp),
    ]


class HeadsFusedArgs(ctypes.Structure):
    """struct tgp_heads_fused_args (include/tgpose.h)"""
    _fields_ = [
        ("fine", c_vp), ("ldf", c_int), ("K", c_int),
        ("wa_planes", c_vp),
        ("p1", c_vp), ("ldp1", c_int), ("idx1", c_vp),
        ("p2", c_vp), ("ldp2", c_int), ("idx2", c_vp),
        ("w2p", c_vp),
        ("bias2", c_vp), ("scale2", c_vp), ("shift2", c_vp),
        ("keys", c_vp),
        ("M", c_int), ("rows_per_obj", c_int), ("B", c_int), ("heads", c_int),
        ("overflow", c_vp),
        ("rows", c_int),
        ("fine_planes", c_vp), ("fine_kt", c_int), ("fine_amax", c_vp),
    ]


class DecFusedArgs(ctypes.Structure):
    """struct tgp_dec_fused_args (include/tgpose.h)"""
    _fields_ = [
        ("h1_planes", c_vp), ("h1_kt", c_int), ("h1_amax", c_vp),
        ("units", c_vp),
        ("vec", (c_vp * 3) * 3),
        ("w5", c_vp), ("b5", c_vp),
        ("map", c_vp), ("rows_per_obj", c_int),
        ("out", c_vp),
        ("flag", c_vp),
        ("M", c_int),
    ]


class HsChainArgs(ctypes.Structure):
    """struct tgp_hs_chain_args (include/tgpose.h)"""
    _fields_ = [
        ("a_planes", c_vp), ("a_kt", c_int), ("a_amax", c_vp),
        ("M", c_int), ("K1", c_int), ("N1", c_int), ("N2", c_int),
        ("units", c_vp),
        ("rowbias", c_vp), ("ldrb", c_int), ("rows_per_obj", c_int),
        ("res1", c_vp), ("ldr1", c_int),
        ("res2", c_vp), ("ldr2", c_int),
        ("scale1", c_vp), ("shift1", c_vp),
        ("relu", c_int),
        ("c1", c_vp), ("ldc1", c_int),
        ("c1_planes", c_vp), ("c1_kt", c_int), ("c1_kt0", c_int), ("c1_amax", c_vp),
        ("bias2", c_vp),
        ("c2", c_vp), ("ldc2", c_int),
        ("flag", c_vp),
    ]


class ProjPlanesArgs(ctypes.Structure):
    """struct tgp_proj_planes_args (include/tgpose.h)"""
    _fields_ = [
        ("a_planes", c_vp), ("a_kt", c_int), ("a_amax", c_vp),
        ("a", c_vp), ("lda", c_int),
        ("M", c_int), ("K", c_int), ("N", c_int),
        ("units", c_vp),
        ("w", c_vp), ("ldw", c_int),
        ("bias", c_vp),
        ("c", c_vp), ("ldc", c_int),
    ]


class DecL1Args(ctypes.Structure):
    """struct tgp_dec_l1_args (include/tgpose.h)"""
    _fields_ = [
        ("fine_planes", c_vp), ("fine_kt", c_int), ("fine_amax", c_vp),
        ("wa_planes", c_vp),
        ("p1", c_vp), ("ldp1", c_int), ("idx1", c_vp),
        ("p2", c_vp), ("ldp2", c_int), ("idx2", c_vp),
        ("bias", c_vp), ("scale", c_vp), ("shift", c_vp),
        ("rowbias", c_vp), ("ldrb", c_int), ("rows_per_obj", c_int),
        ("h1_planes", c_vp), ("h1_kt", c_int), ("h1_amax", c_vp),
        ("flag", c_vp),
        ("M", c_int),
    ]


SIGNATURES = {
    "tgp_version": (c_int, []),
    "tgp_graph_node_counts": (c_int, [c_vp, c_vp]),
    "tgp_knn_max_points": (c_int, []),
    "tgp_knn_max_k": (c_int, []),
    "tgp_center": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_center_zero": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "tgp_knn_xyz": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "tgp_knn_feat_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "tgp_knn_feat": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_vp]),
    "tgp_knn_feat_form": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_int, c_vp]),
    "tgp_knn_feat_dirs": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "tgp_nn1": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "tgp_nn1_pair": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_nn1_pair_tail": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_vp]),
    "tgp_normalize_dirs": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "tgp_normalize_dirs_bwd": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp]),
    "tgp_gconv_surface_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "tgp_gconv_hs_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_gconv_hs_fwd_dirs": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_orl_partial_floats": (c_i64, [c_int, c_int, c_int]),
    "tgp_orl_global": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_orl_rowbias": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_orl_rowbias_planes": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "tgp_orl_rowbias_fused": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp,
                                      c_vp]),
    "tgp_pool_fwd": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "tgp_pool_fwd_planes": (c_int, [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_int,
                                    c_vp, c_vp]),
    "tgp_gather_rows": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_fill_tail": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp]),
    "tgp_split_bf16": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_split_f16": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_planes_bytes": (c_i64, [c_i64, c_int]),
    "tgp_planes_split": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_planes_split_cols": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_planes_gather": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_gemm_tn_workspace_floats": (c_i64, [c_i64, c_int, c_int]),
    "tgp_gemm_tn_f32": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_bw_workspace_floats": (c_i64, [c_i64, c_int]),
    "tgp_colsum": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_bn_bwd": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_int, c_f32, c_vp,
                           c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_absmax_scale_from_bits": (c_int, [c_vp, c_int, c_f32, c_vp, c_vp]),
    "tgp_bn_bwd_absmax_words": (c_i64, [c_i64, c_int]),
    "tgp_bn_bwd_pooled": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_f32, c_vp, c_vp,
                                  c_int, c_f32, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "tgp_colmax_arg": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_int, c_f32, c_vp, c_vp,
                               c_int, c_vp, c_int, c_vp]),
    "tgp_colsum_objects": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_colmax_bwd": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_transpose": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_gconv_bwd_workspace_floats": (c_i64, [c_int, c_int, c_int]),
    "tgp_gconv_surface_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_gconv_hs_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp,
                                 c_vp, c_vp]),
    "tgp_nbrmax_bwd": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_f32, c_vp, c_int, c_vp]),
    "tgp_iou3d_pairs": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "tgp_rt_error_pairs": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "tgp_absmax_scale": (c_int, [c_vp, c_int, c_i64, c_int, c_f32, c_vp, c_vp, c_vp]),
    "tgp_transpose_scaled": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "tgp_transpose_split_f16": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "tgp_sum_slabs": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_vp]),
    "tgp_pose_terms_fwd": (c_int, [c_vp] * 11 + [c_int, c_int, c_int, c_f32, c_vp, c_vp]),
    "tgp_pose_terms_bwd": (c_int, [c_vp] * 11 + [c_int, c_int, c_int, c_f32] + [c_vp] * 8 + [c_vp]),
    "tgp_sym_recon_workspace_floats": (c_i64, [c_int, c_int]),
    "tgp_sym_recon_fwd": (c_int, [c_vp] * 5 + [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_sym_recon_bwd": (c_int, [c_vp] * 5 + [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tgp_rowl1_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_rowl1_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_feat_consistency_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_feat_consistency_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_pose_transform_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_pose_transform_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_gather_rows_bwd": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "tgp_gemm_f32": (c_int, [ctypes.POINTER(GemmArgs), c_vp]),
    "tgp_colmax_decode": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_colmax": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "tgp_sigmoid": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "tgp_head_post": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_add_mean": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "tgp_bn_workspace_floats": (c_i64, [c_i64, c_int]),
    "tgp_bn_stats": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tgp_bn_stats_running": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp]),
    "tgp_bn_apply": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_int, c_f32, c_vp, c_vp, c_int, c_vp,
                             c_int, c_int, c_int, c_vp]),
    "tgp_dropout_apply": (c_int, [c_vp, c_vp, c_f32, c_i64, c_vp, c_vp]),
    "tgp_chamfer_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_chamfer_bwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_dcd_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_f32, c_int, c_vp, c_vp, c_vp, c_vp]),
    "tgp_dcd_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_f32, c_vp, c_vp, c_vp]),
    "tgp_generate_rt": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_canonicalize": (c_int, [c_vp] * 9 + [c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_heads_fused": (c_int, [ctypes.POINTER(HeadsFusedArgs), c_vp]),
    "tgp_conv_max_fused": (c_int, [ctypes.POINTER(ConvMaxFusedArgs), c_vp]),
    "tgp_heads_pack_w2": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "tgp_heads_w2_bytes": (c_i64, [c_int]),
    "tgp_dec_fused": (c_int, [ctypes.POINTER(DecFusedArgs), c_vp]),
    "tgp_hs_chain": (c_int, [ctypes.POINTER(HsChainArgs), c_vp]),
    "tgp_proj_planes": (c_int, [ctypes.POINTER(ProjPlanesArgs), c_vp]),
    "tgp_proj_pack_bytes": (c_i64, [c_int, c_int]),
    "tgp_proj_pack": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "tgp_hs_chain_pack_bytes": (c_i64, [c_int, c_int, c_int]),
    "tgp_hs_chain_pack": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_dec_pack_bytes": (c_i64, []),
    "tgp_dec_pack": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "tgp_dec_l1": (c_int, [ctypes.POINTER(DecL1Args), c_vp]),
    "tgp_sort_by_parent": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "tgp_roi_cloud": (c_int, [c_vp] * 7 + [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_cloud_select": (c_int, [c_vp] * 5 + [c_int, c_int, c_int, c_vp, c_vp]),
    "tgp_pose_rotation_fwd": (c_int, [c_vp] * 6 + [c_int, c_vp, c_vp, c_vp]),
    "tgp_pose_rotation_bwd": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp]),
    "tgp_rows_out": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_rows_out_pred": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "tgp_pose_tail": (c_int, [c_vp] * 8 + [c_int] + [c_vp] * 8),
    "tgp_head_post_bwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int] + [c_vp] * 10),
    "tgp_transpose_both": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "tgp_gemm_tn_split": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "tgp_reverse_graph": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_nbrmax_bwd_gather": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_f32,
                                      c_vp, c_vp, c_int, c_vp]),
    "tgp_gconv_hs_bwd_gather": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "tgp_gconv_hs_fwd_slots": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "tgp_child_lists": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_segsum_rows": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
    "tgp_roi_cloud_ex": (c_int, [c_vp] * 7 + [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp]),
    "tgp_cloud_select_ex": (c_int, [c_vp] * 5 + [c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "tgp_cloud_sample": (c_int, [c_vp] * 5 + [c_int, c_int, c_int, ctypes.c_uint64, c_vp, c_vp]),
}

ABI_VERSION = 7
_lib = None


class TgpError(RuntimeError):
    pass


def lib():
    """Load the shared library once; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgpError(
                "tgpose_amd: %s not found. Build it with `python -m tgpose_amd.build` (hipcc, gfx950). "
                "There is no CPU/torch fallback for the HIP path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.tgp_version() != ABI_VERSION:
            raise TgpError("tgpose_amd: ABI version mismatch (%d != %d)" % (handle.tgp_version(), ABI_VERSION))
        if os.environ.get("TGP_TRACE", "0") != "0":
            # diagnosis: every entry point announces itself on stderr before it launches (with HIP_LAUNCH_BLOCKING=1 the last
            # name printed is the launch that faulted)
            class _Traced(object):
                def __init__(self, h):
                    self._h = h

                def __getattr__(self, name):
                    fn = getattr(self._h, name)

                    def call(*a):
                        sys.stderr.write("tgp: %s\n" % name)
                        sys.stderr.flush()
                        return fn(*a)
                    return call
            handle = _Traced(handle)
        _lib = handle
    return _lib


_ERR = {-1: "TGP_EINVAL (bad pointer / size / stride / alignment)", -2: "TGP_EUNSUPPORTED (shape not supported)"}


def check(rc, what):
    if rc != 0:
        raise TgpError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))
