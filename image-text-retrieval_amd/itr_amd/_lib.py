"""ctypes binding of libitr_hip.so (C ABI declared in include/itr_hip.h).

There is NO CPU fallback: importing the ops without the built library, or calling them with
CPU tensors, raises.  Error codes map onto the reference's exception types:
ITR_ERR_BADARG -> ValueError, ITR_ERR_UNSUPPORTED -> NotImplementedError, ITR_ERR_HIP ->
RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libitr_hip.so")

ABI_VERSION = 30

i32, i64, f32, vp, sz, u64 = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_size_t, C.c_uint64

# name -> (restype, argtypes).  Kept in one table so tests can check that every symbol the
# header declares is exported with the arity the binding expects.
SIGNATURES = {
    "itr_last_error": (C.c_char_p, []),
    "itr_abi_version": (i32, []),
    "itr_l2norm_rows": (i32, [vp, vp, i64, i32, f32, i32, i32, vp]),
    "itr_mean_mid": (i32, [vp, vp, i64, i32, i32, vp]),
    "itr_split_bf16": (i32, [vp, vp, i64, i64, vp]),
    "itr_split_f16": (i32, [vp, vp, vp, i64, i64, vp]),
    "itr_gemm_nt_f16x3": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, i64, i64, i64, i64, i32, vp]),
    "itr_gemm_nt_bf16": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i32, i32, vp]),
    "itr_dropout": (i32, [vp, vp, i64, f32, u64, u64, vp]),
    "itr_add_ln_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp]),
    "itr_ln_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_gelu": (i32, [vp, vp, vp, i64, vp]),
    "itr_mha_train_fwd": (i32, [vp, vp, vp, i64, vp, i64, i32, i32, i32, f32, f32, u64, vp, vp, i64, vp]),
    "itr_mha_train_bwd": (i32, [vp, vp, vp, i64, vp, i64, i32, i32, i32, f32, f32, u64, vp, vp, i64, vp, vp, vp, i64, vp]),
    "itr_relu_maxpool_arg": (i32, [vp, i64, i32, i32, vp, vp, vp]),
    "itr_relu_maxpool_bwd": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "itr_bcast_mid": (i32, [vp, vp, i64, i32, i32, f32, vp]),
    "itr_angular_fwd": (i32, [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp]),
    "itr_angular_bwd": (i32, [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "itr_diversity_fwd": (i32, [vp, i64, i32, i32, vp, vp, vp]),
    "itr_diversity_bwd": (i32, [vp, i64, i32, i32, vp, vp, vp]),
    "itr_ew_mul": (i32, [vp, vp, vp, i64, vp]),
    "itr_act_bwd": (i32, [vp, vp, vp, i64, i32, vp]),
    "itr_gate_apply": (i32, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_gate_apply_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_bn_train_scratch_bytes": (sz, [i64, i32]),
    "itr_bn_train_fwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp, vp]),
    "itr_bn_train_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]),
    "itr_l2norm_mid_fwd": (i32, [vp, vp, vp, i64, i32, i32, f32, vp]),
    "itr_l2norm_mid_bwd": (i32, [vp, vp, vp, vp, i64, i32, i32, f32, vp]),
    "itr_smry_fwd": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "itr_smry_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "itr_bmm_small": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]),
    "itr_groupmax_fwd": (i32, [vp, i64, i32, i64, vp, vp, vp]),
    "itr_groupmax_bwd": (i32, [vp, vp, i64, i32, i64, vp, vp]),
    "itr_gru_cell_fwd": (i32, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_gru_cell_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_add_bcast_mid_act": (i32, [vp, vp, vp, i64, i32, i32, i32, vp]),
    "itr_add_bcast_mid_act_bwd": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "itr_addattn_score": (i32, [vp, vp, vp, vp, i64, i32, i32, vp]),
    "itr_addattn_score_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "itr_nll_logsoftmax_fwd": (i32, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_nll_logsoftmax_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "itr_gcn_relation": (i32, [vp, i64, vp, i64, i64, i32, i32, vp]),
    "itr_order_scores": (i32, [vp, vp, vp, i64, i64, i32, vp]),
    "itr_order_bwd": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i32, vp]),
    "itr_row_sqnorm": (i32, [vp, vp, i64, i32, vp]),
    "itr_pdist_finish": (i32, [vp, vp, vp, i64, i64, vp]),
    "itr_gemm_nt": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "itr_gemm_nt_algo": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i32, i32, vp]),
    "itr_proj_l2norm": (i32, [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "itr_gru_workspace_bytes": (sz, [i64, i64, i32, i32, i32]),
    "itr_gru_fwd": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp,
                          i32, i32, i32, vp, vp, vp, sz, vp]),
    "itr_cosine_scores": (i32, [vp, vp, vp, i64, i64, i32, i64, vp]),
    "itr_mvm_scores": (i32, [vp, vp, vp, i64, i64, i32, i32, i64, vp]),
    "itr_hinge_maxviol_fwd": (i32, [vp, i32, i64, f32, i32, vp, vp, vp, vp, vp]),
    "itr_hinge_maxviol_bwd": (i32, [vp, i32, i64, f32, i32, vp, vp, vp, vp, i64, vp]),
    "itr_scan_plan_tiles": (i32, [vp, i64, i32, vp, vp, vp]),
    "itr_sgr_plan_node_groups": (i32, [vp, i64, i32, vp, vp, vp]),
    "itr_scan_workspace_bytes": (sz, [i64, i32, i64, i64, i64, i32]),
    "itr_scan_prepare": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, vp, sz, vp]),
    "itr_scan_xattn_scores": (i32, [vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, i64, vp, sz, vp]),
    "itr_scan_bf16_workspace_bytes": (sz, [i64, i32, i64, i32]),
    "itr_scan_xattn_scores_bf16x3": (i32, [vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, i64, vp, sz, vp, sz, i32, vp]),
    "itr_bert_embed_ln": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, i32, i32, f32, vp]),
    "itr_add_layernorm": (i32, [vp, vp, vp, vp, vp, i64, i32, f32, vp]),
    "itr_mha_small": (i32, [vp, vp, vp, i64, i64, i64, vp, vp, i64, i64, i32, i32, i32, f32, vp]),
    "itr_relu_maxpool": (i32, [vp, vp, i64, i64, i32, i32, i32, vp]),
    "itr_gemm_nt_acc": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i32, vp]),
    "itr_mul_rows": (i32, [vp, vp, i64, vp, i64, i32, vp]),
    "itr_gemm_nt_residual": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, i32, vp]),
    "itr_agsa_gate": (i32, [vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "itr_affine_cols": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "itr_camera_posenc": (i32, [vp, vp, vp, i64, i32, vp]),
    "itr_camera_summarize": (i32, [vp, vp, vp, i64, i32, i32, i32, vp]),
    "itr_sgraf_workspace_bytes": (sz, [i64, i64, i64, i64, i32, i32, i32, i32, i32]),
    "itr_sgraf_pick_image_block": (i32, [i64, i64, i64, i64, i32, i32, i32, i32, sz, vp]),
    "itr_sgraf_scores": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, i32, i32, vp, i64, vp, sz, vp]),
    "itr_debug_scan_occupancy": (i32, [vp, vp]),
    "itr_debug_scan_clock_probe": (i32, [vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, i64, vp, sz, vp]),
    "itr_rank_gather_gt": (i32, [vp, i64, i64, i64, i64, i32, vp, vp]),
    "itr_rank_workspace_bytes": (sz, [i64, i64]),
    "itr_rank_counts": (i32, [vp, i64, i64, i64, i64, i32, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
    "itr_rank_gather_gt_f64": (i32, [vp, i64, i64, i64, i64, i32, vp, vp]),
    "itr_rank_counts_f64": (i32, [vp, i64, i64, i64, i64, i32, vp, vp, vp, vp, vp, vp]),
    "itr_rank_t2i_top1_f64": (i32, [vp, i64, i64, i64, i64, vp, vp, vp]),
    "itr_recall_from_ranks": (i32, [vp, i64, vp]),
    # ---- training step
    "itr_l2norm_fwd_save": (i32, [vp, vp, vp, i64, i32, f32, vp]),
    "itr_l2norm_bwd": (i32, [vp, vp, vp, vp, i64, i32, f32, vp]),
    "itr_transpose2d": (i32, [vp, vp, i64, i64, vp]),
    "itr_colsum_workspace_bytes": (sz, [i64, i64]),
    "itr_colsum": (i32, [vp, vp, i64, i64, i32, vp, sz, vp]),
    "itr_gemm_nt_splitk_workspace_bytes": (sz, [i64, i64, i64]),
    "itr_gemm_nt_splitk": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, i32, vp, sz, vp]),
    "itr_gemm_tn_workspace_bytes": (sz, [i64, i64, i64]),
    "itr_gemm_tn": (i32, [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp, vp, sz, vp]),
    "itr_gemm_tn_batched": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i64, i64, i64, vp]),
    "itr_sgt_attn_fwd": (i32, [vp, i64, vp, i32, i32, i32, i32, i32, f32, f32, vp, vp]),
    "itr_sgt_attn_bwd": (i32, [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32, vp, vp]),
    "itr_sgt_ctx_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "itr_sgt_ctx_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "itr_sgt_ctx_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, sz, vp]),
    "itr_sgt_dp": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
    "itr_sgt_pair_sqdiff_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "itr_sgt_pair_sqdiff_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "itr_sgt_nodes": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "itr_sgt_graph_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "itr_sgt_graph_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "itr_sgt_segbn_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp]),
    "itr_sgt_segbn_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "itr_sgt_saf_pool_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp]),
    "itr_sgt_saf_pool_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp]),
    "itr_sgt_seg_mean": (i32, [vp, vp, i32, i32, vp, i32, i32, vp]),
    "itr_sgt_seg_smry_fwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "itr_sgt_seg_smry_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "itr_embed_scatter_add": (i32, [vp, vp, i64, i64, i32, vp, vp]),
    "itr_gather_rows": (i32, [vp, i64, vp, i64, i32, vp, vp, vp]),
    "itr_sq_sum_blocks": (i32, [i64]),
    "itr_sq_sum": (i32, [vp, i64, vp, vp]),
    "itr_clip_coef": (i32, [vp, i64, f32, vp, vp]),
    "itr_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, vp, vp]),
    "itr_sq_sum_multi": (i32, [vp, vp, i64, vp, vp]),
    "itr_adam_step_multi": (i32, [vp, vp, vp, i64, f32, f32, f32, f32, i64, vp, vp]),
    "itr_gru_train_save_bytes": (sz, [i64, i32, i32]),
    "itr_gru_train_workspace_bytes": (sz, [i64, i64, i32, i32]),
    "itr_gru_fwd_train": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, sz, vp]),
    "itr_gru_bwd": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                          vp, sz, vp]),
    "itr_scan_train_prepare": (i32, [vp, vp, i64, i64, i32, i32, vp, vp, vp]),
    "itr_scan_train_fwd": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, vp]),
    "itr_scan_train_bwd": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp]),
    "itr_scan_train_finish": (i32, [vp, i64, i64, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]),
    "itr_scan_train_i2t_prepare": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, vp, vp]),
    "itr_scan_train_i2t_fwd": (i32, [vp, i64, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, vp]),
    "itr_scan_train_i2t_bwd": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp]),
    "itr_scan_train_i2t_finish": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]),
}



class SgrafWeights(C.Structure):
    """itr_sgraf_weights (include/itr_hip.h)."""
    _single = ["v_loc_w", "v_loc_b", "v_loc_bn_w", "v_loc_bn_b", "v_loc_bn_mean", "v_loc_bn_var",
               "v_glo_w", "v_glo_b", "v_glo_bn_w", "v_glo_bn_b", "v_glo_bn_mean", "v_glo_bn_var", "v_com_w", "v_com_b",
               "t_loc_w", "t_loc_b", "t_glo_w", "t_glo_b", "t_com_w", "t_com_b",
               "loc_w", "loc_b", "glo_w", "glo_b", "eval_w", "eval_b",
               "saf_w", "saf_b", "saf_bn_w", "saf_bn_b", "saf_bn_mean", "saf_bn_var"]
    _arrays = ["sgr_q_w", "sgr_q_b", "sgr_k_w", "sgr_k_b", "sgr_g_w", "sgr_g_b"]
    _fields_ = [(n, C.c_void_p) for n in _single] + [(n, C.c_void_p * 8) for n in _arrays]


_lib = None


class ItrLibraryMissing(ImportError):
    pass


def load():
    """Load (once) and return the ctypes library; raises ItrLibraryMissing loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ItrLibraryMissing(
            "libitr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C image-text-retrieval_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.itr_abi_version() != ABI_VERSION:
        raise ItrLibraryMissing("libitr_hip.so ABI %d != binding ABI %d: rebuild" %
                                (lib.itr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code):
    if code == 0:
        return
    msg = load().itr_last_error().decode("utf-8", "replace")
    if code == -1:
        raise ValueError(msg)
    if code == -2:
        raise NotImplementedError(msg)
    raise RuntimeError("libitr_hip (code %d): %s" % (code, msg))
