"""ctypes binding of libattnshift_hip.so (the C ABI declared in include/attnshift.h).

There is NO fallback: if the library is missing or a call fails the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libattnshift_hip.so")

AS_F32, AS_BF16 = 0, 1

_c_void_p, _c_int, _c_float, _c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_c_double = ctypes.c_double

# name -> (restype, argtypes); every symbol include/attnshift.h declares
SIGNATURES = {
    "as_version": (_c_int, []),
    "as_last_error": (ctypes.c_char_p, []),
    "as_npad": (_c_int, [_c_int]),
    "as_linear_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 5 + [_c_void_p]),
    "as_linear_small_fwd": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p] + [_c_int] * 5 + [_c_void_p]),
    "as_linear_sk_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "as_linear_sk_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 5 + [_c_void_p, _c_size_t, _c_void_p]),
    "as_linear_gelu_fwd": (_c_int, [_c_void_p] * 5 + [_c_int] * 4 + [_c_void_p]),
    "as_linear_dgelu_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p]),
    "as_deconv2x2_fwd": (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p]),
    "as_qkv_fwd": (_c_int, [_c_void_p] * 6 + [_c_int] * 5 + [_c_void_p]),
    "as_sdpa_fwd_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "as_sdpa_fwd": (_c_int, [_c_void_p] * 6 + [_c_size_t] + [_c_int] * 4 + [_c_void_p]),
    "as_sdpa_bwd_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "as_sdpa_bwd": (_c_int, [_c_void_p] * 8 + [_c_size_t] + [_c_int] * 4 + [_c_void_p]),
    "as_attn_bwd_workspace_bytes": (_c_size_t, [_c_int] * 5),
    "as_attn_bwd": (_c_int, [_c_void_p] * 15 + [_c_size_t] + [_c_int] * 5 + [_c_void_p]),
    "as_window_attn_fwd": (_c_int, [_c_void_p] * 5 + [_c_int] * 8 + [_c_void_p]),
    "as_add_layernorm_bwd_workspace_bytes": (_c_size_t, [_c_int] * 2),
    "as_add_layernorm_bwd": (_c_int, [_c_void_p] * 4 + [ctypes.c_float] + [_c_void_p] * 5 + [_c_size_t] + [_c_int] * 3 + [_c_void_p]),
    "as_add_layernorm": (_c_int, [_c_void_p] * 4 + [ctypes.c_float] + [_c_void_p] * 2 + [_c_int] * 3 + [_c_void_p]),
    "as_add_layernorm_scaled": (_c_int, [_c_void_p] * 4 + [ctypes.c_float] + [_c_void_p] * 2 + [_c_int] * 3 + [_c_void_p, _c_int, _c_void_p]),
    "as_add_layernorm_bwd_scaled": (_c_int, [_c_void_p] * 4 + [ctypes.c_float] + [_c_void_p] * 5 + [_c_size_t] + [_c_int] * 3
                                    + [_c_void_p, _c_int, _c_void_p]),
    "as_linear_splitk_workspace_bytes": (ctypes.c_size_t, [_c_int] * 3),
    "as_linear_splitk_fwd": (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p, ctypes.c_size_t, _c_void_p]),
    "as_linear_bwd_workspace_bytes": (ctypes.c_size_t, [_c_int] * 3),
    "as_linear_bwd": (_c_int, [_c_void_p] * 6 + [_c_int] * 5 + [_c_void_p, ctypes.c_size_t, _c_void_p]),
    "as_linear_bwd_dgelu": (_c_int, [_c_void_p] * 7 + [_c_int] * 5 + [_c_void_p, ctypes.c_size_t, _c_void_p]),
    "as_assemble_tokens": (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p]),
    "as_maxpool_nhwc": (_c_int, [_c_void_p] * 2 + [_c_int] * 5 + [ctypes.c_longlong, _c_void_p]),
    "as_mask_count": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p]),
    "as_window_attn_bwd_workspace_bytes": (_c_size_t, [_c_int] * 5),
    "as_window_attn_bwd": (_c_int, [_c_void_p] * 8 + [_c_size_t] + [_c_int] * 8 + [_c_void_p]),
    "as_attn_fwd": (_c_int, [_c_void_p] * 12 + [_c_size_t] + [_c_int] * 5 + [_c_void_p]),
    "as_attn_mean_rows": (_c_int, [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p]),
    "as_rollout_rfrag_bytes": (_c_size_t, [_c_int] * 3),
    "as_rollout_pack": (_c_int, [_c_void_p] * 2 + [_c_int] * 4 + [_c_void_p]),
    "as_rollout_top": (_c_int, [_c_void_p] * 5 + [_c_int] * 5 + [_c_void_p]),
    "as_rollout_step_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "as_rollout_step": (_c_int, [_c_void_p] * 8 + [_c_size_t] + [_c_int] * 5 + [_c_void_p]),
    "as_ccl_2d": (_c_int, [_c_void_p] * 2 + [_c_int] * 3 + [_c_void_p]),
    "as_cam_boxes_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "as_cam_boxes": (_c_int, [_c_void_p] * 2 + [_c_float] * 2 + [_c_int] * 4 + [_c_void_p] * 5 + [_c_size_t, _c_void_p]),
    "as_cam_sample_masks_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "as_cam_sample_masks": (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [_c_float] * 2 + [_c_void_p] * 3 + [_c_size_t, _c_void_p]),
    "as_mask_candidates_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "as_mask_candidates": (_c_int, [_c_void_p] * 3 + [_c_float] * 3 + [_c_int] + [_c_void_p] * 5 + [_c_size_t]
                           + [_c_int] * 3 + [_c_void_p]),
    "as_part_select": (_c_int, [_c_void_p] * 7 + [_c_int] * 5 + [_c_void_p] * 8 + [_c_void_p]),
    "as_part_stats": (_c_int, [_c_void_p] * 3 + [_c_float] + [_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p]),
    "as_filter_parts": (_c_int, [_c_void_p] * 2 + [_c_float] * 2 + [_c_void_p] + [_c_int] * 3 + [_c_void_p]),
    "as_draw_distinct": (_c_int, [_c_void_p, _c_int, _c_int] + [_c_void_p] * 5 + [_c_int] * 3 + [_c_void_p]),
    "as_mt_sample_ranks": (_c_int, [_c_void_p] * 4 + [_c_int] * 2 + [_c_void_p]),
    "as_mt_perm_ranks": (_c_int, [_c_void_p] * 4 + [_c_int] * 2 + [_c_void_p]),
    "as_small_attn_fwd": (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p]),
    "as_small_attn_bwd_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "as_small_attn_bwd": (_c_int, [_c_void_p] * 6 + [_c_size_t] + [_c_int] * 5 + [_c_void_p]),
    "as_roi_align_fwd": (_c_int, [_c_void_p] * 3 + [_c_int] * 6 + [_c_float] + [_c_int] * 2 + [_c_void_p]),
    "as_roi_align_bwd": (_c_int, [_c_void_p] * 3 + [_c_int] * 6 + [_c_float] + [_c_int] * 2 + [_c_void_p]),
    "as_chamfer_2d_fwd": (_c_int, [_c_void_p] * 6 + [_c_int] * 3 + [_c_void_p]),
    "as_chamfer_2d_bwd": (_c_int, [_c_void_p] * 8 + [_c_int] * 3 + [_c_void_p]),
    "as_merge_plan": (_c_int, [_c_void_p] * 4 + [_c_int] * 2 + [_c_void_p]),
    "as_semantic_prestage": (_c_int, [_c_void_p, _c_float] + [_c_int] * 5 + [_c_void_p] * 4),
    "as_cosine_shift_workspace_bytes": (_c_size_t, [_c_int] * 6),
    "as_cosine_shift": (_c_int, [_c_void_p] * 5 + [_c_double] * 2 + [_c_int] + [_c_void_p] * 4 + [_c_size_t]
                        + [_c_int] * 6 + [_c_void_p]),
    "as_cosine_shift_strided": (_c_int, [_c_void_p, ctypes.c_longlong] + [_c_void_p] * 4 + [_c_double] * 2 + [_c_int] + [_c_void_p] * 4
                                + [_c_size_t] + [_c_int] * 6 + [_c_void_p]),
    "as_refine_similarity_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "as_refine_similarity": (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_float, _c_int] + [_c_void_p] * 3
                             + [_c_size_t] + [_c_int] * 3 + [_c_void_p]),
    "as_instance_maps_workspace_bytes": (_c_size_t, [_c_int] * 2),
    "as_crop_threshold_erode_workspace_bytes": (_c_size_t, [_c_int] * 3),
    "as_crop_threshold_erode": (_c_int, [_c_void_p] * 2 + [_c_float, _c_int, _c_int] + [_c_void_p] * 3 + [_c_size_t]
                                + [_c_int] * 3 + [_c_void_p]),
    "as_rank_select_workspace_bytes": (_c_size_t, [_c_int] * 2),
    "as_rank_select": (_c_int, [_c_void_p] * 4 + [_c_size_t] + [_c_int] * 3 + [_c_void_p]),
    "as_rank_select_xy": (_c_int, [_c_void_p] * 4 + [_c_size_t] + [_c_int] * 5 + [_c_void_p]),
    "as_merge_parts": (_c_int, [_c_void_p] * 2 + [_c_float] + [_c_void_p] * 3 + [_c_int] * 4 + [_c_void_p]),
    "as_select_median_boxes": (_c_int, [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p] * 8 + [_c_int, _c_void_p]),
    "as_rank_draw_xy": (_c_int, [_c_void_p, _c_int] + [_c_void_p] * 4 + [_c_int, _c_int, ctypes.c_int64, _c_int, _c_void_p, _c_size_t]
                        + [_c_int] * 5 + [_c_void_p]),
    "as_instance_maps": (_c_int, [_c_void_p] * 2 + [_c_int] * 6 + [_c_void_p] * 3 + [_c_size_t, _c_void_p]),
}

_lib = None


class AttnShiftError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AttnShiftError(
            f"{LIB_PATH} is missing: build it with `python -m attentionshift_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().as_last_error()
        raise AttnShiftError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
