"""The C-ABI library loads and exports every symbol include/attnshift.h declares (no compute calls:
this runs without a GPU).  Also checks the argument validation paths that return before any launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "attnshift.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(as_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from attentionshift_amd.csrc import build
    lib_path = build.build(verbose=False)
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in attnshift.h but not exported"
    from attentionshift_amd import _lib
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and header disagree"
    loaded = _lib.load()
    assert loaded.as_version() == 100


def test_every_entry_point_is_mapped_to_the_reference_in_integration_md():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in declared_symbols() if n not in doc and not n.endswith("_bytes")]
    assert not missing, f"entry points without a row in INTEGRATION.md: {missing}"


def test_bad_arguments_return_error_codes_without_launching():
    from attentionshift_amd import _lib
    lib = _lib.load()
    assert lib.as_npad(4197) == 4224 and lib.as_npad(64) == 64
    assert lib.as_linear_fwd(None, None, None, None, 1, 1, 32, 0, 0, None) == -1          # AS_E_BADARG
    assert b"null" in lib.as_last_error()
    assert lib.as_cosine_shift_workspace_bytes(2, 768, 64, 64, 6, 20) > 0
    assert lib.as_cosine_shift_workspace_bytes(0, 768, 64, 64, 6, 20) == 0
    assert lib.as_cam_boxes_workspace_bytes(21, 64, 64, 16) >= 21 * 1024 * 66 * 4 * 3        # runs + parents + areas
    with pytest.raises(_lib.AttnShiftError):
        _lib.check(-2, "unit test")
    # every entry point validates before it launches (no GPU needed to get the error code)
    assert lib.as_sdpa_fwd(None, None, None, None, None, None, 0, 1, 64, 1, 1, None) == -1
    assert lib.as_sdpa_bwd(None, None, None, None, None, None, None, None, 0, 1, 64, 1, 1, None) == -1
    assert lib.as_attn_bwd(*([None] * 15), 0, 1, 64, 64, 1, 1, None) == -1
    assert lib.as_window_attn_fwd(None, None, None, None, None, 1, 14, 14, 64, 2, 7, 0, 1, None) == -1
    assert lib.as_add_layernorm(None, None, None, None, 1e-6, None, None, 4, 64, 1, None) == -1
    assert lib.as_mask_count(None, None, 1, 16, None) == -1
    assert lib.as_linear_bwd(None, None, None, None, None, None, 8, 32, 32, 1, 0, None, 0, None) == -1
    assert lib.as_linear_bwd_workspace_bytes(0, 32, 32) == 0
    # W^T + dy^T + x^T (rows padded to 64) + the column-sum partials + one fp32 split-K partial of a 1-tile output
    assert lib.as_linear_bwd_workspace_bytes(100, 64, 32) >= 2 * (32 * 64 + 64 * 128 + 32 * 128) + 64 * 64 * 4 + 64 * 32 * 4
    assert lib.as_linear_splitk_workspace_bytes(768, 768, 8448) == 7 * 768 * 768 * 4       # 36 tiles -> 7 ranges on 256 CUs
    assert lib.as_merge_plan(None, None, None, None, 3, 20, None) == -1
    assert lib.as_small_attn_fwd(None, None, None, 4, 50, 8, 32, 1, None) == -1
    assert lib.as_small_attn_bwd(None, None, None, None, None, None, 0, 4, 50, 8, 32, 1, None) == -1
    assert lib.as_small_attn_bwd_workspace_bytes(4, 50, 8) == 4 * 8 * 50 * 4
    assert lib.as_chamfer_2d_fwd(None, None, None, None, None, None, 2, 9, 7, None) == -1
    assert lib.as_chamfer_2d_bwd(*([None] * 8), 2, 9, 7, None) == -1
    assert lib.as_filter_parts(None, None, 0.8, 0.85, None, 3, 20, 4096, None) == -1
    assert lib.as_draw_distinct(None, 2, 1, None, None, None, None, None, 3, 32, 10, None) == -1
    assert lib.as_part_stats(None, None, None, 16.0, None, None, None, None, 4, 64, 64, None) == -1
    assert lib.as_mask_candidates(None, None, None, 0.35, 0.8, 0.35, 21, None, None, None, None, None, 0, 3, 64, 64, None) == -1
    assert lib.as_semantic_prestage(None, 0.35, 11, 3, 64, 64, 16, None, None, None, None) == -1
    assert lib.as_cam_sample_masks(None, None, None, 3, 64, 64, 16, 0.1, 0.2, None, None, None, 0, None) == -1
    assert lib.as_cam_sample_masks_workspace_bytes(3, 64, 64, 16) == 0
    assert lib.as_cam_sample_masks_workspace_bytes(12, 64, 64, 16) == 4 * 1024 * 1024
    assert lib.as_rollout_step(*([None] * 8), 0, 1, 64, 1, 8, 1, None) == -1
    # workspace size queries are pure host functions
    assert lib.as_sdpa_bwd_workspace_bytes(2, 4197, 12, 1) == 2 * 12 * 4224 * (5 * 64 * 2 + 8)      # 5 operand copies + delta + lse2
    assert lib.as_attn_bwd_workspace_bytes(2, 4197, 768, 12, 1) > lib.as_sdpa_bwd_workspace_bytes(2, 4197, 12, 1)
    assert lib.as_rollout_step_workspace_bytes(2, 4197, 100) == 16 * 2 * 100 * 4197 * 4      # up to 16 partial products
    assert lib.as_sdpa_bwd_workspace_bytes(0, 4197, 12, 1) == 0


def test_product_path_refuses_cpu_tensors():
    import torch
    from attentionshift_amd import ops
    with pytest.raises(ops.AttnShiftError):
        ops.linear(torch.zeros(4, 32), torch.zeros(8, 32))
