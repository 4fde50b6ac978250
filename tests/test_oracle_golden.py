"""Pin the CPU oracle (oracle/attnshift_oracle.py) against the golden vectors that
tools/gen_golden.py produced by executing the reference's own functions.

CPU-only: these run in the build container and on the GPU box alike.  Tolerances: fp32 1e-5
relative for floats (oracle and reference are both torch CPU fp32, differing only in summation
order of cosine similarities); bit-exact for every integer / index / mask output.
"""
import numpy as np
import pytest
import torch

import attnshift_oracle as O
from helpers import assert_close, assert_equal, backbone_cfg, backbone_state_dict, shift_case_inputs, t

torch.set_grad_enabled(False)


@pytest.mark.parametrize("tag", ["small", "tiny224"])
def test_backbone_forward_matches_reference(golden, tag):
    g = golden(f"backbone_{tag}")
    cfg = backbone_cfg(g)
    sd = backbone_state_dict(g)
    from attentionshift_amd import synthetic
    img = synthetic.images(cfg["batch"], *cfg["img_hw"], seed=cfg["seed"])
    out = O.backbone_forward(img, sd, patch_size=16, depth=cfg["depth"], num_heads=cfg["num_heads"],
                             out_indices=cfg["out_indices"], point_tokens_num=cfg["point_tokens_num"])
    for k in ("last_feat", "point_tokens", "outputs_class", "outputs_coord"):
        assert_close(t(g[k]), out[k], 1e-5, 1e-6, k)
    for i, f in enumerate(out["feature"]):
        st = int(g[f"feature{i}_stride"])
        assert_close(t(g[f"feature{i}"]), f[:, :, ::st, ::st], 1e-5, 1e-6, f"feature{i}")
    T, Lc = cfg["point_tokens_num"], cfg["cam_layer"]
    for key in g.files:
        if key.startswith("attn") and key[4:].isdigit():
            assert_close(t(g[key]), out["attns"][int(key[4:])], 1e-5, 1e-7, key)
    # A3: the row-sliced roll-out equals the rows of the reference's dense roll-out
    assert_close(t(g["rollout_rows"]), O.rollout_rows(out["attns"][-Lc:], T), 1e-5, 1e-7, "rollout_rows")
    assert_close(t(g["rollout_rows"]), O.rollout_full(out["attns"][-Lc:])[:, :, -T:, :], 1e-5, 1e-7, "rollout_full")


def test_upsample_formula_is_bit_identical_to_torch():
    gen = torch.Generator().manual_seed(5)
    for shape, out in (((3, 2, 14, 14), (224, 224)), ((2, 64, 64), (1024, 1024)), ((1, 9, 7), (144, 112))):
        x = torch.randn(*shape, generator=gen)
        assert_equal(O.upsample_bilinear(x, *out), O.upsample_bilinear_explicit(x, *out), f"upsample {shape}")


def test_ccl_canonical_labels():
    a = np.zeros((6, 8), np.uint8)
    a[0, 0] = a[1, 1] = 1          # diagonal neighbours: one component under 8-connectivity
    a[0, 5:8] = 1
    a[4:6, 3] = 1
    lab = O.ccl_labels(a)
    assert lab[0, 0] == 1 and lab[1, 1] == 1
    assert (lab[0, 5:8] == 6).all()
    assert (lab[4:6, 3] == 4 * 8 + 3 + 1).all()
    assert (lab[a == 0] == 0).all()
    assert O.ccl_labels(np.zeros((4, 4), np.uint8)).sum() == 0


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_cam_boxes_match_reference(golden, tag):
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    boxes, _ = O.cam_boxes_from_rollout(inp["cams"], inp["points"], float(g["cam_thr"]), float(g["area_ratio"]))
    assert_equal(t(g["ref_boxes"]), boxes, "cam boxes")


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_refine_and_mask_points_match_reference(golden, tag):
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    hp, wp, G = int(g["hp"]), int(g["wp"]), int(g["G"])
    H, W = hp * 16, wp * 16
    cams = O.upsample_bilinear(inp["cams"], H, W)
    best, rois = t(g["best_idx"]), t(g["rois"])
    attn_sel = cams[best, torch.arange(G)]
    torch.manual_seed(int(g["seed"]) + 1)
    fg_pts, bg_pts = O.sample_refine_inputs(attn_sel, inp["points"])
    assert_equal(t(g["points_fg"]), fg_pts, "sampled fg points")
    assert_equal(t(g["points_bg"]), bg_pts, "sampled bg points")
    m_fg, m_bg, f_fg, f_bg = O.cosine_refined_maps(attn_sel, inp["vit_feat"], rois, fg_pts, bg_pts, 2, float(g["obj_tau"]))
    assert_close(t(g["map_fg_last"]), m_fg[-1], 1e-4, 1e-5, "map_fg[-1]")
    assert_close(t(g["map_bg_last"]), m_bg[-1], 1e-4, 1e-5, "map_bg[-1]")
    assert_close(t(g["map_fg_sub"]), m_fg[:, :, ::4, ::4], 1e-4, 1e-5, "map_fg levels")
    assert_close(t(g["map_bg_sub"]), m_bg[:, :, ::4, ::4], 1e-4, 1e-5, "map_bg levels")
    assert_close(t(g["fg_feat"]), f_fg, 1e-5, 1e-5, "fg_feat")
    assert_close(t(g["bg_feat"]), f_bg, 1e-5, 1e-5, "bg_feat")
    # B2': same RNG stream continues; later stages are fed the reference's maps (stage-wise pinning)
    coords, labels = O.mask_sample_points(t(g["map_fg_last"]), t(g["map_bg_last"]), rois, float(g["pos_thr"]),
                                          float(g["neg_thr"]), int(g["num_gt"]), int(g["corr_size"]))
    assert_equal(t(g["mask_coords"]), coords, "mask point coords")
    assert_equal(t(g["mask_labels"]), labels, "mask point labels")
    assert_equal((t(g["map_fg_last"]) > t(g["map_fg_last"]).flatten(1).max(1)[0][:, None, None] * float(g["pos_thr"])).numpy().astype(np.uint8),
                 O.pseudo_masks(t(g["map_fg_last"]), float(g["pos_thr"])), "pseudo masks")


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_semantic_centers_match_reference(golden, tag):
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    rois = t(g["rois"])
    trace = []
    res = O.semantic_centers(t(g["map_fg_last"]), t(g["map_bg_last"]), rois, inp["vit_feat"], float(g["pos_thr"]),
                             int(g["n_shift"]), inp["labels"], num_semantic_points=int(g["num_semantic_points"]),
                             trace=trace)
    assert len(trace) == int(g["n_shift"])
    for it, (assign, tau) in enumerate(trace):
        assert_equal(t(g["ref_assign"][it]), assign.int(), f"cluster assignment it{it}")       # bit-exact argmax
        assert_close(t(g["ref_tau"][it]), tau, 1e-3, 2e-6, f"tau it{it}")                       # 1-cos noise floor
    assert_close(t(g["ref_prot"]), res["prot"], 1e-3, 1e-4, "prototypes")
    assert_close(t(g["ref_sim"]), res["sim"], 1e-4, 1e-5, "sim maps")
    assert_equal(g["num_parts"], np.array(res["num_parts"]), "num_parts")
    assert_equal(g["corres_gt"], res["corres_gt"], "corres_gt")
    assert_close(t(g["coords_org"]), res["coords_org"], 0, 0, "centre coords")
    assert_equal(g["labels_org"], res["labels_org"], "centre labels")
    for i, n in enumerate(g["n_sim_parts"].tolist()):
        if n:
            assert_close(t(g[f"sim_parts{i}"]), res["sim_parts"][i], 1e-4, 1e-5, f"sim_parts{i}")
        else:
            assert res["sim_parts"][i].numel() == 0


def test_cosine_shift_duplicate_seed_ties_resolve_to_lowest_index():
    """Structural ties: duplicated seeds give identical softmax rows, and
    argmax over prototypes must return the LOWEST index, leaving the duplicates empty."""
    gen = torch.Generator().manual_seed(9)
    feats = torch.randn(1, 50, 16, generator=gen)
    prot = feats[0, [3, 3, 7, 3]].clone()[None]          # prototypes 0,1,3 identical
    trace = []
    O.cosine_shift(prot, feats, feats[0], n_shift=1, trace=trace)
    assign = trace[0][0][0]
    assert set(assign.tolist()) <= {0, 2}
    assert trace[0][1][0, 1] == 1.0 and trace[0][1][0, 3] == 1.0      # empty cluster -> tau = 1


@pytest.mark.parametrize("tag", ["w14_s0", "w14_s3", "w16_s3_pad", "w9_s0_pad"])
def test_swin_block_oracle_matches_reference(golden, tag):
    """A6: the oracle's restatement of SwinTransformerBlock.forward (pad / shift / partition / WindowAttention with
    relative-position bias and shift mask / reverse) vs outputs of the reference block itself."""
    g = golden(f"swin_{tag}")
    p = {k[2:]: t(g[k]) for k in g.files if k.startswith("p.")}
    hw, ws, shift, heads = int(g["hw"]), int(g["ws"]), int(g["shift"]), int(g["heads"])
    y, attn = O.swin_block(t(g["x"]), p, heads, ws, shift)
    assert_close(t(g["y"]), y, 1e-5, 1e-6, "swin block output")
    assert_close(t(g["attn"]), attn, 1e-5, 1e-7, "window attention probabilities")
    if shift > 0:
        assert_equal(t(g["attn_mask"]), O.swin_attn_mask(hw, hw, ws, shift), "shift mask")
