"""Box / mask head training losses vs the REFERENCE's own get_targets / loss functions executed on seeded inputs
(tools/gen_golden_head_losses.py -> tests/golden/head_losses.npz).  CPU only."""
import types

import torch

import attentionshift_amd as A
from attentionshift_amd import bbox_loss as BL, mae_heads


def test_box_targets_and_losses_equal_the_reference(golden):
    g = golden("head_losses")
    t = lambda k: torch.from_numpy(g[k])
    K = int(g["K"])
    res = [types.SimpleNamespace(pos_bboxes=t(f"pos_bboxes{i}").reshape(-1, 4), neg_bboxes=t(f"neg_bboxes{i}").reshape(-1, 4),
                                 pos_gt_bboxes=t(f"pos_gt_bboxes{i}").reshape(-1, 4), pos_gt_labels=t(f"pos_gt_labels{i}"))
           for i in range(int(g["n_img"]))]
    for tag, decoded, loss_cfg in (("giou", True, dict(type="GIoULoss", loss_weight=10.0)), ("l1", False, dict(type="L1Loss", loss_weight=1.0))):
        head = A.build_head(dict(type="MAEBoxHeadRec", in_channels=32, embed_dim=32, depth=1, num_heads=1, num_classes=K,
                                 with_reconstruct=False, reg_decoded_bbox=decoded, loss_bbox=loss_cfg,
                                 bbox_coder=dict(type="DeltaXYWHBBoxCoder", target_means=[0.] * 4, target_stds=[.1, .1, .2, .2])))
        targets = head.get_targets(res)
        for got, name in zip(targets, ("labels", "label_weights", "bbox_targets", "bbox_weights")):
            want = t(f"{tag}_{name}")
            assert got.shape == want.shape and torch.allclose(got.float(), want.float(), atol=1e-5), (tag, name)
        out = head.loss(t("cls_score"), t("bbox_pred"), t("rois"), *targets)
        for k in ("loss_cls", "acc", "loss_bbox"):
            assert abs(float(out[k]) - float(g[f"{tag}_{k}"][0])) <= 2e-5 * max(1.0, abs(float(g[f"{tag}_{k}"][0]))), (tag, k)


def test_mask_point_loss_equals_the_reference(golden):
    g = golden("head_losses")
    t = lambda k: torch.from_numpy(g[k])
    K = int(g["K"])
    head = mae_heads.MAEMaskHeadPointSup(num_classes=K, in_channels=32, embed_dim=32, depth=1, num_heads=1)
    for kind in ("bool", "long"):
        got = head.loss(t("mask_pred"), t(f"mask_tgt_{kind}"), t("mask_labels"))["loss_mask"]
        assert abs(float(got) - float(g[f"mask_loss_{kind}"][0])) < 1e-6, kind
    empty = head.loss(t("mask_pred")[:0], t("mask_tgt_long")[:0], t("mask_labels")[:0])["loss_mask"]
    assert float(empty) == float(g["mask_loss_empty"][0]) == 0.0
