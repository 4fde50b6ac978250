"""Restated mmdet helpers (box coder, IoU / GIoU, max-IoU assignment, focal / cross-entropy losses, accuracy, mask
pasting) against fixtures produced by EXECUTING the reference's own functions (tools/gen_golden_mmdet_pure.py).  CPU."""
import torch
import torch.nn.functional as F

from attentionshift_amd import assign as AS, bbox_loss as BL, inference as I, point_loss as PL


def test_coder_iou_assigner_losses_and_paste_equal_the_reference(golden):
    g = golden("mmdet_pure")
    t = lambda k: torch.from_numpy(g[k])
    stds = (0.1, 0.1, 0.2, 0.2)
    props, gts = t("props"), t("gts")
    assert torch.allclose(BL.bbox2delta(props, gts, stds=stds), t("deltas"), atol=1e-5)
    assert torch.allclose(BL.delta2bbox(props, t("d4k"), stds=stds, max_shape=(260, 340, 3)), t("decoded_clip"), atol=1e-3)
    assert torch.allclose(BL.delta2bbox(props, t("d4k")[:, :4] * 4, stds=stds), t("decoded_free"), rtol=1e-5, atol=1e-2)
    a, b = t("ov_a"), t("ov_b")
    assert torch.allclose(AS.bbox_overlaps(a, b), t("iou"), atol=1e-6)
    assert torch.allclose(1 - BL.giou_loss(a[:9], b), t("giou_aligned"), atol=1e-6)
    # assignment: proposals `a` against ground truths `b`
    got, _ = AS.max_iou_assign(a, b, 0.5, 0.5, 0.5, False)
    assert torch.equal(got, t("assign_rcnn"))
    got_rpn, _ = AS.max_iou_assign(a, b, 0.7, 0.3, 0.3, True)
    assert torch.equal(got_rpn, t("assign_rpn"))
    assert torch.equal(AS.max_iou_assign(a, b[:0])[0], t("assign_nogt"))
    # losses
    fl = PL.sigmoid_focal_loss(t("fl_logits"), t("fl_target"), t("fl_weight"), 2.0, 0.25, avg_factor=7.0)
    assert torch.allclose(fl, t("fl_out"), rtol=1e-5)
    target, w = t("fl_target"), t("fl_weight")
    ce = (F.cross_entropy(t("ce_logits"), target, reduction="none") * w).sum() / 33.0
    assert torch.allclose(ce, t("ce_out"), rtol=1e-5)
    out = BL.bbox_head_loss(t("ce_logits"), None, target, w, None, None, 20)
    assert torch.allclose(out["loss_cls"], t("ce_out") * 33.0 / max(float((w > 0).sum()), 1.0), rtol=1e-5)
    assert torch.allclose(out["acc"], t("acc_out")[0], atol=1e-4)
    # mask pasting
    assert torch.allclose(I.paste_masks(t("pm_masks"), t("pm_boxes"), 150, 170), t("pm_out"), atol=1e-6)
