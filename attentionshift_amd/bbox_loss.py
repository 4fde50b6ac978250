"""Box-head training pieces of the reference's RoI head (host-side tensor logic):

    bbox2delta / delta2bbox   mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:80-130, 133-230 (DeltaXYWHBBoxCoder,
                              configs/mae/attnshift_voc12aug.py:106-109: means 0, stds (0.1, 0.1, 0.2, 0.2))
    bbox_targets              BBoxHead._get_target_single / get_targets as used by MAEBoxHeadRec (background label =
                              num_classes, positives regress to the encoded deltas of their GT box)
    bbox_head_loss            mae_bbox_head_rec.py:170-228: cross entropy over K + 1 classes normalised by the number of
                              weighted samples, L1 on the positives' deltas of THEIR class normalised by the number of
                              samples, top-1 accuracy
    giou_loss                 mmdet/models/losses/iou_loss.py:87-102 + iou2d_calculator.py:127-158 (aligned GIoU, the
                              union and the enclosing area clamped to eps); the shipped config regresses DECODED boxes
                              with it (reg_decoded_bbox=True, GIoULoss weight 10, attnshift_voc12aug.py:110-114)
IoU assignment and sampling are in attentionshift_amd.assign; proposal generation (mmdet's RPN) is not part of this build.
"""
import math

import torch
import torch.nn.functional as F


_CONSTS = {}


def _const(values, like, repeat=1):
    """A small constant vector on `like`'s device / dtype, uploaded once (a `new_tensor(list)` on a GPU is a pageable
    host -> device copy that waits for the stream)."""
    key = (tuple(float(v) for v in values), repeat, like.device, like.dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(key[0] * repeat, dtype=like.dtype).to(like.device)
    return t


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    proposals, gt = proposals.float(), gt.float()
    px, py = (proposals[..., 0] + proposals[..., 2]) * 0.5, (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw, ph = proposals[..., 2] - proposals[..., 0], proposals[..., 3] - proposals[..., 1]
    gx, gy = (gt[..., 0] + gt[..., 2]) * 0.5, (gt[..., 1] + gt[..., 3]) * 0.5
    gw, gh = gt[..., 2] - gt[..., 0], gt[..., 3] - gt[..., 1]
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return (deltas - _const(means, deltas)) / _const(stds, deltas)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None, wh_ratio_clip=16 / 1000):
    """rois [N,4], deltas [N, 4 * k] -> boxes [N, 4 * k] (clipped to max_shape = (H, W[, C]) when given)."""
    k = deltas.size(-1) // 4
    d = deltas * _const(stds, deltas, k) + _const(means, deltas, k)
    dx, dy, dw, dh = d[..., 0::4], d[..., 1::4], d[..., 2::4], d[..., 3::4]
    max_ratio = abs(math.log(wh_ratio_clip))
    dw, dh = dw.clamp(-max_ratio, max_ratio), dh.clamp(-max_ratio, max_ratio)
    px = ((rois[..., 0] + rois[..., 2]) * 0.5).unsqueeze(-1)
    py = ((rois[..., 1] + rois[..., 3]) * 0.5).unsqueeze(-1)
    pw = (rois[..., 2] - rois[..., 0]).unsqueeze(-1)
    ph = (rois[..., 3] - rois[..., 1]).unsqueeze(-1)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    if max_shape is not None:
        x1, x2 = x1.clamp(0, max_shape[1]), x2.clamp(0, max_shape[1])
        y1, y2 = y1.clamp(0, max_shape[0]), y2.clamp(0, max_shape[0])
    return torch.stack([x1, y1, x2, y2], dim=-1).flatten(-2)


def giou_loss(pred, target, eps=1e-6):
    """1 - GIoU of aligned xyxy boxes [n,4] -> [n]."""
    area_p = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    area_t = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    wh = (torch.min(pred[:, 2:], target[:, 2:]) - torch.max(pred[:, :2], target[:, :2])).clamp(min=0)
    overlap = wh[:, 0] * wh[:, 1]
    union = (area_p + area_t - overlap).clamp(min=eps)
    enc = (torch.max(pred[:, 2:], target[:, 2:]) - torch.min(pred[:, :2], target[:, :2])).clamp(min=0)
    enc_area = (enc[:, 0] * enc[:, 1]).clamp(min=eps)
    return 1 - (overlap / union - (enc_area - union) / enc_area)


def bbox_targets(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, num_classes, means=(0., 0., 0., 0.),
                 stds=(0.1, 0.1, 0.2, 0.2), pos_weight=-1, reg_decoded_bbox=False):
    """Per-image lists -> concatenated (labels, label_weights, bbox_targets [n,4], bbox_weights [n,4]); positives first
    inside every image, as mmdet's samplers order them."""
    out = ([], [], [], [])
    for pb, nb, gb, gl in zip(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels):
        n_pos, n = pb.shape[0], pb.shape[0] + nb.shape[0]
        labels = pb.new_full((n,), num_classes, dtype=torch.long)
        lw, bt, bw = pb.new_zeros(n), pb.new_zeros(n, 4), pb.new_zeros(n, 4)
        if n_pos:
            labels[:n_pos] = gl
            lw[:n_pos] = 1.0 if pos_weight <= 0 else pos_weight
            bt[:n_pos] = gb if reg_decoded_bbox else bbox2delta(pb, gb, means, stds)
            bw[:n_pos] = 1
        lw[n_pos:] = 1.0
        for lst, v in zip(out, (labels, lw, bt, bw)):
            lst.append(v)
    return tuple(torch.cat(v) for v in out)


def bbox_head_loss(cls_score, bbox_pred, labels, label_weights, bbox_targets_, bbox_weights, num_classes,
                   reg_class_agnostic=False, loss_cls_weight=1.0, loss_bbox_weight=1.0, rois=None, loss_bbox_type="L1Loss",
                   means=(0., 0., 0., 0.), stds=(0.1, 0.1, 0.2, 0.2), pos_index=None, num_weighted=None):
    """cls_score [n, K+1] | None, bbox_pred [n, 4 or 4K] | None -> dict(loss_cls, acc, loss_bbox).  With `rois` [n,4]
    the predictions are decoded against them first (reg_decoded_bbox) and `bbox_targets_` are absolute boxes.
    `pos_index` (long tensor: the rows with a foreground label, ascending) and `num_weighted` (the number of rows with
    a positive label weight) are what a caller that built the targets from host-side sampling results already knows;
    given, nothing here reads a count back from the device."""
    losses = {}
    if cls_score is not None and cls_score.numel() > 0:
        avg = max(float((label_weights > 0).sum()) if num_weighted is None else float(num_weighted), 1.0)
        ce = F.cross_entropy(cls_score, labels, reduction="none") * label_weights
        losses["loss_cls"] = loss_cls_weight * ce.sum() / avg
        losses["acc"] = (cls_score.argmax(1) == labels).float().mean() * 100.0
    if bbox_pred is not None:
        if pos_index is None:
            pos_index = torch.nonzero((labels >= 0) & (labels < num_classes), as_tuple=False).flatten()
        if pos_index.numel():
            if rois is not None:
                bbox_pred = delta2bbox(rois, bbox_pred, means, stds)
            pred = bbox_pred.view(bbox_pred.size(0), 4)[pos_index] if reg_class_agnostic else \
                bbox_pred.view(bbox_pred.size(0), -1, 4)[pos_index, labels[pos_index]]
            if loss_bbox_type == "GIoULoss":
                per = giou_loss(pred, bbox_targets_[pos_index]) * bbox_weights[pos_index].mean(-1)
            else:
                per = (pred - bbox_targets_[pos_index]).abs() * bbox_weights[pos_index]
            losses["loss_bbox"] = loss_bbox_weight * per.sum() / bbox_targets_.size(0)
        else:
            losses["loss_bbox"] = bbox_pred[pos_index].sum()
    return losses
