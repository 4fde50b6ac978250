"""Point-token loss of the RoI head (second half): how the backbone's `outputs_class` / `outputs_coord`
are supervised by the GT points.  Host-side tensor logic on a [B, 100, K] / [B, 100, 2] problem.

    targets   stdroi_point_deform_attn_reppoints.py:2568-2596 (Hungarian assignment per image, the helper already on the
              pseudo-label path) + _get_target_single :3332-3359 (background label = num_classes, every token weighs 1
              in the classification, matched tokens regress to their GT point)
    loss      :3430-3514: sigmoid focal loss (configs/mae/attnshift_voc12aug.py:116-121: gamma 2, alpha 0.25, weight 1)
              and L1 on coordinates divided by (W, H) (:115, weight 10), both normalised by the number of matched
              tokens averaged over the ranks (mmdet `reduce_mean`); `pos_point_acc` = top-1 accuracy of the matched.
"""
import torch
import torch.nn.functional as F

from .roi_head import hungarian_point_match


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, avg_factor=None):
    """mmdet's py_sigmoid_focal_loss: pred [n,K] logits, target [n] class index (K = background), weight [n] | None."""
    onehot = F.one_hot(target, pred.shape[1] + 1)[:, :pred.shape[1]].to(pred.dtype)
    p = pred.sigmoid()
    pt = (1 - p) * onehot + p * (1 - onehot)
    fw = (alpha * onehot + (1 - alpha) * (1 - onehot)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, onehot, reduction="none") * fw
    if weight is not None:
        loss = loss * weight.reshape(-1, 1).to(loss.dtype)
    return loss.sum() / avg_factor if avg_factor is not None else loss.mean()


def point_targets(point_cls, point_reg, gt_points, gt_labels, img_shapes, num_classes, point_pos_weight=1,
                  cls_cost=1.0, reg_cost=10.0):
    """Per image Hungarian match of the tokens to the GT points; returns flattened (labels [B*T] long, label_weights
    [B*T], point_targets [B*T,2], point_weights [B*T,2]) as get_targets(concat=True) does."""
    B, T = point_reg.shape[:2]
    dev = point_reg.device
    labels = torch.full((B, T), num_classes, dtype=torch.long, device=dev)
    label_w = torch.ones(B, T, device=dev)
    tgt = torch.zeros(B, T, 2, device=dev)
    tgt_w = torch.zeros(B, T, 2, device=dev)
    for i in range(B):
        pos, matched = hungarian_point_match(point_reg[i].detach(), point_cls[i].detach(), gt_points[i], gt_labels[i],
                                             img_shapes[i], cls_weight=cls_cost, reg_weight=reg_cost)
        if pos.numel():
            labels[i, pos] = gt_labels[i][matched]
            label_w[i, pos] = 1.0 if point_pos_weight <= 0 else float(point_pos_weight)
            tgt[i, pos] = gt_points[i][matched].to(tgt.dtype)
            tgt_w[i, pos] = 1.0
    return labels.flatten(), label_w.flatten(), tgt.flatten(0, 1), tgt_w.flatten(0, 1)


def point_token_loss(point_cls, point_reg, gt_points, gt_labels, img_shapes, num_classes=20, loss_point_weight=10.0,
                     loss_cls_weight=1.0, gamma=2.0, alpha=0.25, ranks=None, **assign_kw):
    """point_cls [B,T,K] logits, point_reg [B,T,2] in (0,1) -> dict(loss_point_cls, loss_point, pos_point_acc)."""
    B, T, K = point_cls.shape
    labels, label_w, tgt, tgt_w = point_targets(point_cls, point_reg, gt_points, gt_labels, img_shapes, num_classes, **assign_kw)
    cls_score, pred = point_cls.reshape(-1, K).float(), point_reg.reshape(-1, 2)
    pos = (labels >= 0) & (labels < num_classes)
    num_pos = pos.sum().float()
    if ranks is not None and getattr(ranks, "world", 1) > 1:           # mmdet reduce_mean
        num_pos = torch.as_tensor(ranks.sum_over_ranks(float(num_pos)) / ranks.world, device=pred.device)
    avg = num_pos.clamp(min=1e-6) if float(num_pos) == 0 else num_pos
    out = dict(loss_point_cls=loss_cls_weight * sigmoid_focal_loss(cls_score, labels, label_w, gamma, alpha, avg_factor=avg))
    out["pos_point_acc"] = ((cls_score[pos].argmax(1) == labels[pos]).float().mean() * 100.0 if pos.any()
                            else cls_score.new_zeros(()))
    if pos.any():
        whwh = torch.cat([pred.new_tensor([s[1], s[0]]).expand(T, 2) for s in img_shapes])      # (W, H) per token
        l1 = (pred[pos] - tgt[pos] / whwh[pos]).abs() * tgt_w[pos]
        out["loss_point"] = loss_point_weight * l1.sum() / avg
    else:
        out["loss_point"] = pred.sum() * 0
    return out
