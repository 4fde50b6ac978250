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


def point_matches(point_cls, point_reg, gt_points, gt_labels, img_shapes, cls_cost=1.0, reg_cost=10.0, costs_host=None):
    """Hungarian match of every image's tokens to its GT points -> list of (token indices ascending, GT index of each),
    numpy int64 on the HOST.  All images' cost matrices are built on the device and read back together (one host sync);
    `costs_host` (list of host [T, G_i] matrices, None for an image without points) skips that readback."""
    import numpy as np
    from .roi_head import hungarian_rows_cols, point_match_cost, read_back
    B = point_reg.shape[0]
    live = [i for i in range(B) if gt_points[i].shape[0] and point_reg.shape[1]]
    if costs_host is None:
        costs = [point_match_cost(point_reg[i].detach(), point_cls[i].detach(), gt_points[i], gt_labels[i], img_shapes[i],
                                  cls_weight=cls_cost, reg_weight=reg_cost) for i in live]
        costs_host = [None] * B
        for i, c in zip(live, read_back(costs)):
            costs_host[i] = c
    empty = np.zeros(0, dtype=np.int64)
    return [hungarian_rows_cols(np.asarray(costs_host[i])) if i in live else (empty, empty) for i in range(B)]


def point_targets(point_cls, point_reg, gt_points, gt_labels, img_shapes, num_classes, point_pos_weight=1,
                  cls_cost=1.0, reg_cost=10.0, matches=None):
    """Per image Hungarian match of the tokens to the GT points; returns flattened (labels [B*T] long, label_weights
    [B*T], point_targets [B*T,2], point_weights [B*T,2]) as get_targets(concat=True) does.  `matches`: the result of
    `point_matches` when the caller already has it."""
    from .roi_head import to_device
    B, T = point_reg.shape[:2]
    dev = point_reg.device
    if matches is None:
        matches = point_matches(point_cls, point_reg, gt_points, gt_labels, img_shapes, cls_cost, reg_cost)
    labels = torch.full((B, T), num_classes, dtype=torch.long, device=dev)
    label_w = torch.ones(B, T, device=dev)
    tgt = torch.zeros(B, T, 2, device=dev)
    tgt_w = torch.zeros(B, T, 2, device=dev)
    for i, (pos_h, matched_h) in enumerate(matches):
        if len(pos_h):
            pos, matched = to_device(pos_h, dev, torch.long), to_device(matched_h, dev, torch.long)
            labels[i, pos] = gt_labels[i][matched]
            # (index_fill_: a scalar on the right of an indexed assignment is uploaded as a tensor, which waits)
            label_w[i].index_fill_(0, pos, 1.0 if point_pos_weight <= 0 else float(point_pos_weight))
            tgt[i, pos] = gt_points[i][matched].to(tgt.dtype)
            tgt_w[i].index_fill_(0, pos, 1.0)
    return labels.flatten(), label_w.flatten(), tgt.flatten(0, 1), tgt_w.flatten(0, 1)


def point_token_loss(point_cls, point_reg, gt_points, gt_labels, img_shapes, num_classes=20, loss_point_weight=10.0,
                     loss_cls_weight=1.0, gamma=2.0, alpha=0.25, ranks=None, matches=None, point_pos_weight=1,
                     cls_cost=1.0, reg_cost=10.0):
    """point_cls [B,T,K] logits, point_reg [B,T,2] in (0,1) -> dict(loss_point_cls, loss_point, pos_point_acc).
    The matched tokens are known on the host after the Hungarian match, so everything after it is index arithmetic on
    host arrays + gathers on the device: the match's cost readback is the only host sync."""
    import numpy as np
    from .roi_head import to_device
    B, T, K = point_cls.shape
    if matches is None:
        matches = point_matches(point_cls, point_reg, gt_points, gt_labels, img_shapes, cls_cost, reg_cost)
    labels, label_w, tgt, tgt_w = point_targets(point_cls, point_reg, gt_points, gt_labels, img_shapes, num_classes,
                                                point_pos_weight, cls_cost, reg_cost, matches=matches)
    cls_score, pred = point_cls.reshape(-1, K).float(), point_reg.reshape(-1, 2)
    # the positives (0 <= label < num_classes) are the matched tokens: flat indices in ascending order, as a mask gives
    pos_h = np.concatenate([i * T + m[0] for i, m in enumerate(matches)]) if B else np.zeros(0, dtype=np.int64)
    n_pos = float(len(pos_h))
    if ranks is not None and getattr(ranks, "world", 1) > 1:           # mmdet reduce_mean
        n_pos = ranks.sum_over_ranks(n_pos) / ranks.world
    avg = max(n_pos, 1e-6) if n_pos == 0 else n_pos
    out = dict(loss_point_cls=loss_cls_weight * sigmoid_focal_loss(cls_score, labels, label_w, gamma, alpha, avg_factor=avg))
    if len(pos_h):
        pos = to_device(pos_h, pred.device, torch.long)
        out["pos_point_acc"] = (cls_score[pos].argmax(1) == labels[pos]).float().mean() * 100.0
        wh = np.concatenate([np.tile(np.asarray([[img_shapes[i][1], img_shapes[i][0]]], dtype=np.float32), (len(m[0]), 1))
                             for i, m in enumerate(matches)])                       # (W, H) of each matched token's image
        l1 = (pred[pos] - tgt[pos] / to_device(wh, pred.device, pred.dtype)).abs() * tgt_w[pos]
        out["loss_point"] = loss_point_weight * l1.sum() / avg
    else:
        out["pos_point_acc"] = cls_score.new_zeros(())
        out["loss_point"] = pred.sum() * 0
    return out
