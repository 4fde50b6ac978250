"""Consumers of the pseudo labels: how `seed_pseudo_gt`'s mask points and semantic centres become the
point-supervised mask loss of the reference's RoI head.  Host-side tensor logic (tiny, ragged), no kernels:

    update_coords_with_semantic_centers   stdroi_point_deform_attn_reppoints.py:117-141
    get_point_coords_wrt_box              :1157-1180
    point_sample                          mmcv.ops.point_sample as called at :3154 (bilinear grid_sample on [0,1]^2)
    mask_point_targets                    the target half of _mask_forward_train :3094-3160
    point_mask_loss                       mae_mask_head_pointSup.py:234-275 (BCE on the sampled logits, label 2 = ignore)
"""
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence


def update_coords_with_semantic_centers(points_coords, points_labels, semantic_centers, labels_host=None):
    """Per image: keep the NEGATIVE mask points of every object and replace its positive ones by the object's
    semantic centres.  points_coords[i] [G,P,2], points_labels[i] [G,P] bool, semantic_centers[i] = list of [k_g,2]
    (empty list: the image is left as it is).  Ragged rows are padded with (-1,-1) / False, images to a common length.

    The row lengths depend on the labels, so a HOST copy of them decides the shapes (`labels_host`: per image a bool
    array / CPU tensor [G,P] or None; read back here, all images in one transfer, when not given); the device side is
    one gather per image."""
    from .roi_head import read_back, to_device
    todo = [i for i, c in enumerate(semantic_centers) if len(c)]
    if labels_host is None:
        labels_host = [None] * len(points_labels)
        for i, h in zip(todo, read_back([points_labels[i] for i in todo])):
            labels_host[i] = h
    new_coords, new_labels = [], []
    for i, (coords, labels, centers) in enumerate(zip(points_coords, points_labels, semantic_centers)):
        if len(centers) == 0:
            new_coords.append(coords)
            new_labels.append(labels)
            continue
        neg = ~torch.as_tensor(labels_host[i]).bool()                                            # host [G,P]
        G, maxneg = neg.shape[0], int(neg.sum(dim=1).max()) if neg.numel() else 0
        # row g of `order` lists the positions of object g's negatives first, in order (stable sort of the flags)
        order = torch.sort((~neg).to(torch.uint8), dim=1, stable=True)[1][:, :maxneg]
        valid = torch.gather(neg, 1, order)                                                      # [G, maxneg]
        order_d, valid_d = to_device(order, coords.device), to_device(valid, coords.device)
        rows = torch.arange(G, device=coords.device)[:, None]
        neg_coords = torch.where(valid_d[..., None], coords[rows, order_d], coords.new_full((), -1.0))   # [G, maxneg, 2]
        neg_labels = torch.zeros(G, maxneg, dtype=labels.dtype, device=labels.device)
        ctr_coords = pad_sequence(list(centers), padding_value=-1.0).transpose(0, 1)             # [G, maxk, 2]
        ctr_labels = torch.ones(ctr_coords.shape[:-1], dtype=neg_labels.dtype, device=neg_labels.device)
        new_coords.append(torch.cat((neg_coords, ctr_coords), dim=1))
        new_labels.append(torch.cat((neg_labels, ctr_labels), dim=1))
    width = max(c.shape[1] for c in new_coords)
    new_coords = [F.pad(c, (0, 0, 0, width - c.shape[1]), value=-1) for c in new_coords]
    new_labels = [F.pad(l, (0, width - l.shape[1]), value=False) for l in new_labels]
    return new_coords, new_labels


def get_point_coords_wrt_box(boxes_coords, point_coords):
    """Image coordinates [R,P,2] -> box-normalised [0,1]^2 coordinates of the boxes [R,4]."""
    with torch.no_grad():
        out = point_coords.clone()
        out[:, :, 0] -= boxes_coords[:, None, 0]
        out[:, :, 1] -= boxes_coords[:, None, 1]
        out[:, :, 0] = out[:, :, 0] / (boxes_coords[:, None, 2] - boxes_coords[:, None, 0])
        out[:, :, 1] = out[:, :, 1] / (boxes_coords[:, None, 3] - boxes_coords[:, None, 1])
    return out


def point_sample(inp, points, align_corners=False):
    """inp [N,C,H,W], points [N,P,2] in [0,1]^2 (x,y) -> [N,C,P]: bilinear samples (grid_sample on 2*p-1)."""
    grid = 2.0 * points.unsqueeze(2) - 1.0                                                       # [N,P,1,2]
    return F.grid_sample(inp, grid, align_corners=align_corners).squeeze(3)


def mask_point_targets(pos_bboxes, assigned_gt_inds, points_coords, points_labels, semantic_centers, literal=True,
                       labels_host=None):
    """Target half of _mask_forward_train (:3106, :3134-3152): per image i, `pos_bboxes[i]` [R_i,4] are the positive
    proposals and `assigned_gt_inds[i]` [R_i] their object; returns (sites [R,P,2] box-normalised, targets [R,P]).

    The reference marks points outside their proposal with `mask_targets[point_ignores] = 2` on a BOOL tensor, which
    stores True, so its loss (`mask_targets == 2`) never ignores anything and those points count as foreground.
    `literal=True` reproduces that (targets stay bool); `literal=False` gives the evident intent: long targets with
    1 = foreground, 0 = background, 2 = ignored."""
    coords, labels = update_coords_with_semantic_centers(points_coords, points_labels, semantic_centers, labels_host)
    boxes = torch.cat(list(pos_bboxes))
    sites = torch.cat([c[idx] for c, idx in zip(coords, assigned_gt_inds)])
    targets = torch.cat([l[idx] for l, idx in zip(labels, assigned_gt_inds)])
    if not literal:
        targets = targets.long()
    sites = get_point_coords_wrt_box(boxes, sites)
    ignore = (sites[:, :, 0] < 0) | (sites[:, :, 0] > 1) | (sites[:, :, 1] < 0) | (sites[:, :, 1] > 1)
    targets[ignore] = 2
    return sites, targets


def point_mask_loss(mask_pred, sites, targets, labels, class_agnostic=False, loss_weight=1.0):
    """Point-supervised mask loss (mae_mask_head_pointSup.py:253-273): mask_pred [R,K,h,w] logits sampled at `sites`,
    BCE against `targets` with weight 0 where targets == 2, mean over ALL points, times the head's loss weight."""
    if mask_pred.shape[0] == 0:
        return mask_pred.sum()
    point_preds = point_sample(mask_pred, sites, align_corners=False)                            # [R,K,P]
    cls = torch.zeros_like(labels) if class_agnostic else labels
    logits = point_preds[torch.arange(mask_pred.shape[0], device=mask_pred.device), cls]
    loss = F.binary_cross_entropy_with_logits(logits, targets.to(torch.float32), reduction="mean", weight=~(targets == 2))
    return loss * loss_weight
