"""Test-time post-processing of the RoI head (host-side tensor logic; restated from their definitions because mmcv's
ops are absent here):

    nms / multiclass_nms   mmcv.ops.nms (greedy, IoU without the +1 offset) + mmdet/core/post_processing/bbox_nms.py:
                           scores above score_thr, class-aware suppression through per-class coordinate offsets, the
                           max_per_img best detections (configs/mae/attnshift_voc12aug.py:200-204: 0.05 / IoU 0.5 / 100)
    get_det_bboxes         BBoxHead.get_bboxes: softmax scores, DeltaXYWH decode clipped to the image, rescale
    paste_masks            fcn_mask_head._do_paste_mask: the RoI-sized mask resampled onto the image grid of its box
    get_seg_masks          mae_mask_head_pointSup.py:277-375: sigmoid, paste, threshold, per-class lists
    bbox2result            mmdet/core/bbox/transforms.py: per-class [n,5] arrays
"""
import numpy as np
import torch
import torch.nn.functional as F

from .assign import bbox_overlaps
from .bbox_loss import delta2bbox


def nms(boxes, scores, iou_threshold):
    """Greedy NMS; returns kept indices sorted by decreasing score."""
    if boxes.numel() == 0:
        return torch.zeros(0, dtype=torch.long, device=boxes.device)
    order = scores.argsort(descending=True, stable=True)
    iou = bbox_overlaps(boxes[order], boxes[order], eps=1e-12).cpu()
    n = iou.shape[0]
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if alive[i]:
            keep.append(i)
            alive &= ~(iou[i] > iou_threshold)
            alive[i] = False
    return order[torch.tensor(keep, dtype=torch.long, device=boxes.device)]


def multiclass_nms(multi_bboxes, multi_scores, score_thr, iou_threshold, max_num=-1):
    """multi_bboxes [n, 4K] or [n, 4], multi_scores [n, K+1] (last column = background) ->
    (dets [m,5] sorted by score, labels [m])."""
    K = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), multi_bboxes.shape[1] // 4, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), K, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(K, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    valid = scores > score_thr
    bboxes, scores, labels = bboxes[valid], scores[valid], labels[valid]
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros(0, 5), labels
    offsets = labels.to(bboxes) * (bboxes.max() + 1)                        # boxes of different classes never overlap
    keep = nms(bboxes + offsets[:, None], scores, iou_threshold)
    if max_num > 0:
        keep = keep[:max_num]
    return torch.cat((bboxes[keep], scores[keep, None]), dim=1), labels[keep]


def get_det_bboxes(rois, cls_score, bbox_pred, img_shape, scale_factor, rescale, score_thr=0.05, iou_threshold=0.5,
                   max_per_img=100, means=(0., 0., 0., 0.), stds=(0.1, 0.1, 0.2, 0.2)):
    """rois [n,5] (batch index first) of ONE image -> (det_bboxes [m,5], det_labels [m])."""
    scores = F.softmax(cls_score, dim=-1) if cls_score is not None else None
    if bbox_pred is not None:
        bboxes = delta2bbox(rois[:, 1:], bbox_pred, means, stds, max_shape=img_shape)
    else:
        bboxes = rois[:, 1:].clone()
        bboxes[:, 0::2] = bboxes[:, 0::2].clamp(0, img_shape[1])
        bboxes[:, 1::2] = bboxes[:, 1::2].clamp(0, img_shape[0])
    if rescale and bboxes.size(0) > 0:
        sf = bboxes.new_tensor(scale_factor)
        bboxes = (bboxes.view(bboxes.size(0), bboxes.shape[1] // 4, 4) / sf).view(bboxes.size(0), -1)
    return multiclass_nms(bboxes, scores, score_thr, iou_threshold, max_per_img)


def paste_masks(masks, boxes, img_h, img_w):
    """masks [n,1,h,w] probabilities, boxes [n,4] -> [n, img_h, img_w]: bilinear samples of every mask on the pixel
    centres of the image, zero outside its box."""
    n = masks.shape[0]
    if n == 0:
        return masks.new_zeros(0, img_h, img_w)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = torch.arange(0, img_h, device=masks.device, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, img_w, device=masks.device, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(n, img_h, img_w)
    gy = img_y[:, :, None].expand(n, img_h, img_w)
    return F.grid_sample(masks.float(), torch.stack([gx, gy], dim=3), align_corners=False)[:, 0]


def get_seg_masks(mask_pred, det_bboxes, det_labels, num_classes, ori_shape, scale_factor, rescale, mask_thr_binary=0.5,
                  class_agnostic=False):
    """mask_pred [n,K,h,w] logits -> list over classes of lists of bool numpy masks [H,W]."""
    cls_segms = [[] for _ in range(num_classes)]
    if mask_pred.shape[0] == 0:
        return cls_segms
    probs = mask_pred.sigmoid()
    boxes = det_bboxes[:, :4]
    if rescale:
        img_h, img_w = ori_shape[:2]
        boxes = boxes / boxes.new_tensor(scale_factor)
    else:
        sf = np.asarray(scale_factor, dtype=np.float64).reshape(-1)
        w_scale, h_scale = (sf[0], sf[1]) if sf.size > 1 else (sf[0], sf[0])
        img_h, img_w = int(np.round(ori_shape[0] * h_scale)), int(np.round(ori_shape[1] * w_scale))
    idx = torch.arange(probs.shape[0], device=probs.device)
    sel = probs[idx, torch.zeros_like(det_labels) if class_agnostic else det_labels][:, None]
    full = paste_masks(sel, boxes, img_h, img_w)
    full = (full >= mask_thr_binary) if mask_thr_binary >= 0 else (full * 255).to(torch.uint8)
    full = full.cpu().numpy()
    for i, lab in enumerate(det_labels.tolist()):
        cls_segms[lab].append(full[i])
    return cls_segms


def bbox2result(bboxes, labels, num_classes):
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    b, l = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [b[l == i, :] for i in range(num_classes)]
