"""Proposal -> pseudo-GT assignment and sampling for the R-CNN branches (host-side tensor logic; mmdet's
MaxIoUAssigner and RandomSampler as configured at configs/mae/attnshift_voc12aug.py:160-175: pos / neg / min-pos IoU
0.5, no low-quality matches, 512 samples per image, at most a quarter positive, GT boxes added to the proposals).
bbox_overlaps and the assignment rule are pinned to the reference's own functions by tests/golden/mmdet_pure.npz (tools/
gen_golden_mmdet_pure.py executes them); the sampler reproduces the reference's RandomSampler draw for draw under the same
torch seed (tests/golden/sampler.npz, tools/gen_golden_sampler.py)."""
from types import SimpleNamespace

import torch


def bbox_overlaps(a, b, eps=1e-6):
    """IoU matrix [len(a), len(b)] of xyxy boxes."""
    if a.numel() == 0 or b.numel() == 0:
        return a.new_zeros(a.shape[0], b.shape[0])
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter).clamp(min=eps)


def max_iou_assign(proposals, gt_bboxes, pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False):
    """-> assigned gt index + 1 per proposal (0 = negative, -1 = ignored), max IoU per proposal."""
    n = proposals.shape[0]
    assigned = proposals.new_full((n,), -1, dtype=torch.long)
    if gt_bboxes.shape[0] == 0 or n == 0:
        assigned[:] = 0
        return assigned, proposals.new_zeros(n)
    iou = bbox_overlaps(gt_bboxes, proposals[:, :4])                      # [G, n]
    max_iou, arg = iou.max(dim=0)
    # (selects instead of masked assignments: a boolean-mask gather reads its count back to the host)
    assigned = torch.where((max_iou >= 0) & (max_iou < neg_iou_thr), torch.zeros_like(assigned), assigned)
    assigned = torch.where(max_iou >= pos_iou_thr, arg + 1, assigned)
    if match_low_quality:
        gt_max, _ = iou.max(dim=1)
        for g in range(gt_bboxes.shape[0]):
            hit = (iou[g] == gt_max[g]) & (gt_max[g] >= min_pos_iou)
            assigned = torch.where(hit, torch.full_like(assigned, g + 1), assigned)
    return assigned, max_iou


def random_sample(proposals, gt_bboxes, gt_labels, assigned, num=512, pos_fraction=0.25, add_gt_as_proposals=True,
                  generator=None, assigned_host=None):
    """mmdet RandomSampler.sample: returns a namespace with pos_inds / neg_inds (into the [gt; proposals] list when the
    GT boxes are added), pos_bboxes, neg_bboxes, pos_assigned_gt_inds, pos_gt_bboxes, pos_gt_labels, bboxes.

    The draw needs the candidate COUNTS on the host (the reference permutes `randperm(count)` from the CPU generator), so
    the selection runs on a host copy of `assigned` (`assigned_host`, a CPU tensor of the proposals' assignment: read back
    here -- the one host sync -- when the caller has not done it) and only index lists go back to the device.  The
    namespace also carries the host copies (`*_host`) for the target builders that follow."""
    from .roi_head import to_device
    dev = proposals.device
    boxes = proposals[:, :4]
    a = (assigned if assigned_host is None else assigned_host).detach().cpu().long()
    G = gt_bboxes.shape[0]
    if add_gt_as_proposals and G:
        boxes = torch.cat((gt_bboxes, boxes), dim=0)
        a = torch.cat((torch.arange(1, G + 1), a))

    def pick(cand, k):
        if cand.numel() <= k:
            return cand
        perm = torch.randperm(cand.numel(), generator=generator)[:k]
        return cand[perm]

    pos_h = pick(torch.nonzero(a > 0, as_tuple=False).flatten(), int(num * pos_fraction)).unique()
    neg_h = pick(torch.nonzero(a == 0, as_tuple=False).flatten(), num - pos_h.numel()).unique()
    gt_h = a[pos_h] - 1
    pos_inds, neg_inds, gt_inds = to_device(pos_h, dev), to_device(neg_h, dev), to_device(gt_h, dev)
    return SimpleNamespace(pos_inds=pos_inds, neg_inds=neg_inds, pos_bboxes=boxes[pos_inds], neg_bboxes=boxes[neg_inds],
                           pos_assigned_gt_inds=gt_inds,
                           pos_gt_bboxes=gt_bboxes[gt_inds] if G else gt_bboxes.new_zeros(0, 4),
                           pos_gt_labels=gt_labels[gt_inds] if gt_labels.shape[0] else gt_labels.new_zeros(0),
                           bboxes=torch.cat((boxes[pos_inds], boxes[neg_inds])),
                           pos_inds_host=pos_h, neg_inds_host=neg_h, pos_assigned_gt_inds_host=gt_h)
