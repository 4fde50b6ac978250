"""Golden vectors for the box / mask head training losses: executes the reference's own
BBoxHead.get_targets / _get_target_single (mmdet/models/roi_heads/bbox_heads/bbox_head.py), MAEBoxHeadRec.loss
(mae_bbox_head_rec.py:170-221) with its CrossEntropyLoss / GIoULoss / L1Loss modules and DeltaXYWHBBoxCoder in the
shipped configuration (reg_decoded_bbox + GIoU x 10) and in the encoded-delta L1 configuration, and
MAEMaskHeadPointSup.loss (mae_mask_head_pointSup.py:253-273) -- extracted with ast, decorators dropped -- on seeded
inputs; stores inputs + outputs in tests/golden/head_losses.npz.  Container-only (needs /root/reference)."""
import ast
import functools
import os
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/mmdet"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grab(ns, path, names, cls=None, keep_decorators=()):
    tree = ast.parse(open(path).read())
    nodes = tree.body
    if cls is not None:
        nodes = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    for n in nodes:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names:
            n.decorator_list = [d for d in n.decorator_list if getattr(d, "id", None) in keep_decorators]
            exec(compile(ast.Module(body=[n], type_ignores=[]), path, "exec"), ns)


def main():
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "functools": functools, "partial": functools.partial, "map": map, "zip": zip,
          "BaseBBoxCoder": object, "warnings": __import__("warnings"), "math": __import__("math")}
    grab(ns, REF + "/core/bbox/coder/delta_xywh_bbox_coder.py", ("bbox2delta", "delta2bbox", "DeltaXYWHBBoxCoder"))
    grab(ns, REF + "/core/bbox/iou_calculators/iou2d_calculator.py", ("bbox_overlaps",))
    grab(ns, REF + "/models/losses/utils.py", ("reduce_loss", "weight_reduce_loss", "weighted_loss"))
    grab(ns, REF + "/models/losses/cross_entropy_loss.py", ("cross_entropy", "_expand_onehot_labels", "binary_cross_entropy",
                                                            "mask_cross_entropy", "CrossEntropyLoss"))
    grab(ns, REF + "/models/losses/iou_loss.py", ("giou_loss", "GIoULoss"), keep_decorators=("weighted_loss",))
    grab(ns, REF + "/models/losses/smooth_l1_loss.py", ("l1_loss", "L1Loss"), keep_decorators=("weighted_loss",))
    grab(ns, REF + "/models/losses/accuracy.py", ("accuracy",))
    grab(ns, REF + "/core/utils/misc.py", ("multi_apply",))
    grab(ns, REF + "/models/roi_heads/bbox_heads/bbox_head.py", ("_get_target_single", "get_targets"), cls="BBoxHead")
    get_single, get_targets = ns["_get_target_single"], ns["get_targets"]
    grab(ns, REF + "/models/roi_heads/bbox_heads/mae_bbox_head_rec.py", ("loss",), cls="MAEBoxHeadRec")
    box_loss = ns["loss"]
    grab(ns, REF + "/models/roi_heads/mask_heads/mae_mask_head_pointSup.py", ("loss",), cls="MAEMaskHeadPointSup")
    mask_loss = ns["loss"]
    gen = torch.Generator().manual_seed(4242)
    K = 6
    st = {"K": np.array(K)}

    def boxes(n, size=300.0):
        xy = torch.rand(n, 2, generator=gen) * size
        return torch.cat((xy, xy + 8 + torch.rand(n, 2, generator=gen) * 100), 1)

    results, counts = [], [(5, 9), (0, 6), (3, 0)]
    for i, (npos, nneg) in enumerate(counts):
        gtb = boxes(npos)
        pos = gtb + (torch.rand(npos, 4, generator=gen) - 0.5) * 12
        res = types.SimpleNamespace(pos_bboxes=pos, neg_bboxes=boxes(nneg), pos_gt_bboxes=gtb,
                                    pos_gt_labels=torch.randint(0, K, (npos,), generator=gen))
        res.bboxes = torch.cat((res.pos_bboxes, res.neg_bboxes))
        results.append(res)
        for k in ("pos_bboxes", "neg_bboxes", "pos_gt_bboxes", "pos_gt_labels"):
            st[f"{k}{i}"] = getattr(res, k).numpy()
    st["n_img"] = np.array(len(counts))
    rois = torch.cat([torch.cat((torch.full((r.bboxes.shape[0], 1), float(i)), r.bboxes), 1) for i, r in enumerate(results)])
    n = rois.shape[0]
    cls_score = torch.randn(n, K + 1, generator=gen) * 2
    bbox_pred = torch.randn(n, 4 * K, generator=gen) * 0.3
    st["rois"], st["cls_score"], st["bbox_pred"] = rois.numpy(), cls_score.numpy(), bbox_pred.numpy()
    coder = ns["DeltaXYWHBBoxCoder"](target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])
    for tag, decoded, loss_bbox in (("giou", True, ns["GIoULoss"](loss_weight=10.0)), ("l1", False, ns["L1Loss"](loss_weight=1.0))):
        head = types.SimpleNamespace(num_classes=K, reg_class_agnostic=False, reg_decoded_bbox=decoded, bbox_coder=coder,
                                     loss_cls=ns["CrossEntropyLoss"](use_sigmoid=False, loss_weight=1.0), loss_bbox=loss_bbox,
                                     loss_weight_bbox_start=1.0)
        head._get_target_single = types.MethodType(get_single, head)
        targets = get_targets(head, results, None, None, types.SimpleNamespace(pos_weight=-1), True)
        for name, tval in zip(("labels", "label_weights", "bbox_targets", "bbox_weights"), targets):
            st[f"{tag}_{name}"] = tval.numpy()
        out = box_loss(head, cls_score, bbox_pred, rois, *targets)
        for k, v in out.items():
            st[f"{tag}_{k}"] = np.asarray(v.detach().numpy(), dtype=np.float32).reshape(-1)
        print(tag, {k: float(np.asarray(v.detach()).reshape(-1)[0]) for k, v in out.items()})
    # mask loss on sampled point logits
    R, P = 7, 12
    mask_pred = torch.randn(R, K, P, generator=gen) * 2
    tgt_bool = torch.rand(R, P, generator=gen) > 0.5                                    # the reference's literal bool targets
    tgt_long = tgt_bool.long()
    tgt_long[torch.rand(R, P, generator=gen) > 0.8] = 2                                 # and the evident intent: 2 = ignore
    labels = torch.randint(0, K, (R,), generator=gen)
    mh = types.SimpleNamespace(class_agnostic=False, loss_weight_mask_start=1.0)
    st["mask_pred"], st["mask_tgt_bool"], st["mask_tgt_long"], st["mask_labels"] = mask_pred.numpy(), tgt_bool.numpy(), tgt_long.numpy(), labels.numpy()
    st["mask_loss_bool"] = mask_loss(mh, mask_pred, tgt_bool, labels)["loss_mask"].numpy().reshape(-1)
    st["mask_loss_long"] = mask_loss(mh, mask_pred, tgt_long, labels)["loss_mask"].numpy().reshape(-1)
    st["mask_loss_empty"] = mask_loss(mh, mask_pred[:0], tgt_long[:0], labels[:0])["loss_mask"].numpy().reshape(-1)
    print("mask", st["mask_loss_bool"], st["mask_loss_long"], st["mask_loss_empty"])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "head_losses.npz"), **st)


if __name__ == "__main__":
    main()
