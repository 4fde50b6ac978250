"""Golden vectors for the pure-torch mmdet helpers the RoI head's training / test paths restate: executes the
reference's own functions (extracted with ast, decorators dropped, because `import mmdet` needs mmcv) on seeded inputs
and stores inputs + outputs in tests/golden/mmdet_pure.npz.  Container-only (needs /root/reference).

    bbox2delta, delta2bbox     mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:88-235
    bbox_overlaps (iou, giou)  mmdet/core/bbox/iou_calculators/iou2d_calculator.py:43-159
    assign_wrt_overlaps        mmdet/core/bbox/assigners/max_iou_assigner.py:127-212 (pos 0.5 / neg 0.5 / min 0.5; and the
                               RPN setting 0.7 / 0.3 / 0.3 with low-quality matches)
    py_sigmoid_focal_loss      mmdet/models/losses/focal_loss.py:11-56 (+ weight_reduce_loss, losses/utils.py:7-54)
    cross_entropy, accuracy    mmdet/models/losses/cross_entropy_loss.py:9-39, accuracy.py:6-50
    _do_paste_mask             mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:306-370
"""
import ast
import os
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/mmdet"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grab(ns, path, names, methods_of=None):
    tree = ast.parse(open(path).read())
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            n.decorator_list = []
            exec(compile(ast.Module(body=[n], type_ignores=[]), path, "exec"), ns)
        if methods_of and isinstance(n, ast.ClassDef) and n.name == methods_of:
            for m in n.body:
                if isinstance(m, ast.FunctionDef) and m.name in names:
                    m.decorator_list = []
                    exec(compile(ast.Module(body=[m], type_ignores=[]), path, "exec"), ns)


def main():
    ns = {"torch": torch, "np": np, "F": F, "nn": torch.nn,
          "AssignResult": lambda num_gts, gt_inds, max_overlaps, labels=None: types.SimpleNamespace(
              gt_inds=gt_inds, max_overlaps=max_overlaps, labels=labels)}
    grab(ns, REF + "/core/bbox/coder/delta_xywh_bbox_coder.py", ("bbox2delta", "delta2bbox"))
    grab(ns, REF + "/core/bbox/iou_calculators/iou2d_calculator.py", ("bbox_overlaps",))
    grab(ns, REF + "/core/bbox/assigners/max_iou_assigner.py", ("assign_wrt_overlaps",), methods_of="MaxIoUAssigner")
    grab(ns, REF + "/models/losses/utils.py", ("reduce_loss", "weight_reduce_loss"))
    grab(ns, REF + "/models/losses/focal_loss.py", ("py_sigmoid_focal_loss",))
    grab(ns, REF + "/models/losses/cross_entropy_loss.py", ("cross_entropy",))
    grab(ns, REF + "/models/losses/accuracy.py", ("accuracy",))
    grab(ns, REF + "/models/roi_heads/mask_heads/fcn_mask_head.py", ("_do_paste_mask",))
    gen = torch.Generator().manual_seed(99)
    st = {}

    def boxes(n, size=300.0):
        xy = torch.rand(n, 2, generator=gen) * size
        return torch.cat((xy, xy + 5 + torch.rand(n, 2, generator=gen) * 120), 1)

    # coder
    props, gts = boxes(40), boxes(40)
    stds = (0.1, 0.1, 0.2, 0.2)
    st["props"], st["gts"] = props.numpy(), gts.numpy()
    deltas = ns["bbox2delta"](props, gts, (0., 0., 0., 0.), stds)
    st["deltas"] = deltas.numpy()
    d4k = torch.randn(40, 12, generator=gen) * 0.5
    st["d4k"] = d4k.numpy()
    st["decoded_clip"] = ns["delta2bbox"](props, d4k, (0., 0., 0., 0.), stds, max_shape=(260, 340, 3)).numpy()
    st["decoded_free"] = ns["delta2bbox"](props, d4k[:, :4] * 4, (0., 0., 0., 0.), stds).numpy()     # hits the ratio clamp
    # overlaps
    a, b = boxes(25), boxes(9)
    st["ov_a"], st["ov_b"] = a.numpy(), b.numpy()
    st["iou"] = ns["bbox_overlaps"](a, b).numpy()
    st["giou_aligned"] = ns["bbox_overlaps"](a[:9], b, mode="giou", is_aligned=True).numpy()
    # assigner, the two settings of the config
    ov = ns["bbox_overlaps"](b, a)                                                   # [gts, proposals]
    for tag, kw in (("rcnn", dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)),
                    ("rpn", dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True))):
        self = types.SimpleNamespace(gt_max_assign_all=True, **kw)
        res = ns["assign_wrt_overlaps"](self, ov.clone())
        st[f"assign_{tag}"] = res.gt_inds.numpy()
    st["assign_nogt"] = ns["assign_wrt_overlaps"](types.SimpleNamespace(gt_max_assign_all=True, pos_iou_thr=0.5, neg_iou_thr=0.5,
                                                                         min_pos_iou=0.5, match_low_quality=False),
                                                   ov[:0].clone()).gt_inds.numpy()
    # focal loss / cross entropy / accuracy
    logits = torch.randn(64, 20, generator=gen) * 2
    target = torch.randint(0, 21, (64,), generator=gen)                             # 20 = background
    weight = torch.rand(64, generator=gen)
    st["fl_logits"], st["fl_target"], st["fl_weight"] = logits.numpy(), target.numpy(), weight.numpy()
    onehot = F.one_hot(target, 21)[:, :20]                                          # the python form takes one-hot targets
    st["fl_out"] = ns["py_sigmoid_focal_loss"](logits, onehot, weight, gamma=2.0, alpha=0.25, reduction="mean",
                                               avg_factor=7.0).numpy()
    ce_logits = torch.randn(64, 21, generator=gen) * 2
    st["ce_logits"] = ce_logits.numpy()
    st["ce_out"] = ns["cross_entropy"](ce_logits, target, weight, reduction="mean", avg_factor=33.0).numpy()
    st["acc_out"] = ns["accuracy"](ce_logits, target).numpy()
    # mask paste
    masks = torch.rand(5, 1, 28, 28, generator=gen)
    pb = boxes(5, 60.0)
    st["pm_masks"], st["pm_boxes"] = masks.numpy(), pb.numpy()
    pasted, _ = ns["_do_paste_mask"](masks, pb, 150, 170, skip_empty=False)
    st["pm_out"] = pasted.numpy()
    path = os.path.join(ROOT, "tests", "golden", "mmdet_pure.npz")
    np.savez_compressed(path, **st)
    print("wrote", path, {k: v.shape for k, v in st.items() if k.endswith("out") or k.startswith("assign")})


if __name__ == "__main__":
    main()
