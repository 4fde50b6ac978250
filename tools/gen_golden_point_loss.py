"""Golden vectors for the point-token loss: executes the reference RoI head's own `get_targets`,
`_get_target_single` and `loss` (stdroi:3284-3514), its HungarianPointAssigner, FocalLoss / L1Loss modules (python focal
path), `accuracy` and `multi_apply` -- all extracted with ast, decorators dropped, because `import mmdet` needs mmcv --
on seeded two-image batches and stores inputs + the three loss values in tests/golden/point_loss.npz.
Container-only (needs /root/reference)."""
import ast
import functools
import os
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

REF = "/root/reference/mmdet"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STDROI = REF + "/models/roi_heads/stdroi_point_deform_attn_reppoints.py"


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
          "linear_sum_assignment": linear_sum_assignment, "BaseAssigner": object,
          "AssignResult": lambda num_gts, gt_inds, max_overlaps, labels=None: types.SimpleNamespace(gt_inds=gt_inds, labels=labels),
          "reduce_mean": lambda t: t, "sigmoid_focal_loss": None}
    grab(ns, REF + "/core/bbox/match_costs/match_cost.py", ("FocalLossCost", "PointL1Cost"))
    grab(ns, REF + "/core/bbox/assigners/hungarian_point_assigner.py", ("HungarianPointAssigner",))
    grab(ns, REF + "/models/losses/utils.py", ("reduce_loss", "weight_reduce_loss", "weighted_loss"))
    grab(ns, REF + "/models/losses/focal_loss.py", ("py_sigmoid_focal_loss", "FocalLoss"))
    grab(ns, REF + "/models/losses/smooth_l1_loss.py", ("l1_loss", "L1Loss"), keep_decorators=("weighted_loss",))
    grab(ns, REF + "/models/losses/accuracy.py", ("accuracy",))
    grab(ns, REF + "/core/utils/misc.py", ("multi_apply",))
    grab(ns, STDROI, ("_get_target_single", "get_targets", "loss"), cls="StandardRoIHeadMaskPointSampleDeformAttnReppoints")
    ns["build_match_cost"] = lambda cfg: ns[cfg["type"]](**{k: v for k, v in cfg.items() if k != "type"})
    asg = ns["HungarianPointAssigner"](cls_cost=dict(type="FocalLossCost", weight=1.0), reg_cost=dict(type="PointL1Cost", weight=10.0))
    K = 20
    head = types.SimpleNamespace(bbox_head=types.SimpleNamespace(
        num_classes=K, loss_point_cls=ns["FocalLoss"](use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
        loss_point=ns["L1Loss"](loss_weight=10.0)))
    head._get_target_single = types.MethodType(ns["_get_target_single"], head)
    cfg = types.SimpleNamespace(point_pos_weight=1)
    gen = torch.Generator().manual_seed(777)
    st = {}
    cases = [((1024, 1024, 3), (800, 1216, 3), (3, 5)), ((224, 224, 3), (224, 320, 3), (2, 0)), ((64, 64, 3), (64, 64, 3), (0, 0))]
    for c, (s0, s1, counts) in enumerate(cases):
        T = 100
        shapes = [s0, s1]
        point_cls = torch.randn(2, T, K, generator=gen) * 2
        point_reg = torch.rand(2, T, 2, generator=gen)
        gt_points = [torch.rand(n, 2, generator=gen) * torch.tensor([s[1], s[0]], dtype=torch.float32) for n, s in zip(counts, shapes)]
        gt_labels = [torch.randint(0, K, (n,), generator=gen) for n in counts]
        results = []
        for i in range(2):
            ar = asg.assign(point_reg[i], point_cls[i], gt_points[i], gt_labels[i], dict(img_shape=shapes[i]))
            pos = torch.nonzero(ar.gt_inds > 0, as_tuple=False).squeeze(-1).unique()         # PointPseudoSampler.sample
            neg = torch.nonzero(ar.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
            gi = ar.gt_inds[pos] - 1
            results.append(types.SimpleNamespace(pos_inds=pos, neg_inds=neg, pos_bboxes=point_reg[i][pos], neg_bboxes=point_reg[i][neg],
                                                 pos_gt_bboxes=gt_points[i][gi].view(-1, 2), pos_gt_labels=ar.labels[pos]))
        targets = ns["get_targets"](head, results, None, None, cfg, True)
        whwh = torch.cat([torch.tensor([[s[1], s[0]]], dtype=torch.float32) for s in shapes])[:, None, :].repeat(1, T, 1)
        out = ns["loss"](head, point_cls.view(-1, K), point_reg.view(-1, 2), *targets, imgs_whwh=whwh)
        st.update({f"cls{c}": point_cls.numpy(), f"reg{c}": point_reg.numpy(), f"shapes{c}": np.array(shapes)})
        for i in range(2):
            st[f"pts{c}_{i}"], st[f"labels{c}_{i}"] = gt_points[i].numpy(), gt_labels[i].numpy()
        st[f"labels_all{c}"], st[f"label_w{c}"] = targets[0].numpy(), targets[1].numpy().astype(np.float32)
        for k, v in out.items():
            st[f"{k}{c}"] = np.asarray(v.detach().numpy(), dtype=np.float32).reshape(-1)
        print(c, {k: float(np.asarray(v.detach()).reshape(-1)[0]) if v.numel() else None for k, v in out.items()})
    st["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "point_loss.npz"), **st)


if __name__ == "__main__":
    main()
