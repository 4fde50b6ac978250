"""Golden vectors for the point-token Hungarian matching (consumer of outputs_class / outputs_coord at the head of
seed_pseudo_gt and of the point loss): executes the reference's own HungarianPointAssigner.assign with its FocalLossCost
and PointL1Cost (mmdet/core/bbox/assigners/hungarian_point_assigner.py:53-107, match_costs/match_cost.py:52-107;
extracted with ast because importing mmdet needs mmcv) on seeded cases and stores inputs + assigned_gt_inds in
tests/golden/hungarian.npz.  Container-only (needs /root/reference)."""
import ast
import os
import types

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

REF = "/root/reference/mmdet/core/bbox"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    ns = {"torch": torch, "linear_sum_assignment": linear_sum_assignment, "BaseAssigner": object,
          "AssignResult": lambda num_gts, gt_inds, max_overlaps, labels=None: types.SimpleNamespace(
              num_gts=num_gts, gt_inds=gt_inds, labels=labels),
          "build_match_cost": None}
    for path, want in ((REF + "/match_costs/match_cost.py", ("FocalLossCost", "PointL1Cost")),
                       (REF + "/assigners/hungarian_point_assigner.py", ("HungarianPointAssigner",))):
        tree = ast.parse(open(path).read())
        for n in tree.body:
            if isinstance(n, ast.ClassDef) and n.name in want:
                n.decorator_list = []
                exec(compile(ast.Module(body=[n], type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    ns = load()
    ns["build_match_cost"] = lambda cfg: ns[cfg["type"]](**{k: v for k, v in cfg.items() if k != "type"})
    asg = ns["HungarianPointAssigner"](cls_cost=dict(type="FocalLossCost", weight=1.0),
                                       reg_cost=dict(type="PointL1Cost", weight=10.0), times=1)
    gen = torch.Generator().manual_seed(2024)
    store, cases = {}, [(100, 20, 3, (1024, 1024, 3)), (100, 20, 7, (800, 1216, 3)), (10, 5, 10, (224, 224, 3)),
                        (6, 4, 9, (224, 320, 3)), (100, 20, 1, (512, 512, 3)), (8, 3, 0, (64, 64, 3))]
    for i, (T, K, G, shape) in enumerate(cases):
        pred = torch.rand(T, 2, generator=gen)
        cls = torch.randn(T, K, generator=gen) * 2
        pts = torch.rand(G, 2, generator=gen) * torch.tensor([shape[1], shape[0]], dtype=torch.float32)
        labels = torch.randint(0, K, (G,), generator=gen)
        res = asg.assign(pred, cls, pts, labels, dict(img_shape=shape))
        store.update({f"pred{i}": pred.numpy(), f"cls{i}": cls.numpy(), f"pts{i}": pts.numpy(), f"labels{i}": labels.numpy(),
                      f"shape{i}": np.array(shape), f"gt_inds{i}": res.gt_inds.numpy()})
    store["n"] = np.array(len(cases))
    path = os.path.join(ROOT, "tests", "golden", "hungarian.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, [int((store[f"gt_inds{i}"] > 0).sum()) for i in range(len(cases))])


if __name__ == "__main__":
    main()
