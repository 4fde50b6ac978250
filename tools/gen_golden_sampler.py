"""Golden vectors for the R-CNN proposal sampling: executes the reference's own MaxIoUAssigner.assign (without
ignore boxes), AssignResult.add_gt_, BaseSampler.sample and RandomSampler (mmdet/core/bbox/{assigners,samplers}) -- classes
extracted with ast, registry decorators dropped -- with the config's settings (512 samples, a quarter positive, GT boxes
added) under torch.manual_seed, and stores inputs + the sampled index sets in tests/golden/sampler.npz.
Container-only (needs /root/reference)."""
import ast
import os

import numpy as np
import torch

REF = "/root/reference/mmdet/core/bbox"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grab(ns, path, names):
    tree = ast.parse(open(path).read())
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names:
            n.decorator_list = []
            exec(compile(ast.Module(body=[n], type_ignores=[]), path, "exec"), ns)


class NiceRepr:
    pass


def main():
    from abc import ABCMeta, abstractmethod
    ns = {"torch": torch, "np": np, "ABCMeta": ABCMeta, "abstractmethod": abstractmethod,
          "util_mixins": type("m", (), {"NiceRepr": NiceRepr}), "build_iou_calculator": lambda cfg: None}
    grab(ns, REF + "/iou_calculators/iou2d_calculator.py", ("bbox_overlaps",))
    ns["build_iou_calculator"] = lambda cfg: ns["bbox_overlaps"]
    grab(ns, REF + "/assigners/assign_result.py", ("AssignResult",))
    grab(ns, REF + "/assigners/base_assigner.py", ("BaseAssigner",))
    grab(ns, REF + "/assigners/max_iou_assigner.py", ("MaxIoUAssigner",))
    grab(ns, REF + "/samplers/sampling_result.py", ("SamplingResult",))
    grab(ns, REF + "/samplers/base_sampler.py", ("BaseSampler",))
    grab(ns, REF + "/samplers/random_sampler.py", ("RandomSampler",))
    import sys, types                                               # RandomSampler.__init__ imports a helper for its rng argument
    dm = types.ModuleType("mmdet.core.bbox.demodata")
    dm.ensure_rng = lambda rng=None: np.random.mtrand._rand if rng is None else rng
    for name in ("mmdet", "mmdet.core", "mmdet.core.bbox"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mmdet.core.bbox"].demodata = dm
    sys.modules["mmdet.core.bbox.demodata"] = dm
    assigner = ns["MaxIoUAssigner"](pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
    gen = torch.Generator().manual_seed(8)
    st = {}

    def boxes(n, size=500.0):
        xy = torch.rand(n, 2, generator=gen) * size
        return torch.cat((xy, xy + 20 + torch.rand(n, 2, generator=gen) * 150), 1)

    cases = [(4, 900, 512, 0.25), (2, 60, 512, 0.25), (3, 300, 64, 0.25), (0, 50, 32, 0.25)]
    for c, (G, n, num, frac) in enumerate(cases):
        gts, labels = boxes(G), torch.randint(0, 20, (G,), generator=gen)
        props = boxes(n)
        if G:                                                       # make a good share of the proposals positives
            k = n // 2
            props[:k] = gts[torch.randint(0, G, (k,), generator=gen)] + (torch.rand(k, 4, generator=gen) - 0.5) * 30
        sampler = ns["RandomSampler"](num=num, pos_fraction=frac, neg_pos_ub=-1, add_gt_as_proposals=True)
        res_a = assigner.assign(props, gts, None, labels)
        torch.manual_seed(1000 + c)
        res = sampler.sample(res_a, props, gts, labels)
        st.update({f"props{c}": props.numpy(), f"gts{c}": gts.numpy(), f"labels{c}": labels.numpy(), f"num{c}": np.array(num),
                   f"pos_inds{c}": res.pos_inds.numpy(), f"neg_inds{c}": res.neg_inds.numpy(),
                   f"pos_assigned{c}": res.pos_assigned_gt_inds.numpy(), f"pos_gt_labels{c}": res.pos_gt_labels.numpy(),
                   f"seed{c}": np.array(1000 + c)})
        print(c, "pos", res.pos_inds.numel(), "neg", res.neg_inds.numel())
    st["n"] = np.array(len(cases))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sampler.npz"), **st)


if __name__ == "__main__":
    main()
