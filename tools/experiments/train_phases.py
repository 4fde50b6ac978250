"""Where a training step spends host and device time, phase by phase: host clock and a HIP event at the entry and exit of
the backbone forward, the attention shift, the RoI head's losses, backward, the gradient reduction and the optimizer.
A phase whose device interval is as long as its host interval with little queued work is latency-bound (host syncs).
    python tools/experiments/train_phases.py [steps]      (GPU box)"""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import attentionshift_amd as A  # noqa: E402
from attentionshift_amd.dist import GradAllReducer, Ranks  # noqa: E402

marks = []


def mark(tag):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((tag, time.perf_counter(), ev))


def wrap(owner, name, tag):
    fn = getattr(owner, name)

    def inner(*a, **k):
        mark(tag + ":in")
        try:
            return fn(*a, **k)
        finally:
            mark(tag + ":out")

    setattr(owner, name, inner)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.cuda.set_device(0)
step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
wrap(A.VisionTransformerDet, "forward", "backbone_fwd")
wrap(type(step.head).__mro__[1], "seed_pseudo_gt", "attention_shift")
wrap(type(step.head).__mro__[1], "forward_train", "roi_losses")
wrap(torch.Tensor, "backward", "backward")
wrap(GradAllReducer, "finish", "grad_reduce")
wrap(torch.optim.AdamW, "step", "optimizer")
if os.environ.get("PHASES_FINE"):                      # the pieces of the RoI head's loss phase
    from attentionshift_amd import assign, mask_targets, point_loss, mae_heads
    wrap(point_loss, "point_token_loss", " point_token_loss")
    wrap(assign, "max_iou_assign", " max_iou_assign")
    wrap(assign, "random_sample", " random_sample")
    wrap(type(step.head).__mro__[1], "_roi_extract", " roi_extract")
    wrap(mae_heads.MAEBoxHeadRec, "forward", " bbox_head_fwd")
    wrap(mae_heads.MAEBoxHeadRec, "get_targets", " bbox_targets")
    wrap(mae_heads.MAEBoxHeadRec, "loss", " bbox_loss")
    wrap(mae_heads.MAEMaskHeadPointSup, "forward", " mask_head_fwd")
    wrap(mask_targets, "mask_point_targets", " mask_point_targets")
    wrap(mask_targets, "point_sample", " point_sample")
    wrap(mae_heads.MAEMaskHeadPointSup, "loss", " mask_loss")
for _ in range(3):
    step()
torch.cuda.synchronize()
acc = collections.OrderedDict()
total_h = total_d = 0.0
for _ in range(n):
    marks.clear()
    mark("step:in")
    step()
    mark("step:out")
    torch.cuda.synchronize()
    t0, e0 = marks[0][1], marks[0][2]
    total_h += marks[-1][1] - t0
    total_d += e0.elapsed_time(marks[-1][2])
    opened = {}
    for tag, t, ev in marks:
        name, kind = tag.rsplit(":", 1)
        if kind == "in":
            opened[name] = (t, ev)
        elif name in opened:                                  # (calls of one name are summed over the step)
            t_in, ev_in = opened.pop(name)
            a = acc.setdefault(name, [0.0] * 5)
            a[0] += (t_in - t0) * 1e3
            a[1] += (t - t_in) * 1e3
            a[2] += e0.elapsed_time(ev_in)
            a[3] += ev_in.elapsed_time(ev)
            a[4] += 1
print(f"step: host {total_h / n * 1e3:.2f} ms, device {total_d / n:.2f} ms")
print(f"{'phase':18s} {'host start':>10s} {'host ms':>8s} {'dev start':>10s} {'dev ms':>8s}   (per step, mean of {n})")
for name, a in acc.items():
    print(f"{name:18s} {a[0] / n:10.2f} {a[1] / n:8.2f} {a[2] / n:10.2f} {a[3] / n:8.2f}")
