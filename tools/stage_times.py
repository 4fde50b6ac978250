"""Where a bench step spends its time: wall-clock per stage with a device sync between stages.
(diagnostic only; python tools/stage_times.py [--reps 3])"""
import argparse
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AS_STAGE_LOG"] = "1"
import bench  # noqa: E402
import attentionshift_amd as A  # noqa: E402
from attentionshift_amd import ops, roi_head as RH  # noqa: E402

T = defaultdict(float)


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def inner(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        T[label] += time.perf_counter() - t0
        return r

    setattr(obj, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    torch.set_num_threads(int(os.environ.get("AS_HOST_THREADS", "8")))
    for n in ("attention_fwd", "linear", "rollout_rows", "cam_boxes", "refine_similarity", "instance_maps", "cosine_shift",
              "crop_threshold_erode"):
        wrap(ops, n, "op:" + n)
    for n in ("sample_point_grid", "seed_features", "mask_sample_points", "grid_seed_coords", "filter_parts", "merge_parts",
              "part_similarity", "part_centers", "rank_select", "_down16", "_minmax_maps"):
        wrap(RH, n, "host:" + n)
    step = bench.build(torch.device("cuda", 0))
    with torch.no_grad():
        step(); step()
        T.clear()
        RH.CLOCK.acc.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            step()
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    print(f"total per step {total / a.reps * 1e3:.2f} ms (with stage syncs)")
    acc = 0.0
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v / a.reps * 1e3:9.3f} ms")
        acc += v
    print(f"  {'(unattributed)':28s} {(total - acc) / a.reps * 1e3:9.3f} ms")
    print("seed_pseudo_gt stage clock (ms/step):")
    for k, v in RH.CLOCK.report(a.reps).items():
        print(f"  {k:28s} {v:9.3f}")


if __name__ == "__main__":
    main()
