"""Which Python lines launch the small ATen kernels of a headline bench step (GPU box):
    python tools/experiments/glue_sites.py [--steps 3]
Per (ATen op, innermost frame inside this repo): launches and device microseconds per step, sorted by device time."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=70)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    step = bench.build(dev, "fast")
    with torch.no_grad():
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.key_averages(group_by_stack_n=12):
        if not ev.key.startswith("aten::") or ev.self_device_time_total <= 0:
            continue
        site = "?"
        for fr in ev.stack:
            if "/attentionshift_amd/" in fr or "/bench.py" in fr:
                site = fr.replace(ROOT + "/", "")
                break
        k = (ev.key, site)
        agg[k][0] += ev.count
        agg[k][1] += ev.self_device_time_total
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"ATen device time per step: {tot / a.steps:.1f} us in {sum(v[0] for _, v in rows) / a.steps:.0f} ops")
    for (op, site), (n, us) in rows[:a.rows]:
        print(f"{us / a.steps:8.1f} us  {n / a.steps:5.1f}x  {op:28s} {site}")


if __name__ == "__main__":
    main()
