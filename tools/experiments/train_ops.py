"""torch.profiler op table of the training leg (GPU box): which ATen ops the ~750 small launches per step are.
    python tools/experiments/train_ops.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from attentionshift_amd.dist import Ranks  # noqa: E402

torch.cuda.set_device(0)
step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print("%9s %6s  %-42s %s" % ("us/step", "n/step", "op", "shapes"))
for e in rows[:70]:
    print("%9.1f %6.1f  %-42s %s" % (e.self_device_time_total / N, e.count / N, e.key[:42], str(e.input_shapes)[:110]))
tot = {}
for e in rows:
    t = tot.setdefault(e.key, [0, 0.0])
    t[0] += e.count
    t[1] += e.self_device_time_total
print("\nby op name:")
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%9.1f us %6.1f x  %s" % (us / N, n / N, k))
