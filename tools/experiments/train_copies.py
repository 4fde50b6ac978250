"""Which aten::copy_ / _to_copy calls a training step makes, grouped by input shapes and dtypes (torch.profiler with
record_shapes; the profiler gives no Python frames on this build, the shapes identify the sites):
    python tools/experiments/train_copies.py            (GPU box)"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::cat",
                 "aten::mul", "aten::add_", "aten::add"):
        dev_us = getattr(e, "device_time_total", None)
        if dev_us is None:
            dev_us = e.cuda_time_total
        rows.append((dev_us / 2, e.count / 2, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for us, n, k, shp in rows[:45]:
    print(f"{us:9.1f} us/step {n:6.1f}x  {k:18s} {shp}")
