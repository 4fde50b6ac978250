"""Which Python lines launch the small ATen kernels of a headline bench step (GPU box):
    python tools/experiments/glue_sites.py [--steps 3]
A TorchDispatchMode logs every ATen op that touches a device tensor together with the innermost frame inside this repo
(the profiler's with_stack gives no Python frames on this build); per (op, site): calls per step, sorted by count.  Device
time per op NAME comes from the profiler table printed underneath."""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_leaves

import bench

NO_KERNEL = {"aten.view.default", "aten.reshape.default", "aten._unsafe_view.default", "aten.permute.default", "aten.transpose.int",
             "aten.select.int", "aten.slice.Tensor", "aten.unsqueeze.default", "aten.squeeze.dim", "aten.expand.default",
             "aten.as_strided.default", "aten.detach.default", "aten.alias.default", "aten.t.default", "aten.unbind.int",
             "aten.split.Tensor", "aten.split_with_sizes.default", "aten.squeeze.default", "aten.empty.memory_format",
             "aten.empty_like.default", "aten.empty_strided.default", "aten.new_empty.default", "aten.unflatten.int",
             "aten.flatten.using_ints", "aten.narrow.default", "aten.view_as.default", "aten.chunk.default",
             "aten.lift_fresh.default", "aten.is_pinned.default", "aten.resize_.default", "aten.new_empty_strided.default",
             "aten.sym_size.int", "aten.stride.int", "aten.numel.default", "aten.size.int", "aten.view.dtype",
             "aten.record_stream.default", "aten.as_strided.default"}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name not in NO_KERNEL and any(isinstance(t, torch.Tensor) and t.is_cuda for t in tree_leaves((args, kwargs, out))):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=24)):
                if ("/attentionshift_amd/" in fr.filename or fr.filename.endswith("/bench.py")) and "glue_sites" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} ({fr.name})"
                    break
            self.n[(name, site)] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=120)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    step = bench.build(dev, "fast")
    with torch.no_grad():
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        mode = Sites()
        with mode:
            for _ in range(a.steps):
                step()
        torch.cuda.synchronize()
        rows = sorted(mode.n.items(), key=lambda kv: (kv[0][1], -kv[1]))
        print(f"ATen ops on device tensors per step: {sum(mode.n.values()) / a.steps:.0f}")
        by_site = collections.Counter()
        for (op, site), n in rows:
            by_site[site] += n
        for (op, site), n in sorted(rows, key=lambda kv: (-by_site[kv[0][1]], kv[0][1], -kv[1]))[:a.rows]:
            print(f"{n / a.steps:6.1f}x  {op:34s} {site}")
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
    tot, cnt = 0.0, 0
    print("\nper op name (profiler): device us per step, calls per step")
    for ev in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total):
        if ev.key.startswith("aten::") and ev.self_device_time_total > 0:
            tot += ev.self_device_time_total
            cnt += ev.count
            print(f"{ev.self_device_time_total / a.steps:8.1f} us  {ev.count / a.steps:5.1f}x  {ev.key}")
    print(f"ATen device time per step: {tot / a.steps:.1f} us in {cnt / a.steps:.0f} ops")


if __name__ == "__main__":
    main()
