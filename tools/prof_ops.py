"""torch.profiler view of one headline bench step: ATen / HIP-extension work grouped by op and input shapes, so the glue
kernels rocprof shows can be traced to their call sites.    python tools/prof_ops.py [--config vitb]   (GPU box)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vitb")
    ap.add_argument("--rows", type=int, default=60)
    a = ap.parse_args()
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[a.config])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    step = bench.build(dev, "fast")
    with torch.no_grad():
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=a.rows,
                                                             max_name_column_width=44, max_shapes_column_width=70))
    print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=40,
                                                      max_src_column_width=90))


if __name__ == "__main__":
    main()
