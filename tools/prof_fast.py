"""Profiling target: the headline step in fast-RNG mode only (bench.py also times the reference-RNG mode, which mixes
two kernel populations in one rocprofv3 trace).  usage: tools/prof_cmd.sh <tag> python tools/prof_fast.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
torch.set_num_threads(8)
step = bench.build(torch.device("cuda", 0), "fast")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with torch.no_grad():
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
