"""N DDP training steps of bench.py's training leg on one GPU (for rocprofv3 / timing):
    python tools/experiments/train_steps.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from attentionshift_amd.dist import Ranks  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.cuda.set_device(0)
step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for _ in range(n):
    h0 = time.perf_counter()
    step()
    host += time.perf_counter() - h0
torch.cuda.synchronize()
print("train step %.2f ms (host returns from step() after %.2f ms on average: host-bound if the two are close)"
      % ((time.perf_counter() - t0) / n * 1e3, host / n * 1e3))
