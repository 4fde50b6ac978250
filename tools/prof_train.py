"""Host-side profile of the DDP training step of bench.py (GPU box): cProfile over two steps with a device sync at the
end, top cumulative entries.    python tools/prof_train.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from attentionshift_amd.dist import Ranks  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
step = bench.build(dev, "fast", train=True, ranks=Ranks())
for _ in range(2):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
