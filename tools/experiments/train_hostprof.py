"""cProfile of the HOST side of one training step (after warm-up): where the interpreter spends the time it needs to queue
the step.    python tools/experiments/train_hostprof.py [n_lines]     (GPU box)"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from attentionshift_amd.dist import Ranks  # noqa: E402

torch.cuda.set_device(0)
step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    step()
pr.disable()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 45
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(n)
st.sort_stats("cumulative").print_stats(r"attentionshift_amd|bench", n)
