"""cProfile of bench steps WITHOUT blocking launches: attributes host (Python / dispatch) time, which is what bounds the
RoI-head phase.  Sequential images so that the per-image chain shows up in the main thread.  Diagnostic."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
torch.set_num_threads(8)
step = bench.build(torch.device("cuda", 0), "fast")
step.head.parallel_images = False
with torch.no_grad():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(60)
