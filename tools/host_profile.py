"""cProfile of bench steps with blocking launches (HIP_LAUNCH_BLOCKING=1) so GPU time lands on the op that
launched it.  Diagnostic: python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("HIP_LAUNCH_BLOCKING", "1")
os.environ.setdefault("AMD_SERIALIZE_KERNEL", "3")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.cuda.set_device(0)
step = bench.build(torch.device("cuda", 0))
with torch.no_grad():
    step(); step()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
