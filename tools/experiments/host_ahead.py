"""Is the headline step host-bound or device-bound?  Per step: when the host FIRST blocks on a readback (all launches of
the backbone + RoI chains are queued by then), how long it waits there, and the step's wall time (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from attentionshift_amd import roi_head as RH

LOG = []
_orig = RH._to_host_finish


def finish(pending):
    t0 = time.perf_counter()
    out = _orig(pending)
    LOG.append((t0, time.perf_counter()))
    return out


RH._to_host_finish = finish


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    step = bench.build(dev, "fast")
    with torch.no_grad():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        rows = []
        for _ in range(10):
            LOG.clear()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rows.append((t1 - t0, [(a - t0, b - a) for a, b in LOG]))
    for tot, waits in rows:
        print("step %.3f ms; readback waits (at ms, for ms): %s" % (tot * 1e3, ", ".join("%.3f+%.3f" % (a * 1e3, b * 1e3) for a, b in waits)))


if __name__ == "__main__":
    main()
