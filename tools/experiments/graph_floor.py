"""Per-launch floor of a chain of tiny dependent kernels: eager stream launches vs one hipGraph replay (GPU box)."""
import time
import torch

x = torch.zeros(1024, device="cuda")
N = 200


def chain():
    for _ in range(N):
        x.add_(1.0)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / N * 1e3, (time.perf_counter() - t0) / reps / N * 1e6


print("eager  : %.2f us/kernel device span, %.2f us wall" % timed(chain))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
print("graph  : %.2f us/kernel device span, %.2f us wall" % timed(g.replay))
# a long-running kernel first so that the eager chain is fully queued before the device reaches it (device floor, not host)
big = torch.zeros(64 * 1024 * 1024, device="cuda")


def queued():
    for _ in range(40):
        big.add_(1.0)
    chain()


def only_big():
    for _ in range(40):
        big.add_(1.0)


a, _ = timed(queued, 5)
b, _ = timed(only_big, 5)
print("eager, queued behind 40 big kernels: %.2f us/kernel" % (a - b))
