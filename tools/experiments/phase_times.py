"""Wall time of the two phases of a bench step, each measured as a free-running loop with ONE sync at the end
(no per-stage syncs): backbone only, RoI head only (on a frozen backbone output), both.  Diagnostic."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def loop(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    torch.cuda.set_device(0)
    torch.set_num_threads(8)
    step = bench.build(torch.device("cuda", 0), os.environ.get("AS_RNG_MODE", "fast"))
    cells = {c.cell_contents.__class__.__name__: c.cell_contents for c in step.__closure__ if hasattr(c.cell_contents, "__class__")}
    bb = next(v for v in (c.cell_contents for c in step.__closure__) if isinstance(v, torch.nn.Module) and hasattr(v, "blocks"))
    img = next(v for v in (c.cell_contents for c in step.__closure__) if torch.is_tensor(v) and v.dim() == 4 and v.shape[1] == 3)
    pl = next(v for v in (c.cell_contents for c in step.__closure__) if callable(v) and getattr(v, "__name__", "") == "pseudo_labels")
    with torch.no_grad():
        out = bb(img)
        print("backbone only   %.2f ms" % loop(lambda: bb(img)))
        print("roi head only   %.2f ms" % loop(lambda: pl(out)))
        print("full step       %.2f ms" % loop(step))
        step.head.parallel_images = True
        print("roi head only (one thread + stream per image) %.2f ms" % loop(lambda: pl(out)))


if __name__ == "__main__":
    main()
