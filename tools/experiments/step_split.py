"""Wall-clock split of the headline step: backbone forward alone, attention shift alone (on a fixed backbone output),
and the whole step.    python tools/experiments/step_split.py   (GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "vitb"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    parts = {}
    import attentionshift_amd as A
    real_build = A.build_backbone

    def spy(cfg):
        parts["bb"] = real_build(cfg)
        return parts["bb"]

    A.build_backbone = spy
    step = bench.build(dev, os.environ.get("AS_RNG_MODE", "fast"))
    bb = parts["bb"]
    fn = step.__closure__
    cells = {n: c.cell_contents for n, c in zip(step.__code__.co_freevars, fn)}
    pseudo, img = cells["pseudo_labels"], cells["img"]
    with torch.no_grad():
        out = bb(img)
        print(f"backbone forward      : {timed(lambda: bb(img)):7.3f} ms")
        print(f"attention shift (head): {timed(lambda: pseudo(out)):7.3f} ms")
        print(f"whole step            : {timed(step):7.3f} ms")


if __name__ == "__main__":
    main()
