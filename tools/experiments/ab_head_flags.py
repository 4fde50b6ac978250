"""A/B of the RoI-head scheduling flags on one box: full-step time with each flag toggled (interleaved repeats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

def loop(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

torch.cuda.set_device(0); torch.set_num_threads(8)
step = bench.build(torch.device("cuda", 0), "fast")
h = step.head
configs = {"stream per image": dict(image_streams=True), "one stream": dict(image_streams=False),
           "thread + stream per image": dict(image_streams=False, parallel_images=True)}
with torch.no_grad():
    for rep in range(3):
        for name, cfg in configs.items():
            h.parallel_images = False
            for k, v in cfg.items():
                setattr(h, k, v)
            print(f"{name:24s} {loop(step):.2f} ms", flush=True)
