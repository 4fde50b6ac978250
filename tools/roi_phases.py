"""Wall-clock of the RoI head's three natural segments (no extra syncs): [rollout + cam_boxes ... status check],
[per-image chains], [tail].  Diagnostic."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from attentionshift_amd import roi_head as RH

T = {"a": 0.0, "b": 0.0, "n": 0}
orig_run = RH.AttnShiftRoIHead._run_images
def timed_run(self, fn, n):
    t0 = time.perf_counter()
    T["a"] += t0 - T["t_start"]
    r = orig_run(self, fn, n)
    T["b"] += time.perf_counter() - t0
    return r
RH.AttnShiftRoIHead._run_images = timed_run

torch.cuda.set_device(0); torch.set_num_threads(8)
step = bench.build(torch.device("cuda", 0), "fast")
bb = next(v for v in (c.cell_contents for c in step.__closure__) if isinstance(v, torch.nn.Module) and hasattr(v, "blocks"))
img = next(v for v in (c.cell_contents for c in step.__closure__) if torch.is_tensor(v) and v.dim() == 4 and v.shape[1] == 3)
pl = next(v for v in (c.cell_contents for c in step.__closure__) if callable(v) and getattr(v, "__name__", "") == "pseudo_labels")
with torch.no_grad():
    out = bb(img)
    for par in (True, False):
        step.head.parallel_images = par
        for _ in range(3):
            T["t_start"] = time.perf_counter(); pl(out)
        torch.cuda.synchronize()
        T.update(a=0.0, b=0.0)
        t0 = time.perf_counter()
        for _ in range(10):
            T["t_start"] = time.perf_counter(); pl(out)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / 10 * 1e3
        print(f"parallel={par}: total {tot:.2f} ms  pre-chain (rollout, cam_boxes, select) {T['a']/10*1e3:.2f}  per-image chains {T['b']/10*1e3:.2f}")
