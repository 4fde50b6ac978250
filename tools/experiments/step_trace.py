"""Chrome trace of ONE headline step (torch.profiler) -> gpurun_out/step_trace.json, plus a text timeline of the device
activity (kernels / copies per stream with the gaps between them) and of the host-side synchronisation calls.
    python tools/experiments/step_trace.py   (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench


def main():
    bench.CFG.clear()
    bench.CFG.update(bench.CONFIGS["vitb"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    step = bench.build(dev, os.environ.get("AS_RNG_MODE", "fast"))
    with torch.no_grad():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "step_trace.json")
    prof.export_chrome_trace(path)
    ev = json.load(open(path))["traceEvents"]
    dev_ev = sorted((e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e), key=lambda e: e["ts"])
    t0 = dev_ev[0]["ts"]
    lines = []
    for e in dev_ev:
        lines.append(f"{e['ts'] - t0:9.1f} +{e['dur']:7.1f}  s{e.get('args', {}).get('stream', '?'):>3}  {e['name'][:90]}")
    sync = sorted((e for e in ev if e.get("cat") in ("cuda_runtime", "cuda_driver") and any(k in e["name"] for k in
                  ("Synchronize", "Memcpy", "EventQuery")) and "dur" in e), key=lambda e: e["ts"])
    lines.append("---- host synchronisation / copies ----")
    for e in sync:
        lines.append(f"{e['ts'] - t0:9.1f} +{e['dur']:7.1f}  host {e['name']}")
    open(os.path.join(ROOT, "gpurun_out", "step_timeline.txt"), "w").write("\n".join(lines) + "\n")
    print(len(dev_ev), "device events;", len(sync), "host sync/copy calls; span",
          round(dev_ev[-1]["ts"] + dev_ev[-1]["dur"] - t0, 1), "us")


if __name__ == "__main__":
    main()
