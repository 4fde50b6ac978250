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
    # label the head's stages (monkeypatched record_function scopes: no product code involved)
    import functools
    from torch.profiler import record_function
    from attentionshift_amd import roi_head as RH

    def scoped(fn, name):
        @functools.wraps(fn)
        def w(*a, **k):
            with record_function("HEAD:" + name):
                return fn(*a, **k)
        return w

    for name in ("mask_points_and_pseudo_issue", "mask_points_nosync", "grid_seed_nosync", "feature_tokens"):
        if hasattr(RH, name):
            setattr(RH, name, scoped(getattr(RH, name), name))
    head = step.head
    for name in ("rollout_cams", "refine_maps", "_semantic_pre", "mean_shift_batch", "_semantic_post_issue",
                 "_semantic_post_finish"):         # (not layer_selector: the head fuses the default selector by identity)
        if hasattr(head, name):
            setattr(head, name, scoped(getattr(head, name), name))
    from attentionshift_amd import ops as OPS
    OPS.cam_boxes = scoped(OPS.cam_boxes, "cam_boxes")
    with torch.no_grad():
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        # the tracer occasionally stalls the host for tens of ms inside the traced step (seen once as a 72 ms hole between the
        # two images' chains): trace up to four single steps and keep the one with the shortest device span
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "step_trace.json")
        best = None
        for attempt in range(4):
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                step()
                torch.cuda.synchronize()
            tmp = path + ".tmp"
            prof.export_chrome_trace(tmp)
            ev_a = json.load(open(tmp))["traceEvents"]
            dev_a = sorted((e for e in ev_a if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e),
                           key=lambda e: e["ts"])
            span = dev_a[-1]["ts"] + dev_a[-1]["dur"] - dev_a[0]["ts"]
            print(f"traced step {attempt}: device span {span:.0f} us")
            if best is None or span < best[0]:
                best = (span, ev_a, dev_a)
                os.replace(tmp, path)
            if span < 9000:
                break
    ev, dev_ev = best[1], best[2]
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
    # attribute device events to the labelled host scopes through the launch correlation ids
    scopes = sorted((e for e in ev if e.get("cat") == "user_annotation" and e["name"].startswith("HEAD:")), key=lambda e: e["ts"])
    launches = {e["args"]["correlation"]: e for e in ev if e.get("cat") in ("cuda_runtime", "cuda_driver") and "correlation" in e.get("args", {})}
    agg = {}
    for e in dev_ev:
        l = launches.get(e.get("args", {}).get("correlation"))
        label = "(backbone / other)"
        if l is not None:
            inner = None
            for sc in scopes:
                if sc["ts"] <= l["ts"] <= sc["ts"] + sc["dur"] and (inner is None or sc["dur"] < inner["dur"]):
                    inner = sc
            if inner is not None:
                label = inner["name"]
            elif l["ts"] > (scopes[0]["ts"] if scopes else 1e30):
                label = "(head, unlabelled)"
        a = agg.setdefault(label, [0, 0.0])
        a[0] += 1
        a[1] += e["dur"]
    rep = ["label | device events | device us"] + [f"{k} | {v[0]} | {v[1]:.1f}" for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    open(os.path.join(ROOT, "gpurun_out", "step_scopes.txt"), "w").write("\n".join(rep) + "\n")
    print("\n".join(rep))
    print(len(dev_ev), "device events;", len(sync), "host sync/copy calls; span",
          round(dev_ev[-1]["ts"] + dev_ev[-1]["dur"] - t0, 1), "us")


if __name__ == "__main__":
    main()
