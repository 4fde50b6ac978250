"""Which calls of a training step (or, with `infer`, of the headline forward + attention-shift step) synchronise the host
with the device (torch.cuda.set_sync_debug_mode("warn")), by source line inside this repo.
    python tools/experiments/train_syncs.py [infer [mil]]     (GPU box)"""
import collections
import os
import sys
import traceback
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from attentionshift_amd.dist import Ranks  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.set_device(0)
INFER = len(sys.argv) > 1 and sys.argv[1] == "infer"
if INFER:
    step = bench.build(torch.device("cuda", 0), "fast", mil="mil" in sys.argv[2:])
else:
    step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
if INFER:
    torch.set_grad_enabled(False)                    # the headline step is a no-grad forward (as bench.py runs it)
for _ in range(3):
    step()
torch.cuda.synchronize()
sites = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    site = "?"
    stack = traceback.extract_stack(limit=40)
    for k in range(len(stack) - 1, -1, -1):
        fr = stack[k]
        if ROOT in fr.filename and "train_syncs" not in fr.filename:
            site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} ({fr.name})"
            if k + 1 < len(stack) - 2:                       # what the repo line called (torch internals)
                site += " -> " + stack[k + 1].name
            break
    sites[site] += 1


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one %s step:" % ("headline" if INFER else "training"), sum(sites.values()))
for s, n in sites.most_common():
    print(f"{n:4d}  {s}")
