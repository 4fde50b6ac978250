"""Which calls of a training step synchronise the host with the device (torch.cuda.set_sync_debug_mode("warn")), by source
line inside this repo.    python tools/experiments/train_syncs.py     (GPU box)"""
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
step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
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
print("synchronising calls in one training step:", sum(sites.values()))
for s, n in sites.most_common():
    print(f"{n:4d}  {s}")
