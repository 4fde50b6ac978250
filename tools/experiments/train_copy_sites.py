"""Call sites of the LARGE copies (>= 1M elements: clone / contiguous / _to_copy / copy_ / cat / add_) of one training step,
forward AND backward (the dispatch mode is entered on the autograd thread through a hook as well):
    python tools/experiments/train_copy_sites.py          (GPU box)"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_leaves

import bench
from attentionshift_amd.dist import Ranks

WATCH = ("aten.clone", "aten._to_copy", "aten.copy_", "aten.cat", "aten.add_", "aten.add.", "aten.index", "aten.mul", "aten.zeros",
         "aten.fill_", "aten.zero_", "aten.index_put", "aten.new_zeros", "aten.permute_copy")
sites = collections.Counter()
bytes_ = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(WATCH):
            big = [t for t in tree_leaves((args, out)) if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() >= (1 << 20)]
            if big:
                fr = [f for f in traceback.extract_stack() if ROOT in f.filename and "train_copy_sites" not in f.filename]
                where = f"{os.path.relpath(fr[-1].filename, ROOT)}:{fr[-1].lineno} ({fr[-1].name})" if fr else "autograd engine (no repo frame)"
                shp = "x".join(str(s) for s in big[0].shape)
                key = (name.replace("aten.", ""), where, shp, str(big[0].dtype).replace("torch.", ""))
                sites[key] += 1
                bytes_[key] += sum(t.numel() * t.element_size() for t in big)
        return out


torch.cuda.set_device(0)
step = bench.build(torch.device("cuda", 0), "fast", train=True, ranks=Ranks())
for _ in range(3):
    step()
torch.cuda.synchronize()
with Log():
    step()
torch.cuda.synchronize()
for key, n in sorted(sites.items(), key=lambda kv: -bytes_[kv[0]])[:45]:
    print(f"{bytes_[key] / 1e6:8.1f} MB {n:3d}x  {key[0]:22s} {key[2]:22s} {key[3]:9s} {key[1]}")
