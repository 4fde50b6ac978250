"""Golden vectors for checkpoint loading: executes the reference's own mmcv_custom/checkpoint.py
`load_checkpoint` + `load_state_dict` (extracted with ast; file reading replaced by handing over the dict, mmcv's
`get_dist_info` / `is_module_wrapper` stubbed) on a small Swin-shaped module for three container / prefix variants incl.
a relative-position table of another window size, and stores the checkpoint tensors + the model's final state in
tests/golden/checkpoint_load.npz.  Container-only (needs /root/reference)."""
import ast
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference/mmcv_custom/checkpoint.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Attn(nn.Module):
    def __init__(self, window, heads):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window - 1) ** 2, heads))
        self.qkv = nn.Linear(8, 24)


class Net(nn.Module):                                  # parameter names as in models/swin_transformer.py
    def __init__(self, window=3, heads=2):
        super().__init__()
        self.patch_embed = nn.Conv2d(3, 8, 2, 2)
        self.layers = nn.ModuleList([nn.ModuleDict(dict(attn=Attn(window, heads), norm=nn.BatchNorm2d(8)))])
        self.head_only_in_model = nn.Linear(8, 4)


def main():
    tree = ast.parse(open(REF).read())
    ns = {"torch": torch, "F": F, "OrderedDict": OrderedDict, "get_dist_info": lambda: (0, 1),
          "is_module_wrapper": lambda m: False, "warnings": __import__("warnings")}
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in ("load_state_dict", "load_checkpoint"):
            exec(compile(ast.Module(body=[n], type_ignores=[]), REF, "exec"), ns)
    msgs = []
    logger = type("L", (), {"warning": lambda self, m: msgs.append(m), "info": lambda self, m: None})()
    gen = torch.Generator().manual_seed(11)
    src = Net(window=5, heads=2)                       # the file was trained with window 5: 81-entry tables
    with torch.no_grad():
        for p in src.parameters():
            p.copy_(torch.randn(p.shape, generator=gen))
        src.layers[0]["norm"].running_mean.copy_(torch.randn(8, generator=gen))
    base = OrderedDict((k, v.clone()) for k, v in src.state_dict().items() if not k.startswith("head_only_in_model"))
    base["unexpected.weight"] = torch.randn(3, generator=gen)
    variants = {
        "module_state_dict": dict(state_dict=OrderedDict(("module." + k, v) for k, v in base.items()), meta=dict(epoch=3)),
        "moby_model": dict(model=OrderedDict([("encoder." + k, v) for k, v in base.items()] +
                                             [("projector.w", torch.zeros(2))])),
        "bare": OrderedDict(base),
    }
    st = {"variants": np.array(list(variants))}
    for k, v in base.items():
        st[f"ckpt.{k}"] = v.numpy()
    st["ckpt_keys"] = np.array(list(base))
    for name, ckpt in variants.items():
        torch.manual_seed(5)
        model = Net(window=3, heads=2)
        ns["_load_checkpoint"] = lambda filename, map_location=None, c=ckpt: c
        ns["load_checkpoint"](model, "unused", strict=False, logger=logger)
        for k, v in model.state_dict().items():
            st[f"{name}.{k}"] = v.numpy()
        st[f"{name}_keys"] = np.array(list(model.state_dict()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "checkpoint_load.npz"), **st)
    print("wrote; reference messages:", len(msgs), [m[:90] for m in msgs[:3]])


if __name__ == "__main__":
    main()
