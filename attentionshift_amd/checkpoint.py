"""On-disk formats of the reference's training pipeline, host side only.

* `save_checkpoint` writes the runner's schema `{meta, state_dict, optimizer, amp}`
  (reference mmcv_custom/runner/checkpoint.py:19-58; `amp` is apex state there -- apex is not part of this build, the
  key is kept so that the reference's resume code finds it).
* `load_checkpoint` accepts what the reference's loader accepts (mmcv_custom/checkpoint.py:286-356): a bare state
  dict or one under `state_dict` / `model`; a `module.` (DataParallel) prefix; MoBY's `encoder.` branch; Swin's
  `absolute_pos_embed` stored as tokens; `relative_position_bias_table`s of another window size (bicubic resize);
  non-strict by default with the mismatch report of mmcv_custom/checkpoint.py:41-107.
Nothing here touches the device; modules that cache compute-dtype weights are told to drop them.
"""
import os
import time
from collections import OrderedDict

import torch
import torch.nn.functional as F

VERSION = "attentionshift_amd-r01"


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module) else model


def _state_dict_of(checkpoint):
    if not isinstance(checkpoint, dict):
        raise RuntimeError("checkpoint file holds no state_dict")
    for key in ("state_dict", "model"):
        if key in checkpoint:
            return checkpoint[key]
    return checkpoint


def _strip_prefixes(sd):
    keys = list(sd.keys())
    if keys and keys[0].startswith("module."):
        sd = OrderedDict((k[len("module."):], v) for k, v in sd.items())
    if sd and sorted(sd.keys())[0].startswith("encoder"):          # MoBY: the online branch only
        sd = OrderedDict((k.replace("encoder.", ""), v) for k, v in sd.items() if k.startswith("encoder."))
    return sd


def _adapt_swin_tables(model, sd, warn):
    own = model.state_dict()
    ape = sd.get("absolute_pos_embed")
    if ape is not None and "absolute_pos_embed" in own:
        n1, length, c1 = ape.shape
        n2, c2, h, w = own["absolute_pos_embed"].shape
        if n1 != n2 or c1 != c2 or length != h * w:
            warn("absolute_pos_embed does not fit, skipped")
            del sd["absolute_pos_embed"]
        else:
            sd["absolute_pos_embed"] = ape.view(n2, h, w, c2).permute(0, 3, 1, 2)
    for key in [k for k in sd if "relative_position_bias_table" in k]:
        if key not in own:
            continue
        src, dst = sd[key], own[key]
        (l1, h1), (l2, h2) = src.shape, dst.shape
        if h1 != h2:
            warn(f"{key}: {h1} heads in the file, {h2} in the model, skipped")
            del sd[key]
        elif l1 != l2:
            s1, s2 = int(l1 ** 0.5), int(l2 ** 0.5)
            grid = F.interpolate(src.permute(1, 0).reshape(1, h1, s1, s1), size=(s2, s2), mode="bicubic")
            sd[key] = grid.reshape(h2, l2).permute(1, 0)
    return sd


def load_state_dict(module, state_dict, strict=False, logger=None):
    """Non-strict load with the reference's report (mmcv_custom/checkpoint.py:41-107): unexpected keys, missing keys
    (BatchNorm's `num_batches_tracked` ignored), shape mismatches; raises only when `strict`.
    Returns (missing, unexpected)."""
    net = _unwrap(module)
    own = net.state_dict()
    bad = [k for k, v in state_dict.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    msgs = [f"size mismatch for {k}: file {tuple(state_dict[k].shape)} vs model {tuple(own[k].shape)}" for k in bad]
    result = net.load_state_dict(OrderedDict((k, v) for k, v in state_dict.items() if k not in bad), strict=False)
    missing = [k for k in result.missing_keys if "num_batches_tracked" not in k and k not in bad]
    unexpected = list(result.unexpected_keys)
    if unexpected:
        msgs.append("unexpected key in source state_dict: " + ", ".join(unexpected))
    if missing:
        msgs.append("missing keys in source state_dict: " + ", ".join(missing))
    if msgs:
        text = "The model and loaded state dict do not match exactly\n" + "\n".join(msgs)
        if strict:
            raise RuntimeError(text)
        (logger.warning if logger is not None else print)(text)
    return missing, unexpected


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    """mmcv_custom/checkpoint.py:286-356.  Returns the loaded checkpoint object (so callers can read `meta`,
    `optimizer`, `amp`)."""
    if not os.path.isfile(filename):
        raise IOError(f"{filename} is not a checkpoint file")
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    warn = (logger.warning if logger is not None else print)
    sd = _strip_prefixes(OrderedDict(_state_dict_of(checkpoint)))
    own = _unwrap(model).state_dict()
    if not any(k in own for k in sd) and any(k.startswith("backbone.") for k in sd):
        # extension: a detector checkpoint handed to the bare backbone (the reference would load nothing)
        sd = OrderedDict((k[len("backbone."):], v) for k, v in sd.items() if k.startswith("backbone."))
    sd = _adapt_swin_tables(_unwrap(model), sd, warn)
    load_state_dict(model, sd, strict, logger)
    for m in _unwrap(model).modules():                             # compute-dtype weight caches (backbone.py)
        if hasattr(m, "invalidate_cache"):
            m.invalidate_cache()
    return checkpoint


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """mmcv_custom/runner/checkpoint.py:19-58: `{meta, state_dict, optimizer, amp}`; weights on the CPU."""
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError(f"meta must be a dict or None, but got {type(meta)}")
    meta = dict(meta, version=VERSION, time=time.asctime())
    net = _unwrap(model)
    if getattr(net, "CLASSES", None) is not None:
        meta.update(CLASSES=net.CLASSES)
    ckpt = {"meta": meta, "state_dict": OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items())}
    if isinstance(optimizer, torch.optim.Optimizer):
        ckpt["optimizer"] = optimizer.state_dict()
    elif isinstance(optimizer, dict):
        ckpt["optimizer"] = {name: opt.state_dict() for name, opt in optimizer.items()}
    ckpt["amp"] = None                                              # apex.amp.state_dict() in the reference
    folder = os.path.dirname(filename)
    if folder:
        os.makedirs(folder, exist_ok=True)
    with open(filename, "wb") as f:
        torch.save(ckpt, f)
        f.flush()
    return ckpt
