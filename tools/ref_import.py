"""Import the hot-path pieces of the reference (read-only at /root/reference) in THIS container.

Golden-generation infrastructure only.  Nothing under attentionshift_amd/, bench.py,
__graft_entry__.py or the `-m gpu` tests may import this module: /root/reference does not
exist on the GPU box.  See SURVEY.md section 8c for why a plain `import mmdet` is impossible
(mmcv/timm/cv2/cc_torch absent, and the shipped tree does not import even with them).

Two entry points:
  load_roi_functions()  -> namespace dict with every module-level function of
                           mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py plus the
                           selected methods of its RoI-head class, exec'd in file order.
  load_backbone()       -> (vision_transformer module, visual_transformer_det module)
"""
import ast
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("ATTNSHIFT_REFERENCE", "/root/reference")
ROI_SRC = os.path.join(REF, "mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py")

WANTED_METHODS = (
    "mean_shift_grid_prototype",
    "get_semantic_centers",
    "get_mask_sample_points_roi_best_attn_feat_refine",
)


def reference_available():
    return os.path.isfile(ROI_SRC)


def scipy_ccl(x):
    """Stand-in for the absent cc_torch extension: 8-connectivity partition (the only
    property the consumer at stdroi...:68-86 relies on).  Numbering = 1 + min raster index."""
    from scipy import ndimage

    a = x.detach().cpu().numpy().astype(np.uint8)
    lab, n = ndimage.label(a, structure=np.ones((3, 3), dtype=np.int32))
    out = np.zeros(a.shape, dtype=np.int32)
    if n:
        flat = lab.ravel()
        idx = np.arange(flat.size, dtype=np.int64)
        first = np.full(n + 1, flat.size, dtype=np.int64)
        np.minimum.at(first, flat, idx)
        out = np.where(lab > 0, first[lab] + 1, 0).astype(np.int32)
    return torch.from_numpy(out).to(x.device)


def load_roi_functions():
    tree = ast.parse(open(ROI_SRC).read())
    ns = dict(torch=torch, F=F, nn=nn, math=math, np=np,
              connected_components_labeling=scipy_ccl)
    for node in tree.body:  # file order: later duplicate defs win, as in Python
        if isinstance(node, ast.FunctionDef):
            exec(compile(ast.Module([node], []), ROI_SRC, "exec"), ns)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef))
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANTED_METHODS:
            node.decorator_list = []
            exec(compile(ast.Module([node], []), ROI_SRC, "exec"), ns)
    return ns


class _FakeRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_backbone():
    _stub("timm"); _stub("timm.models")
    _stub("timm.models.registry", register_model=lambda f: f)
    if REF not in sys.path:
        sys.path.insert(0, REF)  # root utils.py -> trunc_normal_
    vt = _load_by_path("models.vision_transformer", os.path.join(REF, "models/vision_transformer.py"))
    _stub("models", VisionTransformer=vt.VisionTransformer, vision_transformer=vt)
    _stub("mmcv_custom", load_checkpoint=lambda *a, **k: None)
    import logging
    _stub("mmdet"); _stub("mmdet.utils", get_root_logger=lambda *a, **k: logging.getLogger("ref"))
    _stub("mmdet.models"); _stub("mmdet.models.builder", BACKBONES=_FakeRegistry())
    det = _load_by_path("ref_vtd", os.path.join(REF, "mmdet/models/backbones/visual_transformer_det.py"))
    return vt, det


def load_swin():
    """models/swin_transformer.py with `timm` stubbed (DropPath at rate 0 is the identity; to_2tuple; trunc_normal_)."""
    import torch.nn as nn

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x

    _stub("timm"); _stub("timm.models")
    _stub("timm.models.registry", register_model=lambda f: f)
    _stub("timm.models.layers", DropPath=DropPath, to_2tuple=lambda v: (v, v) if not isinstance(v, tuple) else v,
          trunc_normal_=lambda t, std=0.02: nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std))
    return _load_by_path("ref_swin", os.path.join(REF, "models/swin_transformer.py"))


def load_swin_det():
    """mmdet/models/backbones/swin_transformer.py (the BACKBONES-registered detection backbone, :448-630) with timm /
    mmcv_custom / mmdet stubbed as above."""
    import logging
    load_swin()                                  # installs the timm.models.layers stub
    _stub("mmcv_custom", load_checkpoint=lambda *a, **k: None)
    _stub("mmdet"); _stub("mmdet.utils", get_root_logger=lambda *a, **k: logging.getLogger("ref"))
    _stub("mmdet.models"); _stub("mmdet.models.builder", BACKBONES=_FakeRegistry())
    _stub("mmdet.models.backbones")
    path = os.path.join(REF, "mmdet/models/backbones/swin_transformer.py")
    spec = importlib.util.spec_from_file_location("mmdet.models.backbones.swin_transformer", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["mmdet.models.backbones.swin_transformer"] = mod
    spec.loader.exec_module(mod)
    return mod
