"""Golden vectors for the MAE-decoder heads: instantiates the REFERENCE's MAEBoxHeadRec, MAEBoxHeadMIL and
MAEMaskHeadPointSup (their files loaded by path with mmcv / mmdet stubbed out -- only constructors and `forward` run,
which are plain torch + the reference's own models/vision_transformer.Block), randomises every parameter, runs forward
on seeded RoI features and stores state dicts + inputs + outputs in tests/golden/mae_heads.npz.
Container-only (needs /root/reference)."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


class FakeRegistry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def ident_decorator(*a, **k):
    return lambda f: f


class BBoxHead(nn.Module):                         # what the MAE box heads read off mmdet's BBoxHead
    def __init__(self, with_cls=True, with_reg=True, num_classes=20, reg_class_agnostic=False, **kw):
        super().__init__()
        self.with_cls, self.with_reg, self.num_classes, self.reg_class_agnostic = with_cls, with_reg, num_classes, reg_class_agnostic


def load(name, path, package):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    sys.path.insert(0, REF)                                                  # root utils.py -> trunc_normal_
    stub("timm"); stub("timm.models"); stub("timm.models.registry", register_model=lambda f: f)
    vt = load("models.vision_transformer", REF + "/models/vision_transformer.py", "models")
    stub("models", vision_transformer=vt)
    sys.modules["models.vision_transformer"] = vt
    stub("mmcv"); stub("mmcv.runner", auto_fp16=ident_decorator, force_fp32=ident_decorator, _load_checkpoint=None,
                       load_state_dict=None)
    stub("mmcv.cnn", Conv2d=nn.Conv2d, ConvModule=None, build_upsample_layer=None)
    stub("mmcv.ops"); stub("mmcv.ops.carafe", CARAFEPack=None)
    stub("mmdet"); stub("mmdet.utils", get_root_logger=lambda *a, **k: None)
    stub("mmdet.core", mask_target=None)
    stub("mmdet.models"); stub("mmdet.models.builder", HEADS=FakeRegistry(), build_loss=lambda cfg: None)
    stub("mmdet.models.losses", accuracy=None)
    stub("refpkg"); stub("refpkg.roi_heads"); stub("refpkg.roi_heads.bbox_heads"); stub("refpkg.roi_heads.mask_heads")
    stub("refpkg.roi_heads.bbox_heads.bbox_head", BBoxHead=BBoxHead)
    stub("refpkg.utils"); stub("refpkg.utils.positional_encoding", get_2d_sincos_pos_embed=None)
    rec = load("refpkg.roi_heads.bbox_heads.mae_bbox_head_rec", REF + "/mmdet/models/roi_heads/bbox_heads/mae_bbox_head_rec.py",
               "refpkg.roi_heads.bbox_heads")
    mil = load("refpkg.roi_heads.bbox_heads.mae_bbox_head_mil", REF + "/mmdet/models/roi_heads/bbox_heads/mae_bbox_head_mil.py",
               "refpkg.roi_heads.bbox_heads")
    msk = load("refpkg.roi_heads.mask_heads.mae_mask_head_pointSup",
               REF + "/mmdet/models/roi_heads/mask_heads/mae_mask_head_pointSup.py", "refpkg.roi_heads.mask_heads")
    gen = torch.Generator().manual_seed(321)
    st = {}

    def randomise(m):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * (0.5 if p.dim() == 1 else 1.0 / max(p.shape[-1], 1) ** 0.5))

    def dump(tag, m):
        sd = m.state_dict()
        st[f"{tag}_keys"] = np.array(list(sd.keys()))
        for k, v in sd.items():
            st[f"{tag}.{k}"] = v.numpy()

    C, E, K = 48, 64, 5
    box = rec.MAEBoxHeadRec(in_channels=C, img_size=224, patch_size=16, embed_dim=E, depth=2, num_heads=2, mlp_ratio=4.,
                            with_reconstruct=True, num_classes=K, cam_layer=3).eval()
    randomise(box); dump("box", box)
    x7 = torch.randn(6, C, 7, 7, generator=gen)
    x5x9 = torch.randn(3, C, 5, 9, generator=gen)                           # non-square RoI grid: resized pos-embed
    with torch.no_grad():
        for tag, x in (("box7", x7), ("box59", x5x9)):
            cls, reg, recon = box(x)
            st[f"{tag}_x"], st[f"{tag}_cls"], st[f"{tag}_reg"], st[f"{tag}_rec"] = x.numpy(), cls.numpy(), reg.numpy(), recon.numpy()
    Lq = 3
    m = mil.MAEBoxHeadMIL(in_channels=C, embed_dim=E, num_classes=K, num_layers_query=Lq, hidden_dim=32, roi_size=7,
                          with_cls=False, with_reg=False).eval()
    randomise(m); dump("mil", m)
    xm = torch.randn(4 * Lq, C, 7, 7, generator=gen)
    labels = torch.tensor([1, 4, 0, 2])
    with torch.no_grad():
        idx, loss = m(xm, gt_labels=[labels[:3], labels[3:]])
    st["mil_x"], st["mil_labels"], st["mil_idx"], st["mil_loss"] = xm.numpy(), labels.numpy(), idx.numpy(), loss.numpy()
    mk = msk.MAEMaskHeadPointSup(roi_feat_size=14, num_classes=K, in_channels=C, img_size=224, patch_size=16, embed_dim=E,
                                 depth=2, num_heads=2, scale_factor=2, scale_mode="bicubic").eval()
    randomise(mk); dump("mask", mk)
    x14 = torch.randn(3, C, 14, 14, generator=gen)
    with torch.no_grad():
        st["mask14_x"], st["mask14_out"] = x14.numpy(), mk(x14).numpy()
        st["mask7_x"], st["mask7_out"] = x7[:2].numpy(), mk(x7[:2]).numpy()  # the training path feeds the 7x7 box features
    path = os.path.join(ROOT, "tests", "golden", "mae_heads.npz")
    np.savez_compressed(path, **st)
    print("wrote", path, len(st), "arrays;", {k: st[k].shape for k in ("box7_cls", "box59_rec", "mil_idx", "mask14_out", "mask7_out")})


if __name__ == "__main__":
    main()
