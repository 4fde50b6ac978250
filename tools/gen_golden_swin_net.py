"""Golden vectors for the whole Swin backbone (BASELINE config 5 family): runs the reference's own SwinTransformer
(models/swin_transformer.py, timm stubbed by tools/ref_import.py) with deterministic weights on a seeded image and
stores the per-stage tokens and the output in tests/golden/swin_net_*.npz.  Container-only (needs /root/reference)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402
from attentionshift_amd import synthetic  # noqa: E402


def case(tag, img, cfg, seed):
    ref = ref_import.load_swin()
    m = ref.SwinTransformer(**cfg).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if "relative_position_index" not in k and "attn_mask" not in k}
    sd = synthetic.det_state_dict(shapes)
    missing = m.load_state_dict(sd, strict=False)
    assert all("relative_position_index" in k or "attn_mask" in k for k in missing.missing_keys), missing
    x = torch.randn(2, 3, img, img, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        t = m.patch_embed(x).flatten(2).transpose(1, 2)
        stages = []
        for layer in m.layers:
            t = layer(t)
            stages.append(t)
        out = m(x, return_all_tokens=True)
    store = dict(img=np.int64(img), seed=np.int64(seed), out=out.numpy(),
                 cfg_embed_dim=np.int64(cfg["embed_dim"]), cfg_depths=np.array(cfg["depths"]), cfg_heads=np.array(cfg["num_heads"]),
                 param_names=np.array(sorted(shapes)), **{f"stage{i}": s.numpy() for i, s in enumerate(stages)})
    path = os.path.join(ROOT, "tests", "golden", f"swin_net_{tag}.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, out.shape, [s.shape for s in stages])


def det_case(tag, hw, cfg, seed):
    """The detection backbone (mmdet/models/backbones/swin_transformer.py:448-630) on a NON-square image that needs
    patch padding, window padding and odd-grid merging; stores every output map."""
    ref = ref_import.load_swin_det()
    m = ref.SwinTransformer(**cfg)
    m.eval()                                     # (the reference's train() override returns None: no chaining)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if "relative_position_index" not in k}
    sd = synthetic.det_state_dict(shapes)
    missing = m.load_state_dict(sd, strict=False)
    assert all("relative_position_index" in k for k in missing.missing_keys), missing
    x = torch.randn(2, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        outs = m(x)
    store = dict(hw=np.array(hw), seed=np.int64(seed), cfg_embed_dim=np.int64(cfg["embed_dim"]),
                 cfg_depths=np.array(cfg["depths"]), cfg_heads=np.array(cfg["num_heads"]),
                 cfg_out_indices=np.array(cfg["out_indices"]), cfg_ape=np.int64(cfg.get("ape", False)),
                 state_keys=np.array(sorted(m.state_dict().keys())), param_names=np.array(sorted(shapes)),
                 param_shapes=np.array([",".join(map(str, shapes[k])) for k in sorted(shapes)]),
                 **{f"out{i}": o.numpy() for i, o in enumerate(outs)})
    path = os.path.join(ROOT, "tests", "golden", f"swin_det_{tag}.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, [tuple(o.shape) for o in outs])


if __name__ == "__main__":
    if "--det" in sys.argv:
        det_case("r98x118", (98, 118), dict(pretrain_img_size=64, patch_size=4, embed_dim=32, depths=[2, 2, 2], num_heads=[1, 2, 4],
                                            window_size=7, drop_path_rate=0.0, ape=True, out_indices=(0, 1, 2)), 21)
        sys.exit(0)
    case("pad120", 120, dict(img_size=120, patch_size=4, in_chans=3, num_classes=0, embed_dim=32, depths=[2, 2],
                             num_heads=[1, 2], window_size=7), 11)
