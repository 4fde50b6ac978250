"""Checkpoint formats of the reference pipeline: CPU only."""
import os

import pytest
import torch

import attentionshift_amd as A
from attentionshift_amd import checkpoint as CK


def _tiny_backbone(golden=None):
    return A.build_backbone(dict(type="VisionTransformerDet", img_size=64, patch_size=16, embed_dim=128, depth=4,
                                 num_heads=2, mlp_ratio=4., qkv_bias=True, out_indices=(0, 1, 2, 3), last_feat=True,
                                 point_tokens_num=10, num_classes=5, return_attention=True))


def test_save_schema_and_round_trip(tmp_path, golden):
    bb = _tiny_backbone(golden)
    opt = torch.optim.AdamW(bb.parameters(), lr=1e-3)
    path = os.path.join(tmp_path, "sub", "epoch_1.pth")
    CK.save_checkpoint(bb, path, optimizer=opt, meta=dict(epoch=1, iter=10))
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"meta", "state_dict", "optimizer", "amp"}                    # runner/checkpoint.py:46-58
    assert raw["meta"]["epoch"] == 1 and "time" in raw["meta"] and "param_groups" in raw["optimizer"]
    assert all(v.device.type == "cpu" for v in raw["state_dict"].values())
    bb2 = _tiny_backbone(golden)
    with torch.no_grad():
        for p in bb2.parameters():
            p.add_(1.0)
    ck = CK.load_checkpoint(bb2, path, strict=True)
    assert ck["meta"]["iter"] == 10
    for (k, a), (_, b) in zip(bb.state_dict().items(), bb2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(TypeError):
        CK.save_checkpoint(bb, path, meta=3)
    with pytest.raises(IOError):
        CK.load_checkpoint(bb, os.path.join(tmp_path, "missing.pth"))


def test_prefixes_containers_and_non_strict_report(tmp_path, golden):
    bb = _tiny_backbone(golden)
    sd = bb.state_dict()
    logs = []
    logger = type("L", (), {"warning": lambda self, m: logs.append(m)})()
    # DataParallel prefix under the MAE-style 'model' key, one tensor missing, one of the wrong shape, one unknown
    broken = {"module." + k: v.clone() for k, v in sd.items()}
    del broken["module.cls_token"]
    broken["module.pos_embed"] = torch.zeros(1, 5, 3)
    broken["module.decoder_pred.weight"] = torch.zeros(2, 2)
    path = os.path.join(tmp_path, "mae.pth")
    torch.save({"model": broken}, path)
    bb2 = _tiny_backbone(golden)
    before = bb2.pos_embed.detach().clone()
    CK.load_checkpoint(bb2, path, logger=logger)                       # non-strict: loads what fits, reports the rest
    text = "\n".join(logs)
    assert "decoder_pred.weight" in text and "cls_token" in text and "size mismatch for pos_embed" in text
    assert torch.equal(bb2.pos_embed, before)
    assert torch.equal(bb2.blocks[3].attn.qkv.weight, bb.blocks[3].attn.qkv.weight)
    with pytest.raises(RuntimeError):
        CK.load_checkpoint(_tiny_backbone(golden), path, strict=True)
    # MoBY: only the online encoder branch is taken (mmcv_custom/checkpoint.py:318-320)
    moby = {"encoder." + k: v for k, v in sd.items()}
    moby.update({"encoder_k." + k: torch.zeros_like(v) for k, v in sd.items()})
    torch.save(moby, path)
    bb3 = _tiny_backbone(golden)
    CK.load_checkpoint(bb3, path, strict=True)
    assert torch.equal(bb3.point_token, bb.point_token) and torch.equal(bb3.blocks[1].norm1.weight, bb.blocks[1].norm1.weight)
    # a detector checkpoint handed to the bare backbone (extension)
    torch.save({"state_dict": {"backbone." + k: v for k, v in sd.items()}}, path)
    bb4 = _tiny_backbone(golden)
    CK.load_checkpoint(bb4, path, strict=True)
    assert torch.equal(bb4.patch_embed.proj.weight, bb.patch_embed.proj.weight)


def test_init_weights_loads_a_pretrained_file(tmp_path, golden):
    bb = _tiny_backbone(golden)
    path = os.path.join(tmp_path, "pre.pth")
    CK.save_checkpoint(bb, path)
    bb2 = _tiny_backbone(golden)
    bb2.init_weights(path)                                             # visual_transformer_det.py:179-190
    assert torch.equal(bb2.blocks[0].mlp.fc1.weight, bb.blocks[0].mlp.fc1.weight)
    bb2.init_weights(os.path.join(tmp_path, "nope.pth"))               # invalid path: random init, no exception
    with pytest.raises(TypeError):
        bb2.init_weights(123)


def test_swin_relative_position_tables_are_resized(tmp_path):
    from attentionshift_amd.swin import WindowAttention
    src, dst = WindowAttention(32, 7, 4), WindowAttention(32, 12, 4)
    path = os.path.join(tmp_path, "swin.pth")
    torch.save({"model": src.state_dict()}, path)
    CK.load_checkpoint(dst, path)
    table = src.relative_position_bias_table.detach()
    want = torch.nn.functional.interpolate(table.permute(1, 0).reshape(1, 4, 13, 13), size=(23, 23), mode="bicubic")
    assert torch.allclose(dst.relative_position_bias_table, want.reshape(4, 23 * 23).permute(1, 0))


def test_loading_equals_the_reference_loader(tmp_path, golden):
    """Fixture = the reference's own mmcv_custom/checkpoint.py load_checkpoint executed on a Swin-shaped module for the
    `module.`-prefixed runner schema, the MoBY `encoder.` container and a bare state dict, with window-5 relative
    position tables going into a window-3 model (tools/gen_golden_checkpoint.py): same final parameters here."""
    import torch.nn as nn
    from attentionshift_amd import checkpoint as CK
    g = golden("checkpoint_load")

    class Attn(nn.Module):
        def __init__(self, window, heads):
            super().__init__()
            self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window - 1) ** 2, heads))
            self.qkv = nn.Linear(8, 24)

    class Net(nn.Module):
        def __init__(self, window=3, heads=2):
            super().__init__()
            self.patch_embed = nn.Conv2d(3, 8, 2, 2)
            self.layers = nn.ModuleList([nn.ModuleDict(dict(attn=Attn(window, heads), norm=nn.BatchNorm2d(8)))])
            self.head_only_in_model = nn.Linear(8, 4)

    base = {str(k): torch.from_numpy(g[f"ckpt.{k}"]) for k in g["ckpt_keys"]}
    files = {"module_state_dict": dict(state_dict={"module." + k: v for k, v in base.items()}, meta=dict(epoch=3)),
             "moby_model": dict(model={**{"encoder." + k: v for k, v in base.items()}, "projector.w": torch.zeros(2)}),
             "bare": dict(base)}
    for name in [str(v) for v in g["variants"]]:
        path = tmp_path / f"{name}.pth"
        torch.save(files[name], path)
        torch.manual_seed(5)
        model = Net()
        CK.load_checkpoint(model, str(path), strict=False)
        sd = model.state_dict()
        assert list(sd) == [str(k) for k in g[f"{name}_keys"]]
        for k, v in sd.items():
            assert torch.allclose(v.float(), torch.from_numpy(g[f"{name}.{k}"]).float(), atol=1e-6), (name, k)
