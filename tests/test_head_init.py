"""init_weights of the MAE-decoder heads: the MAE pre-training checkpoint's decoder entries are loaded with
the encoder's skipped, and a mask head trained from scratch starts from the 2-D sin-cos position table."""
import numpy as np
import torch

from attentionshift_amd import mae_heads, mil_head


def test_sincos_table_matches_its_definition():
    t = mae_heads.sincos_pos_embed_2d(16, 3, cls_token=True)
    assert t.shape == (10, 16) and (t[0] == 0).all()
    omega = 1.0 / 10000 ** (np.arange(4) / 4.0)
    # patch (h=1, w=2) is row 1 + 1*3 + 2: first 8 channels encode w (sin 4, cos 4), last 8 encode h
    row = t[1 + 1 * 3 + 2]
    assert np.allclose(row[:4], np.sin(2 * omega)) and np.allclose(row[4:8], np.cos(2 * omega))
    assert np.allclose(row[8:12], np.sin(1 * omega)) and np.allclose(row[12:], np.cos(1 * omega))


def test_mask_head_starts_from_sincos_and_loads_pretrained_decoder(tmp_path):
    torch.manual_seed(0)
    kw = dict(in_channels=48, embed_dim=64, depth=1, num_heads=2, num_classes=5, img_size=64, patch_size=16)
    head = mae_heads.MAEMaskHeadPointSup(**kw)
    want = torch.from_numpy(mae_heads.sincos_pos_embed_2d(64, 4, cls_token=True)).float()[None]
    assert torch.equal(head.decoder_pos_embed, want) and not head.decoder_pos_embed.requires_grad
    # a MAE pre-training checkpoint: encoder entries must be skipped, decoder entries loaded
    donor = mae_heads.MAEMaskHeadPointSup(**kw)
    with torch.no_grad():
        for p in donor.parameters():
            p.add_(1.0)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    sd.update({"patch_embed.proj.weight": torch.zeros(3), "blocks.0.attn.qkv.weight": torch.zeros(3), "pos_embed": torch.zeros(3)})
    path = str(tmp_path / "mae.pth")
    torch.save({"model": sd}, path)
    loaded = mae_heads.MAEMaskHeadPointSup(init_cfg=dict(type="Pretrained", checkpoint=path), **kw)
    loaded.init_weights()
    for k, v in donor.state_dict().items():
        if not k.startswith("conv_logits"):
            assert torch.equal(loaded.state_dict()[k], v), k
    assert float(loaded.conv_logits.bias.abs().sum()) == 0.0          # always re-initialised


def test_box_and_mil_heads_load_only_when_pretrained_is_set(tmp_path):
    torch.manual_seed(0)
    kw = dict(in_channels=48, embed_dim=64, depth=1, num_heads=2, num_classes=5, img_size=64, patch_size=16)
    donor = mae_heads.MAEBoxHeadRec(**kw)
    with torch.no_grad():
        for p in donor.parameters():
            p.add_(0.5)
    path = str(tmp_path / "mae.pth")
    torch.save({"state_dict": donor.state_dict()}, path)
    a = mae_heads.MAEBoxHeadRec(pretrained=True, **kw)
    a.init_weights(path)
    assert all(torch.equal(a.state_dict()[k], v) for k, v in donor.state_dict().items())
    b = mae_heads.MAEBoxHeadRec(pretrained=False, **kw)
    before = {k: v.clone() for k, v in b.state_dict().items()}
    b.init_weights(path)                                             # pretrained=False in the config: path ignored
    assert all(torch.equal(b.state_dict()[k], v) for k, v in before.items())
    m = mil_head.MAEBoxHeadMIL(in_channels=48, embed_dim=64, num_classes=5, num_layers_query=3, hidden_dim=32, pretrained=True)
    m.init_weights(path)
    assert torch.equal(m.decoder_embed.weight, donor.decoder_embed.weight)
