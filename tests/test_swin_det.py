"""The detection-style Swin backbone (BASELINE config 5; mmdet/models/backbones/swin_transformer.py:448-630): registry
surface and state-dict keys on the CPU, outputs vs the reference's own backbone and the full Swin-B / 1024^2 shapes on
the GPU."""
import numpy as np
import pytest
import torch

import attentionshift_amd as A
from attentionshift_amd import synthetic
from helpers import t


def _cfg(g, dtype=torch.float32):
    return dict(type="SwinTransformer", pretrain_img_size=64, patch_size=4, embed_dim=int(g["cfg_embed_dim"]),
                depths=g["cfg_depths"].tolist(), num_heads=g["cfg_heads"].tolist(), window_size=7, drop_path_rate=0.0,
                ape=bool(g["cfg_ape"]), out_indices=tuple(g["cfg_out_indices"].tolist()), compute_dtype=dtype)


def test_registered_with_the_reference_signature_and_state_dict_keys(golden):
    g = golden("swin_det_r98x118")
    net = A.build_backbone(_cfg(g))
    assert type(net).__name__ == "SwinTransformerDet" and "SwinTransformer" in A.BACKBONES
    assert sorted(net.state_dict().keys()) == sorted(g["state_keys"].tolist())            # incl. norm0..2, absolute_pos_embed
    for name, shp in zip(g["param_names"].tolist(), g["param_shapes"].tolist()):
        assert tuple(net.state_dict()[name].shape) == tuple(int(v) for v in shp.split(",")), name
    frozen = A.build_backbone(dict(_cfg(g), frozen_stages=2))
    assert not any(p.requires_grad for p in frozen.patch_embed.parameters())
    assert not frozen.absolute_pos_embed.requires_grad
    assert not any(p.requires_grad for p in frozen.layers[0].parameters()) and any(p.requires_grad for p in frozen.layers[1].parameters())
    frozen.train()
    assert not frozen.layers[0].training and frozen.layers[1].training                   # train() keeps frozen stages in eval
    with pytest.raises(TypeError):
        net.init_weights(pretrained=3)


def test_drop_path_rates_rise_linearly_over_all_blocks_and_only_act_in_training():
    """mmdet swin_transformer.py:520 + :352: dpr = linspace(0, drop_path_rate, sum(depths)) handed to the blocks in order;
    DropPath scales kept samples by 1 / keep and is the identity in eval()."""
    net = A.build_backbone(dict(type="SwinTransformer", pretrain_img_size=64, embed_dim=32, depths=[2, 2, 6, 2],
                                num_heads=[1, 2, 4, 8], drop_path_rate=0.2))
    rates = [blk.drop_path for layer in net.layers for blk in layer.blocks]
    assert rates[0] == 0.0 and abs(rates[-1] - 0.2) < 1e-12 and all(b > a for a, b in zip(rates, rates[1:]))
    assert abs(rates[5] - 0.2 * 5 / 11) < 1e-12
    blk = net.layers[3].blocks[1]
    x = torch.ones(64, 3, 5)
    blk.eval()
    assert blk._drop_path(x) is x
    blk.train()
    torch.manual_seed(0)
    y = blk._drop_path(x)
    kept = y[:, 0, 0] != 0
    assert 0 < kept.sum() < 64                                            # some samples dropped, some kept
    assert torch.allclose(y[kept], torch.full_like(y[kept], 1.0 / 0.8))   # kept samples are scaled by 1 / keep
    assert (y[~kept] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 6e-2)])
def test_swin_det_backbone_matches_reference(golden, dtype, tol):
    """Non-square 98x118 image: patch padding (to 100x120), window padding in every stage (25x30, 13x15, 7x8 tokens,
    window 7), odd-grid patch merging, absolute position embedding resized bicubically, per-stage output norms."""
    g = golden("swin_det_r98x118")
    net = A.build_backbone(_cfg(g, dtype))
    own = net.state_dict()
    names = [k for k in own if "relative_position_index" not in k]
    net.load_state_dict(synthetic.det_state_dict({k: tuple(own[k].shape) for k in names}), strict=False)
    net = net.cuda().eval()
    hw = g["hw"].tolist()
    x = torch.randn(2, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(int(g["seed"]))).cuda()
    with torch.no_grad():
        outs = net(x)
    assert len(outs) == 3
    for i, o in enumerate(outs):
        ref = t(g[f"out{i}"])
        assert tuple(o.shape) == tuple(ref.shape)
        err = float((o.float().cpu() - ref).abs().max() / ref.abs().max())
        assert err < tol, (i, err)


@pytest.mark.gpu
def test_swin_b_1024_full_backbone_shapes_and_determinism():
    """BASELINE config 5 for real: Swin-B (embed 128, depths 2/2/18/2, heads 4/8/16/32, window 7) at 1024^2, batch 2,
    bf16: stage grids 256/128/64/32 with C = 128/256/512/1024 (stages 2-4 never ran in round 1); two runs bitwise equal."""
    net = A.build_backbone(dict(type="SwinTransformer", embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32],
                                window_size=7, drop_path_rate=0.3, out_indices=(0, 1, 2, 3)))
    torch.manual_seed(0)
    net.init_weights()
    net = net.cuda().eval()
    x = synthetic.images(2, 1024, 1024, seed=1).cuda()
    with torch.no_grad():
        a = net(x)
        b = net(x)
    want = [(2, 128, 256, 256), (2, 256, 128, 128), (2, 512, 64, 64), (2, 1024, 32, 32)]
    assert [tuple(o.shape) for o in a] == want
    for u, v in zip(a, b):
        assert torch.isfinite(u).all() and torch.equal(u, v)
