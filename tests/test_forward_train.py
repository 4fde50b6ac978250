"""AttnShiftRoIHead.forward_train: the host composition of the box / mask / point losses over sampled proposals
(SURVEY 8f-2).  CPU only: the decoder blocks' attention is the HIP small-N kernel in the product, so this test swaps in
torch's SDPA for it (test-side stand-in) and checks the composition against the pieces called by hand."""
import pytest
import torch
import torch.nn.functional as F

import attentionshift_amd as A
from attentionshift_amd import assign as AS, mae_heads, mask_targets as MT


def _torch_attention(self, x):
    B, N, C = x.shape
    q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
    return self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))


def _head(with_mask=True):
    cfg = dict(type="AttnShiftRoIHead",
               bbox_roi_extractor=dict(type="SingleRoIExtractor", featmap_strides=[16],
                                       roi_layer=dict(type="RoIAlign", output_size=7, sampling_ratio=0)),
               bbox_head=dict(type="MAEBoxHeadRec", in_channels=48, embed_dim=64, depth=1, num_heads=2, num_classes=5,
                              with_reconstruct=False, reg_decoded_bbox=True, cam_layer=3,
                              bbox_coder=dict(type="DeltaXYWHBBoxCoder", target_means=[0.] * 4, target_stds=[.1, .1, .2, .2]),
                              loss_bbox=dict(type="GIoULoss", loss_weight=10.0)),
               train_cfg=dict(assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                            match_low_quality=False),
                              sampler=dict(type="RandomSampler", num=16, pos_fraction=0.25, add_gt_as_proposals=True),
                              point_assigner=dict(type="HungarianPointAssigner", cls_cost=dict(weight=1.0),
                                                  reg_cost=dict(weight=10.0)), point_pos_weight=1, pos_weight=-1))
    if with_mask:
        cfg["mask_head"] = dict(type="MAEMaskHeadPointSup", in_channels=48, embed_dim=64, depth=1, num_heads=2,
                                num_classes=5, scale_factor=2, scale_mode="bicubic")
    return A.build_head(cfg)


def _inputs(gen):
    fmap = torch.rand(2, 48, 14, 14, generator=gen)
    gts = [torch.tensor([[20., 30., 120., 150.], [100., 40., 200., 200.]]), torch.tensor([[10., 10., 90., 90.]])]
    labels = [torch.tensor([1, 4]), torch.tensor([2])]
    props = []
    for g in gts:
        jit = g.repeat(6, 1) + (torch.rand(g.shape[0] * 6, 4, generator=gen) - 0.5) * 30
        far = torch.rand(12, 2, generator=gen) * 150
        props.append(torch.cat((jit, torch.cat((far, far + 40), 1))))
    P = 6
    coords = [torch.rand(g.shape[0], P, 2, generator=gen) * 200 for g in gts]
    plabels = [torch.rand(g.shape[0], P, generator=gen) > 0.5 for g in gts]
    centres = [[torch.rand(2, 2, generator=gen) * 200 for _ in range(g.shape[0])] for g in gts]
    T = 4
    return dict(fmap=fmap, gts=gts, labels=labels, props=props, coords=coords, plabels=plabels, centres=centres,
                point_cls=torch.randn(2, T, 5, generator=gen), point_reg=torch.rand(2, T, 2, generator=gen),
                gt_points=[(g[:, :2] + g[:, 2:]) / 2 for g in gts], metas=[dict(img_shape=(224, 224, 3))] * 2)


def test_forward_train_composition(monkeypatch):
    monkeypatch.setattr(mae_heads._Attention, "forward", _torch_attention)
    torch.manual_seed(0)
    head = _head()
    assert isinstance(head.bbox_head, mae_heads.MAEBoxHeadRec) and isinstance(head.mask_head, mae_heads.MAEMaskHeadPointSup)
    assert any(k.startswith("bbox_head.decoder_blocks.0.attn.qkv") for k in head.state_dict())
    assert any(k.startswith("mask_head.conv_logits") for k in head.state_dict())
    d = _inputs(torch.Generator().manual_seed(1))
    with torch.enable_grad():
        fmap = d["fmap"].clone().requires_grad_(True)
        losses = head.forward_train(fmap, d["metas"], d["props"], d["gts"], d["labels"], point_cls=d["point_cls"],
                                    point_reg=d["point_reg"], gt_points=d["gt_points"], gt_points_labels=d["labels"],
                                    mask_point_coords=d["coords"], mask_point_labels=d["plabels"],
                                    semantic_centers_split=d["centres"], generator=torch.Generator().manual_seed(5))
        assert set(losses) == {"loss_point_cls", "loss_point", "pos_point_acc", "loss_cls", "acc", "loss_bbox", "loss_mask"}
        total = sum(v for k, v in losses.items() if k.startswith("loss"))
        assert torch.isfinite(total)
        total.backward()
    assert fmap.grad.abs().sum() > 0
    for name in ("bbox_head.fc_cls.weight", "bbox_head.fc_reg.weight", "mask_head.conv_logits.weight"):
        assert dict(head.named_parameters())[name].grad.abs().sum() > 0, name
    res = head.last_sampling_results
    assert all(r.pos_inds.numel() <= 4 and r.pos_inds.numel() + r.neg_inds.numel() <= 16 for r in res)
    assert all(r.pos_inds.numel() >= g.shape[0] for r, g in zip(res, d["gts"]))         # the GT boxes are proposals too
    # the same sampling, the pieces by hand
    with torch.no_grad():
        rois = torch.cat([torch.cat((torch.full((r.bboxes.shape[0], 1), float(i)), r.bboxes), 1) for i, r in enumerate(res)])
        feats = head._roi_extract(d["fmap"], rois)
        cls, reg, _ = head.bbox_head(feats)
        want = head.bbox_head.loss(cls, reg, rois, *head.bbox_head.get_targets(res))
        assert torch.allclose(want["loss_cls"], losses["loss_cls"], atol=1e-6)
        assert torch.allclose(want["loss_bbox"], losses["loss_bbox"], atol=1e-6)
        n_pos = [r.pos_bboxes.shape[0] for r in res]
        pos = torch.cat([torch.arange(r.bboxes.shape[0]) < n for r, n in zip(res, n_pos)])
        sites, tgt = MT.mask_point_targets([r.pos_bboxes for r in res], [r.pos_assigned_gt_inds for r in res],
                                           d["coords"], d["plabels"], d["centres"])
        lm = MT.point_mask_loss(head.mask_head(feats[pos]), sites, tgt, torch.cat([r.pos_gt_labels for r in res]))
        assert torch.allclose(lm, losses["loss_mask"], atol=1e-6)


def test_forward_train_without_positives_or_mask_head(monkeypatch):
    monkeypatch.setattr(mae_heads._Attention, "forward", _torch_attention)
    torch.manual_seed(0)
    head = _head(with_mask=False)
    d = _inputs(torch.Generator().manual_seed(2))
    empty = [g[:0] for g in d["gts"]]
    with torch.enable_grad():
        losses = head.forward_train(d["fmap"], d["metas"], d["props"], empty, [l[:0] for l in d["labels"]],
                                    point_cls=d["point_cls"], point_reg=d["point_reg"], gt_points=[g[:0, :2] for g in d["gts"]],
                                    gt_points_labels=[l[:0] for l in d["labels"]])
    assert "loss_mask" not in losses and float(losses["loss_bbox"].detach()) == 0.0 and float(losses["loss_cls"].detach()) > 0
    assert all(r.pos_inds.numel() == 0 and r.neg_inds.numel() == 16 for r in head.last_sampling_results)
    bare = A.build_head(dict(type="AttnShiftRoIHead", bbox_head=dict(type="MAEBoxHeadRec", cam_layer=3)))
    with pytest.raises(RuntimeError, match="box head"):
        bare.forward_train(d["fmap"], d["metas"], d["props"], d["gts"], d["labels"])


def test_heads_take_an_empty_roi_batch_on_the_tensor_ops():
    """No positive RoIs (images without objects): the heads run on [0, C, h, w] through the plain modules -- the attention
    kernel is never asked for an empty problem -- the loss is 0 and the parameters still get (zero) gradients."""
    common = dict(in_channels=96, img_size=224, patch_size=16, embed_dim=256, depth=1, num_heads=8, mlp_ratio=4., num_classes=20)
    mask = A.build_head(dict(type="MAEMaskHeadPointSup", scale_factor=2, scale_mode="bicubic", **common)).train()
    box = A.build_head(dict(type="MAEBoxHeadRec", with_reconstruct=False, **common)).train()
    with torch.enable_grad():
        pred = mask(torch.zeros(0, 96, 14, 14))
        assert pred.shape == (0, 20, 28, 28)
        loss = mask.loss(pred, torch.zeros(0, 5), torch.zeros(0, dtype=torch.long))["loss_mask"]
        cls, reg, _ = box(torch.zeros(0, 96, 7, 7))
        assert cls.shape == (0, 21) and reg.shape == (0, 80)
        (loss + cls.sum() + reg.sum()).backward()
    assert float(loss.detach()) == 0.0
    for head in (mask, box):
        assert not [n for n, p in head.named_parameters() if p.requires_grad and p.grad is None and "pos_embed" not in n]
