"""Point-token loss: targets and losses against their definitions.  CPU only."""
import torch

from attentionshift_amd import point_loss as PL


def test_targets_follow_the_hungarian_match_and_background_convention():
    gen = torch.Generator().manual_seed(4)
    B, T, K = 2, 12, 5
    cls = torch.randn(B, T, K, generator=gen)
    reg = torch.rand(B, T, 2, generator=gen)
    gt_pts = [torch.tensor([[30., 40.], [200., 100.]]), torch.tensor([[10., 10.]])]
    gt_lab = [torch.tensor([1, 3]), torch.tensor([4])]
    reg[0, 7] = torch.tensor([30 / 320, 40 / 240]); reg[0, 2] = torch.tensor([200 / 320, 100 / 240]); reg[1, 5] = torch.tensor([10 / 320, 10 / 240])
    shapes = [(240, 320, 3)] * 2
    labels, lw, tgt, tw = PL.point_targets(cls, reg, gt_pts, gt_lab, shapes, K)
    labels = labels.reshape(B, T)
    assert labels[0, 7] == 1 and labels[0, 2] == 3 and labels[1, 5] == 4 and (labels == K).sum() == B * T - 3
    assert torch.all(lw == 1) and tw.reshape(B, T, 2)[0, 7].tolist() == [1, 1] and tw.sum() == 6
    assert tgt.reshape(B, T, 2)[0, 2].tolist() == [200., 100.]


def test_losses_match_their_definitions_and_backpropagate():
    gen = torch.Generator().manual_seed(6)
    B, T, K = 2, 10, 4
    gt_pts = [torch.tensor([[50., 60.]]), torch.tensor([[100., 20.], [5., 200.]])]
    gt_lab = [torch.tensor([2]), torch.tensor([0, 3])]
    shapes = [(224, 224, 3)] * 2
    with torch.enable_grad():
        cls = torch.randn(B, T, K, generator=gen, requires_grad=True)
        reg = torch.rand(B, T, 2, generator=gen, requires_grad=True)
        out = PL.point_token_loss(cls, reg, gt_pts, gt_lab, shapes, num_classes=K)
        labels, lw, tgt, tw = PL.point_targets(cls, reg, gt_pts, gt_lab, shapes, K)
        pos = labels < K
        npos = float(pos.sum())
        assert npos == 3
        onehot = torch.zeros(B * T, K)
        onehot[pos, labels[pos]] = 1
        p = cls.reshape(-1, K).sigmoid()
        bce = -(onehot * torch.log(p) + (1 - onehot) * torch.log(1 - p))
        fw = (0.25 * onehot + 0.75 * (1 - onehot)) * ((1 - p) * onehot + p * (1 - onehot)) ** 2
        assert torch.allclose(out["loss_point_cls"], (bce * fw).sum() / npos, atol=1e-6)
        want_l1 = 10.0 * (reg.reshape(-1, 2)[pos] - tgt[pos] / 224.0).abs().sum() / npos
        assert torch.allclose(out["loss_point"], want_l1, atol=1e-6)
        assert 0.0 <= float(out["pos_point_acc"]) <= 100.0
        (out["loss_point_cls"] + out["loss_point"]).backward()
    assert torch.isfinite(cls.grad).all() and reg.grad.abs().sum() > 0
    none = PL.point_token_loss(cls.detach(), reg.detach(), [torch.zeros(0, 2)] * 2, [torch.zeros(0, dtype=torch.long)] * 2, shapes, K)
    assert float(none["loss_point"]) == 0.0 and torch.isfinite(none["loss_point_cls"])


def test_point_token_loss_equals_the_reference_roi_head_loss(golden):
    """Fixture = the reference RoI head's own get_targets + loss (stdroi:3284-3514) with its Hungarian assigner, FocalLoss
    (python path) and L1Loss modules executed on two-image batches (tools/gen_golden_point_loss.py)."""
    import numpy as np
    from attentionshift_amd import point_loss as PL
    g = golden("point_loss")
    for c in range(int(g["n"])):
        cls, reg = torch.from_numpy(g[f"cls{c}"]), torch.from_numpy(g[f"reg{c}"])
        pts = [torch.from_numpy(g[f"pts{c}_{i}"]).reshape(-1, 2) for i in range(2)]
        labels = [torch.from_numpy(g[f"labels{c}_{i}"]) for i in range(2)]
        shapes = [tuple(int(v) for v in s) for s in g[f"shapes{c}"]]
        tl, tw, _, _ = PL.point_targets(cls, reg, pts, labels, shapes, 20, point_pos_weight=1, cls_cost=1.0, reg_cost=10.0)
        assert torch.equal(tl, torch.from_numpy(g[f"labels_all{c}"])) and torch.equal(tw, torch.from_numpy(g[f"label_w{c}"]))
        out = PL.point_token_loss(cls, reg, pts, labels, shapes, num_classes=20, loss_point_weight=10.0, loss_cls_weight=1.0,
                                  cls_cost=1.0, reg_cost=10.0)
        want_cls = float(g[f"loss_point_cls{c}"][0])
        if np.isfinite(want_cls):
            assert abs(float(out["loss_point_cls"]) - want_cls) <= 1e-5 * abs(want_cls), c
            assert abs(float(out["loss_point"]) - float(g[f"loss_point{c}"][0])) <= 1e-5, c
            assert abs(float(out["pos_point_acc"]) - float(g[f"pos_point_acc{c}"][0])) <= 1e-4, c
        else:
            # a batch without any object: the reference divides by avg_factor = 0 (inf); this build clamps the divisor
            # (such batches never reach the loss: the dataset filters images without annotations)
            assert bool(torch.isfinite(out["loss_point_cls"])) and float(out["loss_point"]) == 0.0
