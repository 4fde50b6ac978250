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
