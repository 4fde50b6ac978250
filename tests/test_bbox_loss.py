"""Box coder / targets / loss of the MAE box head.  CPU only, against definitions and the coder's
documented example (mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:174-186)."""
import torch

from attentionshift_amd import bbox_loss as BL


def test_coder_documented_example_and_round_trip():
    rois = torch.tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]])
    want = torch.tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                         [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    assert torch.allclose(BL.delta2bbox(rois, deltas, max_shape=(32, 32, 3)), want, atol=1e-4)
    gen = torch.Generator().manual_seed(0)
    p = torch.rand(50, 2, generator=gen) * 200
    props = torch.cat((p, p + 20 + torch.rand(50, 2, generator=gen) * 100), 1)
    g = torch.rand(50, 2, generator=gen) * 200
    gts = torch.cat((g, g + 20 + torch.rand(50, 2, generator=gen) * 100), 1)
    stds = (0.1, 0.1, 0.2, 0.2)
    back = BL.delta2bbox(props, BL.bbox2delta(props, gts, stds=stds), stds=stds)
    assert torch.allclose(back, gts, atol=1e-3)
    multi = BL.delta2bbox(props[:3], torch.zeros(3, 8))                     # two classes, zero deltas -> the rois twice
    assert torch.allclose(multi, props[:3].repeat(1, 2))


def test_targets_and_loss_definitions():
    K = 4
    pos = [torch.tensor([[10., 10., 50., 60.], [0., 0., 20., 20.]]), torch.zeros(0, 4)]
    neg = [torch.tensor([[5., 5., 9., 9.]]), torch.tensor([[1., 1., 3., 3.], [2., 2., 8., 8.]])]
    gtb = [torch.tensor([[12., 8., 52., 64.], [0., 0., 22., 18.]]), torch.zeros(0, 4)]
    gtl = [torch.tensor([1, 3]), torch.zeros(0, dtype=torch.long)]
    labels, lw, bt, bw = BL.bbox_targets(pos, neg, gtb, gtl, K)
    assert labels.tolist() == [1, 3, K, K, K] and lw.tolist() == [1.] * 5
    assert torch.allclose(bt[:2], BL.bbox2delta(pos[0], gtb[0], stds=(0.1, 0.1, 0.2, 0.2))) and bt[2:].abs().sum() == 0
    assert bw.sum() == 8
    gen = torch.Generator().manual_seed(3)
    with torch.enable_grad():
        cls = torch.randn(5, K + 1, generator=gen, requires_grad=True)
        reg = torch.randn(5, 4 * K, generator=gen, requires_grad=True)
        out = BL.bbox_head_loss(cls, reg, labels, lw, bt, bw, K)
        want_cls = torch.nn.functional.cross_entropy(cls, labels, reduction="sum") / 5
        sel = torch.stack((reg[0, 4:8], reg[1, 12:16]))
        want_reg = (sel - bt[:2]).abs().sum() / 5
        assert torch.allclose(out["loss_cls"], want_cls) and torch.allclose(out["loss_bbox"], want_reg)
        (out["loss_cls"] + out["loss_bbox"]).backward()
    assert reg.grad[2:].abs().sum() == 0 and reg.grad[0, :4].abs().sum() == 0 and reg.grad[0, 4:8].abs().sum() > 0
    none = BL.bbox_head_loss(cls.detach(), reg.detach(), torch.full((5,), K), lw, bt, bw, K)
    assert float(none["loss_bbox"]) == 0.0
    agn = BL.bbox_head_loss(None, reg.detach()[:, :4], labels, lw, bt, bw, K, reg_class_agnostic=True)
    assert "loss_cls" not in agn and float(agn["loss_bbox"]) > 0


def test_giou_and_decoded_regression():
    a = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.], [0., 0., 10., 10.]])
    b = torch.tensor([[0., 0., 10., 10.], [5., 0., 15., 10.], [20., 0., 30., 10.]])
    # identical: 0; half overlap: IoU 1/3, enclosing 150, union 150 -> 1 - 1/3; disjoint: IoU 0, enclose 300, union 200
    assert torch.allclose(BL.giou_loss(a, b), torch.tensor([0., 2. / 3., 1. + 100. / 300.]), atol=1e-6)
    K = 2
    pos, neg = [a[:2]], [torch.tensor([[50., 50., 60., 60.]])]
    labels, lw, bt, bw = BL.bbox_targets(pos, neg, [b[:2]], [torch.tensor([0, 1])], K, reg_decoded_bbox=True)
    assert torch.equal(bt[:2], b[:2])
    rois = torch.cat((a[:2], neg[0]))
    reg = torch.zeros(3, 4 * K)                                               # zero deltas decode to the rois
    out = BL.bbox_head_loss(None, reg, labels, lw, bt, bw, K, rois=rois, loss_bbox_type="GIoULoss", loss_bbox_weight=10.0)
    assert torch.allclose(out["loss_bbox"], torch.tensor(10.0 * (0.0 + 2.0 / 3.0) / 3), atol=1e-5)
