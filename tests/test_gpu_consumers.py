"""SURVEY 8f-1 on the DEVICE path: the consumers of the pseudo labels (box targets / losses, the point-supervised mask
loss, the point-token loss incl. its Hungarian targets, the mask-point targets) evaluated on CUDA tensors against the
fixtures produced by executing the reference's own functions (tools/gen_golden_head_losses.py,
gen_golden_point_loss.py, gen_golden_consumers.py) -- the GPU twins of tests/test_head_losses_golden.py,
tests/test_point_loss.py and tests/test_mask_targets.py."""
import types

import numpy as np
import pytest
import torch

import attentionshift_amd as A
import attnshift_oracle as O
from attentionshift_amd import mae_heads, mask_targets as MT, point_loss as PL

pytestmark = pytest.mark.gpu


def test_box_targets_and_losses_equal_the_reference_on_the_device(golden):
    g = golden("head_losses")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    K = int(g["K"])
    res = [types.SimpleNamespace(pos_bboxes=t(f"pos_bboxes{i}").reshape(-1, 4), neg_bboxes=t(f"neg_bboxes{i}").reshape(-1, 4),
                                 pos_gt_bboxes=t(f"pos_gt_bboxes{i}").reshape(-1, 4), pos_gt_labels=t(f"pos_gt_labels{i}"))
           for i in range(int(g["n_img"]))]
    for tag, decoded, loss_cfg in (("giou", True, dict(type="GIoULoss", loss_weight=10.0)), ("l1", False, dict(type="L1Loss", loss_weight=1.0))):
        head = A.build_head(dict(type="MAEBoxHeadRec", in_channels=32, embed_dim=32, depth=1, num_heads=1, num_classes=K,
                                 with_reconstruct=False, reg_decoded_bbox=decoded, loss_bbox=loss_cfg,
                                 bbox_coder=dict(type="DeltaXYWHBBoxCoder", target_means=[0.] * 4, target_stds=[.1, .1, .2, .2]))).cuda()
        targets = head.get_targets(res)
        for got, name in zip(targets, ("labels", "label_weights", "bbox_targets", "bbox_weights")):
            want = t(f"{tag}_{name}")
            assert got.is_cuda and got.shape == want.shape and torch.allclose(got.float(), want.float(), atol=1e-5), (tag, name)
        out = head.loss(t("cls_score"), t("bbox_pred"), t("rois"), *targets)
        for k in ("loss_cls", "acc", "loss_bbox"):
            assert abs(float(out[k]) - float(g[f"{tag}_{k}"][0])) <= 5e-5 * max(1.0, abs(float(g[f"{tag}_{k}"][0]))), (tag, k)


def test_mask_point_loss_equals_the_reference_on_the_device(golden):
    g = golden("head_losses")
    t = lambda k: torch.from_numpy(g[k]).cuda()
    head = mae_heads.MAEMaskHeadPointSup(num_classes=int(g["K"]), in_channels=32, embed_dim=32, depth=1, num_heads=1).cuda()
    for kind in ("bool", "long"):
        got = head.loss(t("mask_pred"), t(f"mask_tgt_{kind}"), t("mask_labels"))["loss_mask"]
        assert got.is_cuda and abs(float(got) - float(g[f"mask_loss_{kind}"][0])) < 2e-6, kind
    assert float(head.loss(t("mask_pred")[:0], t("mask_tgt_long")[:0], t("mask_labels")[:0])["loss_mask"]) == 0.0


def test_point_token_loss_equals_the_reference_roi_head_loss_on_the_device(golden):
    g = golden("point_loss")
    for c in range(int(g["n"])):
        cls, reg = torch.from_numpy(g[f"cls{c}"]).cuda(), torch.from_numpy(g[f"reg{c}"]).cuda()
        pts = [torch.from_numpy(g[f"pts{c}_{i}"]).reshape(-1, 2).cuda() for i in range(2)]
        labels = [torch.from_numpy(g[f"labels{c}_{i}"]).cuda() for i in range(2)]
        shapes = [tuple(int(v) for v in s) for s in g[f"shapes{c}"]]
        tl, tw, _, _ = PL.point_targets(cls, reg, pts, labels, shapes, 20, point_pos_weight=1, cls_cost=1.0, reg_cost=10.0)
        assert torch.equal(tl.cpu(), torch.from_numpy(g[f"labels_all{c}"])) and torch.equal(tw.cpu(), torch.from_numpy(g[f"label_w{c}"]))
        out = PL.point_token_loss(cls, reg, pts, labels, shapes, num_classes=20, loss_point_weight=10.0, loss_cls_weight=1.0,
                                  cls_cost=1.0, reg_cost=10.0)
        want_cls = float(g[f"loss_point_cls{c}"][0])
        if np.isfinite(want_cls):
            assert abs(float(out["loss_point_cls"]) - want_cls) <= 2e-5 * abs(want_cls), c
            assert abs(float(out["loss_point"]) - float(g[f"loss_point{c}"][0])) <= 2e-5, c
            assert abs(float(out["pos_point_acc"]) - float(g[f"pos_point_acc{c}"][0])) <= 1e-4, c


def test_mask_point_targets_match_the_reference_on_the_device(golden):
    g = golden("consumers")
    t = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    coords = [t(g[f"coords{i}"]) for i in range(3)]
    labels = [t(g[f"labels{i}"]) for i in range(3)]
    centers = [[t(g[f"center{i}_{k}"]) for k in range(len(g[f"ncenters{i}"]))] for i in range(3)]
    out_c, out_l = MT.update_coords_with_semantic_centers(coords, labels, centers)
    for i in range(3):
        assert out_c[i].is_cuda and np.array_equal(g[f"out_coords{i}"], out_c[i].cpu().numpy()), i
        assert np.array_equal(g[f"out_labels{i}"], out_l[i].cpu().numpy()), i
    got = MT.get_point_coords_wrt_box(t(g["boxes"]), t(g["pts"]))
    assert np.array_equal(g["pts_wrt_box"], got.cpu().numpy())


@pytest.mark.parametrize("out,C,hw", [(7, 48, (14, 17)), (14, 768, (64, 64))])
def test_roi_align_hip_matches_the_tensor_op_restatement(out, C, hw):
    """as_roi_align_fwd / _bwd (csrc/roi_align.hip) vs mil_head's tensor-op RoIAlign (the restatement of mmcv's adaptive,
    aligned, average-pooled RoIAlign that the CPU tests hold to its definition): forward 1e-5, backward (the atomic-free
    per-pixel gather, fixed summation order) 1e-4 of the gradient range; boxes that leave the map, a degenerate box (empty sample grid -> 0) and several images."""
    from attentionshift_amd.mil_head import _roi_align_chunk, roi_align
    gen = torch.Generator().manual_seed(9)
    H, W = hw
    feat = torch.randn(2, C, H, W, generator=gen)
    n = 40
    xy = torch.rand(n, 2, generator=gen) * torch.tensor([W * 16.0, H * 16.0]) - 20
    rois = torch.cat((torch.randint(0, 2, (n, 1), generator=gen).float(), xy, xy + 8 + torch.rand(n, 2, generator=gen) * 300), 1)
    rois[5, 3:] = rois[5, 1:3] - 3.0
    wgt = torch.randn(n, C, out, out, generator=gen)
    with torch.enable_grad():
        f_ref = feat.clone().requires_grad_(True)
        y_ref = _roi_align_chunk(f_ref, rois, out, 1.0 / 16, 0, True)
        (y_ref * wgt).sum().backward()
        f_hip = feat.cuda().requires_grad_(True)
        y_hip = roi_align(f_hip, rois.cuda(), out, 1.0 / 16, 0, True)
        (y_hip * wgt.cuda()).sum().backward()
    assert y_hip.shape == y_ref.shape and (y_hip[5] == 0).all()
    assert float((y_hip.cpu() - y_ref).abs().max()) <= 1e-5 * max(1.0, float(y_ref.abs().max()))
    gscale = float(f_ref.grad.abs().max())
    assert float((f_hip.grad.cpu() - f_ref.grad).abs().max()) <= 1e-4 * gscale


@pytest.mark.parametrize("out,C,hw,n", [(7, 768, (64, 64), 48), (14, 768, (64, 64), 32), (7, 48, (14, 17), 40)])
def test_roi_align_hip_matches_the_independent_mmcv_oracle(out, C, hw, n):
    """as_roi_align_fwd / as_roi_align_bwd through the C ABI against oracle.roi_align_mmcv / roi_align_mmcv_backward --
    the scalar-loop restatement of mmcv-full 1.3.8's published kernel that shares no arithmetic with the product
    (tests/test_oracle_roi_nms.py holds it to closed-form cases).  Config-2 map (64x64x768) at 7x7 and 14x14
    (configs/mae/attnshift_voc12aug.py:64-68,123-127), RoIs leaving the map, a degenerate RoI, two images."""
    from attentionshift_amd import ops
    gen = torch.Generator().manual_seed(11 + out)
    H, W = hw
    feat = torch.randn(2, C, H, W, generator=gen)
    xy = torch.rand(n, 2, generator=gen) * torch.tensor([W * 16.0, H * 16.0]) - 24
    rois = torch.cat((torch.randint(0, 2, (n, 1), generator=gen).float(), xy, xy + 8 + torch.rand(n, 2, generator=gen) * 380), 1)
    rois[5, 3:] = rois[5, 1:3] - 3.0                        # empty sample grid -> zeros
    rois[6] = torch.tensor([1.0, -40.0, 100.0, 90.0, 400.0])          # hangs off the left border
    wgt = torch.randn(n, C, out, out, generator=gen)
    nhwc = feat.permute(0, 2, 3, 1).contiguous().cuda()
    y = ops.roi_align_fwd(nhwc, rois.cuda(), out, 1.0 / 16, 0, True)                       # [R, out*out, C]
    d = ops.roi_align_bwd(wgt.permute(0, 2, 3, 1).reshape(n, out * out, C).contiguous().cuda(), rois.cuda(),
                          tuple(nhwc.shape), out, 1.0 / 16, 0, True)                      # [B, H, W, C]
    y_ref = O.roi_align_mmcv(feat.numpy(), rois.numpy(), out, 1.0 / 16, 0, True)
    d_ref = O.roi_align_mmcv_backward(wgt.numpy(), rois.numpy(), tuple(feat.shape), 1.0 / 16, 0, True)
    got = y.view(n, out, out, C).permute(0, 3, 1, 2).cpu().double().numpy()
    assert (got[5] == 0).all()
    assert np.abs(got - y_ref).max() <= 1e-5 * max(1.0, np.abs(y_ref).max())
    gd = d.permute(0, 3, 1, 2).cpu().double().numpy()
    assert np.abs(gd - d_ref).max() <= 1e-4 * np.abs(d_ref).max()


def test_roi_align_hip_closed_forms():
    """Constant map -> the constant; linear ramp -> the ramp at every bin centre (bilinear interpolation is exact on a
    linear function); fully outside -> zeros.  fp32 kernel: 1e-5 of the range."""
    from attentionshift_amd import ops
    B, C, H, W = 1, 8, 64, 64
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    a = torch.arange(1, C + 1, dtype=torch.float32)
    ramp = (0.5 * ys[..., None] - 0.25 * xs[..., None] + 1.0) * a                      # [H, W, C]
    rois = torch.tensor([[0, 40., 56., 600., 480.], [0, 100.5, 33.25, 717.0, 890.75], [0, 2000., 2000., 2100., 2100.]])
    for out in (7, 14):
        y = ops.roi_align_fwd(ramp[None].contiguous().cuda(), rois.cuda(), out, 1.0 / 16, 0, True).cpu().view(3, out, out, C)
        for n in range(2):
            x1, y1, x2, y2 = (rois[n, 1:] / 16.0 - 0.5).tolist()
            cy = y1 + (torch.arange(out) + 0.5) * (y2 - y1) / out
            cx = x1 + (torch.arange(out) + 0.5) * (x2 - x1) / out
            want = (0.5 * cy[:, None, None] - 0.25 * cx[None, :, None] + 1.0) * a
            assert float((y[n] - want).abs().max()) <= 1e-5 * float(want.abs().max()), (out, n)
        assert (y[2] == 0).all()
        const = ops.roi_align_fwd(torch.full((1, H, W, C), 2.5).cuda(), rois[:2].cuda(), out, 1.0 / 16, 0, True)
        assert float((const - 2.5).abs().max()) <= 1e-5


def test_nms_on_device_tensors_matches_the_oracle():
    """inference.multiclass_nms fed CUDA tensors (the test-time path of stdroi:3192-3221) against the oracle's per-class
    greedy NMS."""
    from attentionshift_amd import inference as I
    gen = torch.Generator().manual_seed(2)
    n, K = 300, 20
    xy = torch.rand(n, 2, generator=gen) * 900
    boxes = torch.cat((xy, xy + 10 + torch.rand(n, 2, generator=gen) * 200), 1)
    per_class = (boxes[:, None, :] + torch.randn(n, K, 4, generator=gen) * 4).reshape(n, 4 * K)
    scores = torch.softmax(torch.randn(n, K + 1, generator=gen) * 2.5, 1)
    d, l = I.multiclass_nms(per_class.cuda(), scores.cuda(), 0.05, 0.5, 100)
    dr, lr = O.multiclass_nms_mmdet(per_class.numpy(), scores.numpy(), 0.05, 0.5, 100)
    assert l.cpu().tolist() == lr.tolist()
    assert np.allclose(d.cpu().numpy(), dr, rtol=1e-6, atol=1e-5)


def test_fused_decoder_block_node_equals_the_per_op_bridges(monkeypatch):
    """autograd.DecoderBlockFn (one autograd node per MAE-decoder block) against the per-op bridges it replaces
    (AddLayerNormFn + LinearFn + SmallAttnFn, AS_HEAD_BLOCKS_UNFUSED=1): same kernels in the same order, so the output and
    every gradient -- input, LayerNorm affine, the four Linear weights and biases of each of the 3 blocks -- are bitwise
    equal, in a bf16 autocast region as the training step runs the heads."""
    from attentionshift_amd.mae_heads import DecoderBlock, _run_blocks
    torch.manual_seed(5)
    blocks = torch.nn.ModuleList([DecoderBlock(256, 8) for _ in range(3)]).cuda()
    for p in blocks.parameters():
        torch.nn.init.normal_(p, std=0.05)
    x0 = torch.randn(37, 50, 256, device="cuda")
    w = torch.randn(37, 50, 256, device="cuda")

    def run(unfused):
        if unfused:
            monkeypatch.setenv("AS_HEAD_BLOCKS_UNFUSED", "1")
        else:
            monkeypatch.delenv("AS_HEAD_BLOCKS_UNFUSED", raising=False)
        for p in blocks.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            y = _run_blocks(blocks, x)
            (y.float() * w).sum().backward()
        return y.detach(), x.grad, [p.grad.clone() for p in blocks.parameters()]

    y_u, dx_u, g_u = run(True)
    y_f, dx_f, g_f = run(False)
    assert torch.equal(y_u, y_f) and torch.equal(dx_u, dx_f)
    for (n, _), a, b in zip(blocks.named_parameters(), g_u, g_f):
        assert a.dtype == b.dtype and torch.equal(a, b), n
