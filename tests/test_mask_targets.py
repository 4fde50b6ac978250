"""Consumers of the pseudo labels vs fixtures produced by the reference's own functions
(tools/gen_golden_consumers.py), plus the composition / loss semantics.  CPU only."""
import numpy as np
import torch

from attentionshift_amd import mask_targets as MT
from helpers import t


def _case(golden):
    g = golden("consumers")
    coords = [t(g[f"coords{i}"]) for i in range(3)]
    labels = [torch.from_numpy(g[f"labels{i}"]) for i in range(3)]
    centers = [[t(g[f"center{i}_{k}"]) for k in range(len(g[f"ncenters{i}"]))] for i in range(3)]
    return g, coords, labels, centers


def test_update_coords_with_semantic_centers_matches_reference(golden):
    g, coords, labels, centers = _case(golden)
    out_c, out_l = MT.update_coords_with_semantic_centers(coords, labels, centers)
    for i in range(3):
        assert np.array_equal(g[f"out_coords{i}"], out_c[i].numpy()), i
        assert np.array_equal(g[f"out_labels{i}"], out_l[i].numpy()), i
    # structure: negatives first, then the centre slots; EVERY centre slot is labelled True, the padded ones at (-1,-1)
    # too (the reference's `torch.ones(centers_coords.shape[:-1])`) -- they fall outside every box later
    assert out_l[0].sum(1).tolist() == [4, 4, 4] and (out_c[0][1] == -1).all()   # object 1: no negatives, no centres
    assert out_l[1].sum(1).tolist() == [3, 3]


def test_update_coords_with_a_host_copy_of_the_labels_is_the_same(golden):
    """forward_train reads the labels back once with the other shape-deciding tensors (fp32 copies): same result, and an
    image without centres needs no host copy."""
    g, coords, labels, centers = _case(golden)
    host = [None if len(c) == 0 else l.float() for l, c in zip(labels, centers)]
    out_c, out_l = MT.update_coords_with_semantic_centers(coords, labels, centers, labels_host=host)
    for i in range(3):
        assert np.array_equal(g[f"out_coords{i}"], out_c[i].numpy()), i
        assert np.array_equal(g[f"out_labels{i}"], out_l[i].numpy()), i


def test_get_point_coords_wrt_box_matches_reference(golden):
    g = golden("consumers")
    got = MT.get_point_coords_wrt_box(t(g["boxes"]), t(g["pts"]))
    assert np.array_equal(g["pts_wrt_box"], got.numpy())
    assert torch.equal(t(g["pts"]), torch.from_numpy(g["pts"]))          # the input is not modified


def test_point_sample_is_bilinear_on_the_unit_square():
    m = torch.arange(12.).reshape(1, 1, 3, 4)
    pts = torch.tensor([[[0.125, 1 / 6], [0.875, 5 / 6], [0.5, 0.5]]])   # pixel centres (0,0), (3,2); the middle
    got = MT.point_sample(m, pts)
    assert torch.allclose(got[0, 0], torch.tensor([0.0, 11.0, 5.5]), atol=1e-5)


def test_targets_and_loss_literal_vs_intended(golden):
    g, coords, labels, centers = _case(golden)
    pos_bboxes = [torch.tensor([[0., 0., 500., 500.], [100., 100., 200., 200.]]), torch.tensor([[0., 0., 500., 500.]]),
                  torch.zeros(0, 4)]
    assigned = [torch.tensor([0, 2]), torch.tensor([1]), torch.zeros(0, dtype=torch.long)]
    sites, tg = MT.mask_point_targets(pos_bboxes, assigned, coords, labels, centers, literal=False)
    assert sites.shape == (3, 10, 2) and tg.dtype == torch.long
    inside = ((sites >= 0) & (sites <= 1)).all(-1)
    assert ((tg == 2) == ~inside).all() and (tg[0] != 2).sum() > 0        # padding (-1,-1) and outside points are ignored
    sites_l, tg_l = MT.mask_point_targets(pos_bboxes, assigned, coords, labels, centers, literal=True)
    assert tg_l.dtype == torch.bool and torch.equal(sites, sites_l)
    assert torch.equal(tg_l, (tg == 1) | (tg == 2))                       # the reference's `bool[...] = 2` stores True
    gen = torch.Generator().manual_seed(1)
    pred = torch.randn(3, 5, 14, 14, generator=gen, requires_grad=True)
    cls = torch.tensor([1, 4, 0])
    with torch.enable_grad():                                             # (the suite runs with grad disabled)
        loss = MT.point_mask_loss(pred, sites, tg, cls)
        logits = MT.point_sample(pred, sites)[torch.arange(3), cls]
        want = (torch.nn.functional.binary_cross_entropy_with_logits(logits, (tg == 1).float(), reduction="none") * (tg != 2)).sum() / tg.numel()
        assert torch.allclose(loss, want, atol=1e-6)                      # ignored points weigh 0, mean over ALL points
        loss.backward()
    assert pred.grad is not None and torch.isfinite(pred.grad).all()
    assert MT.point_mask_loss(pred[:0], sites[:0], tg[:0], cls[:0]).item() == 0.0
    assert MT.point_mask_loss(pred, sites_l, tg_l, cls) > 0               # literal targets: nothing ignored
