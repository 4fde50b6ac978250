"""IoU assignment / random sampling of the R-CNN branches.  CPU only."""
import torch

from attentionshift_amd import assign as AS


def test_iou_assign_and_sample():
    gts = torch.tensor([[0., 0., 100., 100.], [200., 200., 300., 280.]])
    props = torch.tensor([[5., 5., 100., 100.], [0., 0., 40., 40.], [210., 205., 300., 280.], [400., 400., 450., 450.],
                          [190., 190., 310., 300.]])
    iou = AS.bbox_overlaps(gts, props)
    assert abs(float(iou[0, 0]) - 0.9025) < 1e-4 and float(iou[1, 3]) == 0.0
    assigned, best = AS.max_iou_assign(props, gts)
    assert assigned.tolist() == [1, 0, 2, 0, 2]                    # IoU(gt1, prop4) = 8000 / 13200 = 0.606
    assert AS.max_iou_assign(props, gts[:0])[0].tolist() == [0] * 5
    gen = torch.Generator().manual_seed(0)
    res = AS.random_sample(props, gts, torch.tensor([3, 7]), assigned, num=6, pos_fraction=0.5, generator=gen)
    assert res.pos_inds.numel() == 3 and res.neg_inds.numel() == 2            # 5 positives (2 gt + 3), capped at 3
    assert (res.pos_gt_labels == torch.tensor([3, 7])[res.pos_assigned_gt_inds]).all()
    assert res.bboxes.shape == (5, 4)
    boxes = torch.cat((gts, props))
    assert torch.equal(res.pos_bboxes, boxes[res.pos_inds]) and torch.equal(res.pos_gt_bboxes, gts[res.pos_assigned_gt_inds])
    none = AS.random_sample(props, gts[:0], torch.zeros(0, dtype=torch.long), torch.zeros(5, dtype=torch.long), num=4)
    assert none.pos_inds.numel() == 0 and none.neg_inds.numel() == 4


def test_assign_and_sample_equal_the_reference_sampler(golden):
    """Fixture = the reference's MaxIoUAssigner.assign + RandomSampler.sample (add_gt_as_proposals) executed under
    torch.manual_seed (tools/gen_golden_sampler.py): the same seed gives the same sampled index sets here."""
    g = golden("sampler")
    for c in range(int(g["n"])):
        t = lambda k: torch.from_numpy(g[f"{k}{c}"])
        props, gts, labels = t("props").reshape(-1, 4), t("gts").reshape(-1, 4), t("labels")
        assigned, _ = AS.max_iou_assign(props, gts, 0.5, 0.5, 0.5, False)
        torch.manual_seed(int(g[f"seed{c}"]))
        res = AS.random_sample(props, gts, labels, assigned, num=int(g[f"num{c}"]), pos_fraction=0.25, add_gt_as_proposals=True)
        assert torch.equal(res.pos_inds, t("pos_inds")) and torch.equal(res.neg_inds, t("neg_inds")), c
        assert torch.equal(res.pos_assigned_gt_inds, t("pos_assigned")) and torch.equal(res.pos_gt_labels, t("pos_gt_labels")), c


def test_sampler_with_a_host_copy_of_the_assignment_draws_the_same_samples():
    """forward_train hands random_sample the host copy of `assigned` it read back with the other shape-deciding tensors:
    same draws, same index sets, and the namespace carries the host-side lists the target builders use."""
    g = torch.Generator().manual_seed(3)
    gts = torch.tensor([[10., 10., 120., 90.], [200., 150., 330., 300.]])
    props = torch.cat((gts[torch.randint(2, (40,), generator=g)] + torch.randn(40, 4, generator=g) * 12,
                       torch.rand(60, 4, generator=g) * 300))
    props[:, 2:] = torch.maximum(props[:, 2:], props[:, :2] + 4)
    labels = torch.tensor([4, 9])
    assigned, _ = AS.max_iou_assign(props, gts, 0.5, 0.5, 0.5, False)
    a = AS.random_sample(props, gts, labels, assigned, num=32, generator=torch.Generator().manual_seed(7))
    b = AS.random_sample(props, gts, labels, assigned, num=32, generator=torch.Generator().manual_seed(7),
                         assigned_host=assigned.float())               # read_back hands fp32 copies
    for k in ("pos_inds", "neg_inds", "pos_assigned_gt_inds", "pos_gt_labels", "bboxes"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(b.pos_inds_host, b.pos_inds) and torch.equal(b.pos_assigned_gt_inds_host, b.pos_assigned_gt_inds)
    assert 0 < a.pos_inds.numel() <= 8 and a.pos_inds.numel() + a.neg_inds.numel() == 32


def test_low_quality_matches_follow_the_reference_order():
    """match_low_quality: every GT's best proposals are re-assigned to it in GT order (a later GT wins a tie), only
    when that best IoU reaches min_pos_iou -- written with selects (no boolean-mask assignment)."""
    gts = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 12.]])
    props = torch.tensor([[0., 0., 10., 11.], [50., 50., 60., 60.], [0., 0., 5., 5.]])
    assigned, best = AS.max_iou_assign(props, gts, pos_iou_thr=0.95, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)
    assert assigned.tolist() == [2, 0, 0]                   # proposal 0 is both GTs' best (IoU 10/11, 11/12): GT 1 last
    assigned, _ = AS.max_iou_assign(props, gts, pos_iou_thr=0.95, neg_iou_thr=0.3, min_pos_iou=0.99, match_low_quality=True)
    assert assigned.tolist() == [-1, 0, 0]
