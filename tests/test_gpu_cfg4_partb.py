"""BASELINE config-4 shape (MAE-ViT-Large at 1280^2: 80 x 80 patches, C = 1024, G = 7 objects per image) parity of the
Part-B kernels against the oracle's VALUES -- the shift kernel's channel split (C = 1024 = 64 k16 steps over 8 waves), the
refinement kernels' 6400-patch aggregation lists and the CAM-box kernels at 1280^2 were held to the oracle only up to
64 x 64 patches / C = 768 before (reference stdroi:1778-1840, 668-707, 50-117).

The oracle here is its matmul form (cos_matrix): the reference's broadcast form would need a [7, 20, 6400, 1024] fp32
temporary per cosine.  As at config-2 size, argmax decisions are checked from the kernel's OWN state of every iteration and
may differ from the matmul-form arithmetic only on rounding coin flips (helpers.check_shift_decisions); everything integer
(boxes, kept-pixel counts) is compared bit for bit.
"""
import pytest
import torch

import attnshift_oracle as O
from helpers import assert_close, assert_equal, check_shift_decisions

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

HP = WP = 80
C, G, LC, P, S = 1024, 7, 7, 20, 5


def dev(x):
    return x.cuda().contiguous()


@pytest.fixture(scope="module")
def ops():
    from attentionshift_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def case():
    from attentionshift_amd import synthetic
    return synthetic.shift_inputs(4321, HP, WP, C, G, LC)


def test_cosine_shift_config4_shape_every_iteration_matches_the_oracle(ops, case):
    """B4: each of the 5 iterations from the kernel's own state against one oracle iteration (argmax on every determined
    decision, prototypes / tau of untouched clusters to 1e-3), then the final similarity map against a direct cosine."""
    feat = case["vit_feat"]
    tok = feat.flatten(1).t().contiguous()
    rois = case["boxes"]
    box_patch = (rois // 16).int()
    inbox = O.box_mask(rois // 16, (HP, WP)).flatten(1)
    feats = tok[None] * inbox[..., None]                                     # [G, Np, C], zero outside each box
    maps = O.box_mask(rois // 16, (HP, WP))                                  # seeds: the grid-strided in-box patches
    coords = O.grid_seed_coords(maps, rois, 0.35, P)
    prot0 = feat.permute(1, 2, 0)[coords[..., 0], coords[..., 1]].contiguous()
    obj_img = torch.zeros(G, dtype=torch.int32)
    run = lambda k, trace=False: ops.cosine_shift(dev(tok[None]), dev(box_patch), dev(obj_img), dev(prot0), k, HP, WP,
                                                  return_trace=trace)
    pout, sim, assign, tau = run(S, True)
    states = [prot0] + [run(k)[0].cpu() for k in range(1, S)] + [pout.cpu()]
    for k in range(S):
        tau_k = 0.1 if k == 0 else tau[k - 1].cpu()[..., None]
        step = O.cosine_shift_step(states[k], feats, tau_k)
        n, near, under = check_shift_decisions(step, assign[k], states[k], feats, tau_k, what=f"config-4 iteration {k}",
                                               max_flips=32)                 # 44 800 decisions per iteration (12 288 at config 2)
        same = torch.ones(G, P, dtype=torch.bool)
        bad = assign[k].long().cpu() != step["win"]
        for gi, ni in zip(*bad.nonzero(as_tuple=True)):
            same[gi, assign[k][gi, ni].long()] = False
            same[gi, step["win"][gi, ni]] = False
        ref_p, got_p = step["prot"][same], states[k + 1][same]
        scale = ref_p.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        assert_close(ref_p / scale, got_p / scale, 0, 1e-3, f"prototypes after iteration {k}")
        assert_close(step["tau"][..., 0][same], tau[k].cpu()[same], 1e-3, 2e-6, f"tau after iteration {k}")
        print(f"[cfg4] iteration {k}: {n} coin-flip patches (near ties {near}, underflow {under}) of {assign[k].numel()}")
    assert_close(O.cos_matrix(pout.cpu(), tok), sim, 1e-3, 1e-5, "final sim == cos(returned prototypes, unmasked features)")
    assert int(assign.min()) >= 0 and int(assign.max()) < P


def test_refine_similarity_config4_shape_matches_the_oracle(ops, case):
    """B2: foreground (G + 1 seeds, selection group) and background (G seeds) refinement, two levels, 6400 patches."""
    feat = case["vit_feat"]
    tok = feat.flatten(1).t().contiguous()
    rois = case["boxes"]
    box_patch = (rois // 16).int()
    g = torch.Generator().manual_seed(77)
    # 20 sampled pixels per seed: inside each object's box for the foreground group (+ one shared background group drawn
    # from the whole image), anywhere for the background seeds -- the shapes get_cosine_similarity_refined_map passes
    def pts_in(box, n):
        x = box[0] + torch.rand(n, generator=g) * (box[2] - box[0])
        y = box[1] + torch.rand(n, generator=g) * (box[3] - box[1])
        return torch.stack((x, y), dim=-1).floor()
    full = torch.tensor([0.0, 0.0, WP * 16 - 1.0, HP * 16 - 1.0])
    pts_fg = torch.stack([pts_in(rois[i], 20) for i in range(G)] + [pts_in(full, 20)])
    pts_bg = torch.stack([pts_in(full, 20) for _ in range(G)])
    tau = 0.9
    for pts, is_sel, what in ((pts_fg, True, "fg"), (pts_bg, False, "bg")):
        seeds = O.seed_features(pts, feat)
        sims, f_out = ops.refine_similarity(dev(tok), dev(seeds), dev(box_patch), G, 2, tau, is_sel, HP, WP)
        o_maps, o_seeds = O.refined_similarity(pts, feat, rois, 2, tau, is_sel)
        assert_close(o_maps.flatten(2), sims, 1e-3, 1e-5, f"patch-grid {what} maps, 3 levels")
        assert_close(o_seeds, f_out, 1e-3, 1e-4, f"refined {what} seeds")


def test_cam_boxes_config4_shape_match_the_oracle(ops, case):
    """B1 at 1280^2: 7 layers x 7 objects; boxes and kept-pixel counts bit for bit, per-map extrema exact."""
    cams = case["cams"]                                                       # [Lc, G, hp, wp]
    pts = case["points"].repeat(LC, 1)
    boxes, status, mm = ops.cam_boxes(dev(cams.reshape(LC * G, HP, WP)), dev(pts), 0.2, 0.5, 16, return_minmax=True)
    ref_boxes, up = O.cam_boxes_from_rollout(cams, case["points"], 0.2, 0.5)
    assert_equal(ref_boxes, boxes.reshape(LC, G, 4).permute(1, 0, 2), "CAM boxes at 1280^2")
    kept = torch.tensor([[int(O.cam_box(up[l, g], case["points"][g], 0.2, 0.5, (HP * 16, WP * 16))[1].sum()) for l in range(LC)]
                         for g in range(G)], dtype=torch.int32)
    assert_equal(kept, status.reshape(LC, G).t().cpu(), "kept-pixel counts")
    flat = up.reshape(LC * G, -1)
    assert_equal(torch.stack((flat.min(1)[0], flat.max(1)[0]), dim=1), mm, "per-map min/max of the upsampled CAMs")
