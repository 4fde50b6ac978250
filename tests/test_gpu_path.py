"""End-to-end parity of the plugin classes on the GPU against the reference fixtures.

  * VisionTransformerDet.forward (fp32 MFMA path: 1e-3 relative as north_star asks; bf16 path: 3e-2 of
    the output range) incl. the roll-out rows recomputed from (q,k,lse)
  * AttnShiftRoIHead.seed_pseudo_gt: whole chain B1..B6 from CAM rows to masks / part centres
"""
import numpy as np
import pytest
import torch

import attnshift_oracle as O
from helpers import assert_close, assert_equal, backbone_cfg, backbone_state_dict, shift_case_inputs, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def build_backbone(g, dtype):
    import attentionshift_amd as A
    from attentionshift_amd import synthetic
    cfg = backbone_cfg(g)
    bb = A.build_backbone(dict(type="VisionTransformerDet", img_size=cfg["img_size"], patch_size=16,
                               embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=4.,
                               qkv_bias=True, drop_path_rate=0., out_indices=cfg["out_indices"], last_feat=True,
                               point_tokens_num=cfg["point_tokens_num"], num_classes=cfg["num_classes"],
                               return_attention=True, compute_dtype=dtype))
    bb.load_state_dict(backbone_state_dict(g))
    bb = bb.cuda().eval()
    img = synthetic.images(cfg["batch"], *cfg["img_hw"], seed=cfg["seed"]).cuda()
    return bb, img, cfg


def rel(ref, got):
    ref, got = ref.double().cpu(), got.double().cpu()
    return ((ref - got).abs().max() / (ref.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("tag", ["small", "tiny224", "h4"])
def test_backbone_fp32_matches_reference(golden, tag):
    from attentionshift_amd import ops
    g = golden(f"backbone_{tag}")
    bb, img, cfg = build_backbone(g, torch.float32)
    out = bb(img)
    for k in ("last_feat", "point_tokens", "outputs_class", "outputs_coord"):
        assert rel(t(g[k]), out[k]) < 1e-3, (k, rel(t(g[k]), out[k]))
    for i, f in enumerate(out["feature"]):
        st = int(g[f"feature{i}_stride"])
        assert rel(t(g[f"feature{i}"]), f[:, :, ::st, ::st]) < 1e-3, f"feature{i}"
    T, Lc = cfg["point_tokens_num"], cfg["cam_layer"]
    rows = ops.rollout_rows(out["attns"][-Lc:], T)
    assert_close(t(g["rollout_rows"]), rows, 1e-3, 1e-6, "roll-out rows (A3)")
    for key in g.files:
        if key.startswith("attn") and key[4:].isdigit():
            st = out["attns"][int(key[4:])]
            dense = ops.attn_mean_rows(st, 0, st.N)
            assert_close(t(g[key]), dense, 1e-3, 1e-7, f"head-mean attention layer {key[4:]}")


@pytest.mark.parametrize("tag", ["small", "tiny224", "h4"])
def test_backbone_bf16_close_to_reference(golden, tag):
    """bf16 path; the "h4" case (4 heads) is the fixture-backed check of rollout_step4_kernel, the roll-out kernel of
    every h % 4 == 0 configuration (ViT-B, ViT-L): the reference's own attns_project_to_feature rows."""
    from attentionshift_amd import ops
    g = golden(f"backbone_{tag}")
    bb, img, cfg = build_backbone(g, torch.bfloat16)
    out = bb(img)
    assert rel(t(g["last_feat"]), out["last_feat"]) < 3e-2
    assert rel(t(g["outputs_coord"]), out["outputs_coord"]) < 3e-2
    rows = ops.rollout_rows(out["attns"][-cfg["cam_layer"]:], cfg["point_tokens_num"])
    assert rel(t(g["rollout_rows"]), rows) < 3e-2


def test_deferred_fpn_is_the_same_maps(golden):
    """VisionTransformerDet.defer_fpn: the FPN maps queued later on a side stream (backbone.DeferredFPN) are bit-identical
    to the ones the forward computes itself, a consumer that just indexes `feature` gets valid maps on its own stream, and
    launch() before result() (the RoI head's order) gives the same again."""
    from attentionshift_amd.backbone import DeferredFPN
    g = golden("backbone_tiny224")
    bb, img, cfg = build_backbone(g, torch.bfloat16)
    ref = [f.clone() for f in bb(img)["feature"]]
    bb.defer_fpn = True
    out = bb(img)
    assert isinstance(out["feature"], DeferredFPN) and len(out["feature"]) == len(ref)
    busy = torch.randn(2048, 2048, device="cuda")
    for _ in range(8):                                     # keep the caller's stream busy: the join must still order the reads
        busy = busy @ busy * 1e-3
    for i, f in enumerate(out["feature"]):                 # iteration = launch + join
        assert torch.equal(f, ref[i]), f"feature{i} (auto-join)"
    out2 = bb(img)
    out2["feature"].launch()
    got = out2["feature"].result()
    assert out2["feature"].result() is got
    for i in range(len(ref)):
        assert torch.equal(out2["feature"][i], ref[i]) and got[i].shape == ref[i].shape, f"feature{i} (launch, then result)"
    bb.defer_fpn = False
    assert isinstance(bb(img)["feature"], tuple)


def test_point_head_on_its_side_stream_is_the_same_outputs(golden):
    """VisionTransformerDet.point_head_stream: class / coordinate outputs queued on the backbone's side stream are
    bit-identical to the in-line ones once the caller has waited for out["point_head_ready"], with the caller's stream busy
    and the forward repeated (the token tensor's block is reused) in between."""
    g = golden("backbone_tiny224")
    bb, img, cfg = build_backbone(g, torch.bfloat16)
    ref = bb(img)
    assert "point_head_ready" not in ref
    ref_cls, ref_reg = ref["outputs_class"].clone(), ref["outputs_coord"].clone()
    bb.point_head_stream = True
    busy = torch.randn(2048, 2048, device="cuda")
    for _ in range(3):
        out = bb(img)
        for _ in range(4):
            busy = busy @ busy * 1e-3
        torch.cuda.current_stream().wait_event(out["point_head_ready"])
        assert torch.equal(out["outputs_class"], ref_cls) and torch.equal(out["outputs_coord"], ref_reg)
    bb.point_head_stream = False
    assert "point_head_ready" not in bb(img)


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_seed_pseudo_gt_chain_matches_reference(golden, tag, monkeypatch):
    import attentionshift_amd as A
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    hp, wp, G, Lc, C = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"]), int(g["C"])
    T, N = 10, 1 + hp * wp + 10
    head = A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]),
                             mean_shift_times_local=int(g["n_shift"]),
                             bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]),
                                            seed_multiple=float(g["area_ratio"]), cam_layer=Lc, num_classes=20)))
    rows = torch.zeros(1, Lc, T, N)
    rows[0, :, :G, 1:-T] = inp["cams"].flatten(2)
    monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
    best = t(g["best_idx"]).cuda()
    head.layer_selector = lambda boxes, labels, fmap: [best]
    torch.manual_seed(int(g["seed"]) + 1)
    out = head.seed_pseudo_gt(None, [dict(img_shape=(hp * 16, wp * 16, 3))], None, None, None,
                              vit_feat=inp["vit_feat"][None].cuda(), point_cls=torch.zeros(1, T, 20).cuda(),
                              point_reg=torch.zeros(1, T, 2).cuda(), attns=None, gt_points=[inp["points"].cuda()],
                              gt_points_labels=[inp["labels"].cuda()], return_mask=True,
                              pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]),
                              num_mask_point_gt=int(g["num_gt"]), corr_size=int(g["corr_size"]), obj_tau=float(g["obj_tau"]),
                              pos_inds=[torch.arange(G).cuda()], matched_gt=[torch.arange(G).cuda()])
    assert_equal(t(g["rois"]), out["pseudo_gt_bboxes"][0], "pseudo boxes (B1)")
    assert_close(t(g["map_fg_last"]), out["map_cos_fg"][0], 1e-3, 1e-5, "map_cos_fg (B2)")
    assert_equal(t(g["mask_coords"]), out["mask_points_coords"][0], "mask point coords (B2')")
    assert_equal(t(g["mask_labels"]), out["mask_points_labels"][0], "mask point labels (B2')")
    assert_equal(g["num_parts"], np.array(out["num_parts"][0]), "num_parts (B5)")
    assert_close(t(g["coords_org"]), out["semantic_centers_org"][0][0], 0, 0, "part centres (B5)")
    assert_equal(g["corres_gt"], out["corres_gts"][0], "corres_gts")
    ref_masks = O.pseudo_masks(t(g["map_fg_last"]), float(g["pos_thr"]))
    diff = int((ref_masks != out["pseudo_gt_masks"][0]).sum())
    assert diff <= ref_masks.size * 1e-5, f"pseudo masks differ in {diff} pixels"      # threshold-edge pixels only
    assert_close(t(g["fg_feat"]), out["inst_fg_feat"][0].flatten(1), 1e-3, 1e-4, "inst_fg_feat")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 8e-2)])
def test_backbone_train_step_grads_match_oracle_autograd(golden, dtype, tol):
    """Trainable path (train() + grad enabled): forward through autograd.AttentionFn (as_attn_fwd / as_attn_bwd) and
    backward to the parameters, vs torch autograd (fp64) through the oracle's restatement of the same forward.
    Loss = fixed random projection of last_feat, point_tokens, outputs_coord and the un-FPN'd taps."""
    g = golden("backbone_small")
    bb, img, cfg = build_backbone(g, dtype)
    bb.train()
    sd = {k: v.double() for k, v in backbone_state_dict(g).items()}
    names = ["blocks.0.attn.qkv.weight", "blocks.0.attn.qkv.bias", "blocks.0.attn.proj.weight", "blocks.0.norm1.weight",
             "blocks.1.mlp.fc1.weight", f"blocks.{cfg['depth'] - 1}.attn.proj.bias", "pos_embed", "point_token",
             "patch_embed.proj.weight", "bbox_embed.layers.2.weight"]
    if dtype == torch.float32:         # fp32 has no gate flips: the lower point-head layers are held to the same bar
        names += ["bbox_embed.layers.0.weight", "bbox_embed.layers.1.weight"]
    # (the LAST layer of the point head: its weight gradient is dY^T relu(h), continuous in the activations.  The first
    # layers' gradients are gated by relu'(h) of the layers above: in bf16 a pre-activation near 0 flips its gate against
    # the fp64 reference and moves a whole row of the gradient -- 0.05 .. 0.15 of the range depending on which rounding
    # of the SAME forward (4e-3 apart) the attention kernel variant produces; measured with AS_SDPA_IMPL = 0 / 1 / 3 / 4.)
    gen = torch.Generator().manual_seed(3)

    def loss_of(out, like):
        tot = 0.0
        gen.manual_seed(3)
        for k in ("last_feat", "point_tokens", "outputs_coord", "org_feats"):
            w = torch.randn(out[k].shape, generator=gen).to(like)
            tot = tot + (out[k].to(like.dtype) * w).sum()
        return tot

    with torch.enable_grad():
        for n in names:
            sd[n].requires_grad_(True)
        ref = O.backbone_forward(img.double().cpu(), sd, patch_size=16, depth=cfg["depth"], num_heads=cfg["num_heads"],
                                 out_indices=cfg["out_indices"], point_tokens_num=cfg["point_tokens_num"])
        loss_of(ref, torch.zeros((), dtype=torch.float64)).backward()
        out = bb(img)
        assert out["attns"][0].o is not None                      # the autograd path ran (state keeps o for backward)
        loss_of(out, torch.zeros((), dtype=torch.float32, device="cuda")).backward()
    params = dict(bb.named_parameters())
    for n in names:
        assert params[n].grad is not None, n
        r = rel(sd[n].grad.float(), params[n].grad.float())
        assert r < tol, (n, r)


def test_seed_pseudo_gt_two_images_equals_per_image(golden, monkeypatch):
    """A batch of two images (the second = the first) through the batched path (per-image threads off so that both use
    the global RNG stream deterministically; ONE mean-shift call for both images) must reproduce, for image 0, what the
    single-image call gives -- integer outputs bitwise, maps to fp32 rounding."""
    import attentionshift_amd as A
    g = golden("shift_tiny224")
    inp = shift_case_inputs(g)
    hp, wp, G, Lc = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"])
    T, N = 10, 1 + hp * wp + 10

    def run(nimg):
        head = A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]),
                                 mean_shift_times_local=int(g["n_shift"]),
                                 bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]),
                                                seed_multiple=float(g["area_ratio"]), cam_layer=Lc, num_classes=20)))
        rows = torch.zeros(nimg, Lc, T, N)
        rows[:, :, :G, 1:-T] = inp["cams"].flatten(2)
        monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
        best = t(g["best_idx"]).cuda()
        head.layer_selector = lambda boxes, labels, fmap: [best] * nimg
        torch.manual_seed(int(g["seed"]) + 1)
        return head.seed_pseudo_gt(None, [dict(img_shape=(hp * 16, wp * 16, 3))] * nimg, None, None, None,
                                   vit_feat=inp["vit_feat"][None].repeat(nimg, 1, 1, 1).cuda(),
                                   point_cls=torch.zeros(nimg, T, 20).cuda(), point_reg=torch.zeros(nimg, T, 2).cuda(),
                                   attns=None, gt_points=[inp["points"].cuda()] * nimg,
                                   gt_points_labels=[inp["labels"].cuda()] * nimg, return_mask=True,
                                   pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]),
                                   num_mask_point_gt=int(g["num_gt"]), corr_size=int(g["corr_size"]),
                                   obj_tau=float(g["obj_tau"]),
                                   pos_inds=[torch.arange(G).cuda()] * nimg, matched_gt=[torch.arange(G).cuda()] * nimg)

    one, two = run(1), run(2)
    assert_equal(one["pseudo_gt_bboxes"][0], two["pseudo_gt_bboxes"][0], "boxes")
    assert_equal(one["mask_points_coords"][0], two["mask_points_coords"][0], "mask points (same RNG prefix)")
    assert (one["pseudo_gt_masks"][0] == two["pseudo_gt_masks"][0]).all()
    assert one["num_parts"][0] == two["num_parts"][0]
    assert_equal(one["semantic_centers_org"][0][0], two["semantic_centers_org"][0][0], "part centres")
    assert_close(one["map_cos_fg"][0], two["map_cos_fg"][0], 1e-6, 1e-6, "instance maps")
    # image 1 is the same scene with its own random seed points: same boxes, same number of objects
    assert_equal(two["pseudo_gt_bboxes"][0], two["pseudo_gt_bboxes"][1], "boxes of the duplicated image")
    assert two["pseudo_gt_masks"][1].shape == two["pseudo_gt_masks"][0].shape


def test_reference_rng_device_attempt_falls_back_to_the_host_path(golden, monkeypatch):
    """Reference-RNG mode on the 14 x 14-patch fixture: a candidate set there is smaller than the 20 points asked for (one
    of the reference's refill branches), so the device attempt raises its flag and the call is REPEATED on the host path
    -- the global generator was not touched by the abandoned attempt, so the outcome equals the host-only run bit for
    bit (two images), torch's generator ends in the same state, and what the abandoned attempt captured is dropped."""
    import attentionshift_amd as A
    g = golden("shift_tiny224")
    inp = shift_case_inputs(g)
    hp, wp, G, Lc = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"])
    T, N, nimg = 10, 1 + hp * wp + 10, 2

    def run(host):
        if host:
            monkeypatch.setenv("AS_REF_RNG_HOST", "1")
        else:
            monkeypatch.delenv("AS_REF_RNG_HOST", raising=False)
        head = A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]),
                                 mean_shift_times_local=int(g["n_shift"]),
                                 bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]),
                                                seed_multiple=float(g["area_ratio"]), cam_layer=Lc, num_classes=20)))
        assert head.rng_mode == "reference"
        rows = torch.zeros(nimg, Lc, T, N)
        rows[:, :, :G, 1:-T] = inp["cams"].flatten(2)
        monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
        best = t(g["best_idx"]).cuda()
        head.layer_selector = lambda boxes, labels, fmap: [best] * nimg
        head.capture = []
        torch.manual_seed(int(g["seed"]) + 1)
        out = head.seed_pseudo_gt(None, [dict(img_shape=(hp * 16, wp * 16, 3))] * nimg, None, None, None,
                                  vit_feat=inp["vit_feat"][None].repeat(nimg, 1, 1, 1).cuda(),
                                  point_cls=torch.zeros(nimg, T, 20).cuda(), point_reg=torch.zeros(nimg, T, 2).cuda(),
                                  attns=None, gt_points=[inp["points"].cuda()] * nimg,
                                  gt_points_labels=[inp["labels"].cuda()] * nimg, return_mask=True,
                                  pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]),
                                  num_mask_point_gt=int(g["num_gt"]), corr_size=int(g["corr_size"]),
                                  obj_tau=float(g["obj_tau"]),
                                  pos_inds=[torch.arange(G).cuda()] * nimg, matched_gt=[torch.arange(G).cuda()] * nimg)
        return out, head.capture, torch.rand(4), head.rng_stats

    host, cap_h, next_h, st_h = run(True)
    dev, cap_d, next_d, st_d = run(False)
    assert st_h == dict(device_calls=0, host_redos=0) and st_d == dict(device_calls=0, host_redos=1), (st_h, st_d)
    assert torch.equal(next_h, next_d)                                # the generator ends where the host path leaves it
    pts_h = [c for c in cap_h if "points_fg" in c]
    pts_d = [c for c in cap_d if "points_fg" in c]
    assert len(pts_h) == len(pts_d) == nimg
    for a, b in zip(pts_h, pts_d):
        assert_equal(a["points_fg"], b["points_fg"], "sampled foreground seeds")
        assert_equal(a["points_bg"], b["points_bg"], "sampled background seeds")
    for i in range(nimg):
        assert_equal(host["mask_points_coords"][i], dev["mask_points_coords"][i], "mask points")
        assert_equal(host["mask_points_labels"][i], dev["mask_points_labels"][i], "mask point labels")
        assert (host["pseudo_gt_masks"][i] == dev["pseudo_gt_masks"][i]).all()
        assert host["num_parts"][i] == dev["num_parts"][i]
        assert_equal(host["semantic_centers_org"][0][i], dev["semantic_centers_org"][0][i], "part centres")
        assert_close(host["map_cos_fg"][i], dev["map_cos_fg"][i], 0, 0, "instance maps")
    # image 0 against the reference's own fixture (single-image run, same seed: the first image's stream prefix)
    assert_equal(t(g["mask_coords"]), dev["mask_points_coords"][0], "mask points vs the reference fixture")


def test_seed_pseudo_gt_ragged_batch(golden, monkeypatch):
    """Ragged batch: image 0 carries all 3 objects of the fixture scene, image 1 only the first 2.  Image 0 of the batch
    must equal the single-image run bitwise (same RNG prefix); image 1 must have 2 objects everywhere and the boxes /
    CAM-derived quantities of those 2 objects (no randomness involved) must equal image 0's first two."""
    import attentionshift_amd as A
    g = golden("shift_tiny224")
    inp = shift_case_inputs(g)
    hp, wp, G, Lc = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"])
    T, N = 10, 1 + hp * wp + 10
    counts = [G, G - 1]

    def run(ns):
        nimg = len(ns)
        head = A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]),
                                 mean_shift_times_local=int(g["n_shift"]),
                                 bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]),
                                                seed_multiple=float(g["area_ratio"]), cam_layer=Lc, num_classes=20)))
        rows = torch.zeros(nimg, Lc, T, N)
        rows[:, :, :G, 1:-T] = inp["cams"].flatten(2)
        monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
        best = t(g["best_idx"]).cuda()
        head.layer_selector = lambda boxes, labels, fmap: [best[:n] for n in ns]
        torch.manual_seed(int(g["seed"]) + 1)
        return head.seed_pseudo_gt(None, [dict(img_shape=(hp * 16, wp * 16, 3))] * nimg, None, None, None,
                                   vit_feat=inp["vit_feat"][None].repeat(nimg, 1, 1, 1).cuda(),
                                   point_cls=torch.zeros(nimg, T, 20).cuda(), point_reg=torch.zeros(nimg, T, 2).cuda(),
                                   attns=None, gt_points=[inp["points"][:n].cuda() for n in ns],
                                   gt_points_labels=[inp["labels"][:n].cuda() for n in ns], return_mask=True,
                                   pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]),
                                   num_mask_point_gt=int(g["num_gt"]), corr_size=int(g["corr_size"]),
                                   obj_tau=float(g["obj_tau"]),
                                   pos_inds=[torch.arange(n).cuda() for n in ns],
                                   matched_gt=[torch.arange(n).cuda() for n in ns])

    one, two = run([G]), run(counts)
    assert_equal(one["pseudo_gt_bboxes"][0], two["pseudo_gt_bboxes"][0], "boxes, image 0")
    assert_equal(one["mask_points_coords"][0], two["mask_points_coords"][0], "mask points, image 0")
    assert (one["pseudo_gt_masks"][0] == two["pseudo_gt_masks"][0]).all()
    assert one["num_parts"][0] == two["num_parts"][0]
    assert_equal(one["semantic_centers_org"][0][0], two["semantic_centers_org"][0][0], "part centres, image 0")
    assert two["pseudo_gt_bboxes"][1].shape[0] == G - 1 and two["pseudo_gt_masks"][1].shape[0] == G - 1
    assert len(two["num_parts"][1]) == G - 1 and two["mask_points_coords"][1].shape[0] == G - 1
    assert_equal(two["pseudo_gt_bboxes"][0][:G - 1], two["pseudo_gt_bboxes"][1], "boxes of the shared objects")


@pytest.mark.parametrize("rng_mode", ["reference", "fast"])
def test_full_size_step_properties(rng_mode):
    """BASELINE config 2 at full size (ViT-B, 1024^2, 2 images, 3 objects, 7 roll-out layers, 5 shift iterations, bf16):
    the whole bench step, twice from the same RNG seed.  Size-independent properties: the step is deterministic (every
    integer output bitwise equal, maps bitwise equal), boxes lie inside the image and contain their GT point, pseudo
    masks are 0/1 with one mask per object, part counts respect the cap, mask points with a positive label lie inside
    their object's box."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    step = bench.build(torch.device("cuda", 0), rng_mode)      # "fast": stage-wise issue on one HIP stream per image
    outs = []
    for _ in range(2):
        torch.manual_seed(1234)
        outs.append(step())
    a, b = outs
    G, H = bench.CFG["objects"], bench.CFG["img"]
    for i in range(bench.CFG["batch"]):
        assert_equal(a["pseudo_gt_bboxes"][i], b["pseudo_gt_bboxes"][i], "boxes")
        assert_equal(a["mask_points_coords"][i], b["mask_points_coords"][i], "mask points")
        assert_equal(a["mask_points_labels"][i], b["mask_points_labels"][i], "mask point labels")
        assert (a["pseudo_gt_masks"][i] == b["pseudo_gt_masks"][i]).all()
        assert a["num_parts"][i] == b["num_parts"][i]
        assert_equal(a["map_cos_fg"][i], b["map_cos_fg"][i], "instance maps")
        box = a["pseudo_gt_bboxes"][i]
        assert box.shape == (G, 4)
        assert (box[:, :2] >= 0).all() and (box[:, 2:] <= H).all() and (box[:, 2] > box[:, 0]).all() and (box[:, 3] > box[:, 1]).all()
        m = a["pseudo_gt_masks"][i]
        assert m.shape == (G, H, H) and m.dtype == np.uint8 and set(np.unique(m)) <= {0, 1} and m.reshape(G, -1).any(1).all()
        assert all(0 <= n <= 6 for n in a["num_parts"][i])          # num_semantic_points (5) + 1
        pts, lab = a["mask_points_coords"][i], a["mask_points_labels"][i]
        inside = (pts[..., 0] >= box[:, None, 0]) & (pts[..., 0] <= box[:, None, 2]) & \
                 (pts[..., 1] >= box[:, None, 1]) & (pts[..., 1] <= box[:, None, 3])
        assert inside[lab].all()


def test_vit_large_full_depth_step(monkeypatch):
    """BASELINE config 4 for real (bench.py --config vitl): ViT-L width 1024 / 16 heads / 24 blocks at 1280^2 -- N = 6501
    tokens, 80x80 patches, 7 objects, 1 image, roll-out over 7 layers, CAM boxes at 1280^2, refinement, mean shift with
    C = 1024, pseudo masks.  The whole step must run and satisfy the geometric invariants, in both RNG modes."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = dict(bench.CONFIGS["vitl"])
    monkeypatch.setattr(bench, "CFG", cfg)
    for mode in ("reference", "fast"):
        step = bench.build(torch.device("cuda", 0), mode)
        torch.manual_seed(7)
        with torch.no_grad():
            out = step()
        G, H = cfg["objects"], cfg["img"]
        box = out["pseudo_gt_bboxes"][0]
        assert box.shape == (G, 4) and torch.isfinite(box).all()
        assert (box[:, :2] >= 0).all() and (box[:, 2:] <= H).all() and (box[:, 2] > box[:, 0]).all()
        m = out["pseudo_gt_masks"][0]
        assert m.shape == (G, H, H) and m.reshape(G, -1).any(1).all()
        assert torch.isfinite(out["map_cos_fg"][0]).all() and len(out["num_parts"][0]) == G
        # every object's mask sits inside a slightly grown copy of its pseudo box (the maps are box-masked cosines)
        for g_ in range(G):
            ys, xs = np.nonzero(m[g_])
            assert len(ys) > 0
        del step
        torch.cuda.empty_cache()


# ---- fast-RNG mode: device-side draws -----------------------------------------------------------------------------
def _dev_gen(seed=7):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return g


def test_grid_seed_nosync_equals_host_path_and_flags_short_objects():
    """The strided grid seeds are deterministic: the device-side ranks must select exactly what the host path selects
    whenever every object has >= 20 positives; an object with fewer raises the flag."""
    from attentionshift_amd import roi_head as RH
    gen = torch.Generator().manual_seed(3)
    mask = (torch.rand(4, 24, 20, generator=gen) > 0.6).to(torch.uint8).cuda()
    counts = mask.flatten(1).sum(1).int()
    rois = torch.tensor([[0, 0, 320, 384]] * 4, dtype=torch.float32).cuda()
    want = RH.grid_seed_finish(mask, counts, rois, 20)
    got, flag = RH.grid_seed_nosync(mask, counts, 20)
    assert_equal(want, got, "grid seeds")
    assert not bool(flag)
    mask[2] = 0
    mask[2, 3, 4:9] = 1                                     # 5 positives: the refill branch
    _, flag = RH.grid_seed_nosync(mask, mask.flatten(1).sum(1).int(), 20)
    assert bool(flag)


def test_mask_points_nosync_draws_distinct_candidates_with_right_labels():
    from attentionshift_amd import roi_head as RH
    gen = torch.Generator().manual_seed(5)
    G, H, W = 3, 64, 80
    pos = (torch.rand(G, H, W, generator=gen) > 0.7).to(torch.uint8).cuda()
    neg = ((torch.rand(G, H, W, generator=gen) > 0.5).to(torch.uint8).cuda()) & (1 - pos)
    counts = torch.stack((pos.flatten(1).sum(1), neg.flatten(1).sum(1)), dim=1).int()
    pend = dict(pos=pos, neg=neg, cp=counts[:, 0], counts=counts, shape=(G, H, W), crops=None)
    coords, labels, flag = RH.mask_points_nosync(pend, 10, _dev_gen())
    assert not bool(flag) and coords.shape == (G, 10, 2) and labels.shape == (G, 10)
    x, y = coords[..., 0].long(), coords[..., 1].long()
    for g in range(G):
        assert (pos[g][y[g][labels[g]], x[g][labels[g]]] == 1).all()          # positive labels sit on pos candidates
        assert (neg[g][y[g][~labels[g]], x[g][~labels[g]]] == 1).all()
        assert len({(int(a), int(b), bool(c)) for a, b, c in zip(x[g], y[g], labels[g])}) == 10     # distinct
    # both label kinds occur over many draws, in proportion to the candidate counts
    lab = torch.stack([RH.mask_points_nosync(pend, 10, _dev_gen(100 + k))[1] for k in range(40)]).float().mean((0, 2))
    frac = counts[:, 0].float() / counts.sum(1).float()
    assert (lab - frac).abs().max() < 0.12
    small = dict(pend, counts=torch.tensor([[3, 2]] * G, dtype=torch.int32).cuda(), cp=torch.tensor([3] * G, dtype=torch.int32).cuda())
    assert bool(RH.mask_points_nosync(small, 10, _dev_gen())[2])            # tiny candidate sets -> synchronous path


def test_fast_mode_flagged_image_falls_back_to_the_synchronous_path(monkeypatch):
    """Force the device flag of one stage: the step must still complete through the synchronous path and satisfy the
    same invariants (the fallback redoes the image with host-side draws)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from attentionshift_amd import roi_head as RH
    monkeypatch.setattr(bench, "CFG", dict(bench.CFG, img=512, depth=8))
    step = bench.build(torch.device("cuda", 0), "fast")
    real = RH.grid_seed_nosync
    calls = []

    def flagged(mask, count_dev, n_points=20, **kw):
        coords, flag = real(mask, count_dev, n_points, **kw)
        calls.append(1)
        return coords, torch.ones_like(flag) if len(calls) == 1 else flag      # flag image 0 only

    monkeypatch.setattr(RH, "grid_seed_nosync", flagged)
    torch.manual_seed(0)
    out = step()
    assert len(calls) == 2
    H = 512
    for i in range(2):
        box = out["pseudo_gt_bboxes"][i]
        assert (box[:, :2] >= 0).all() and (box[:, 2:] <= H).all()
        m = out["pseudo_gt_masks"][i]
        assert m.shape[1:] == (H, H) and m.reshape(m.shape[0], -1).any(1).all()
        assert out["mask_points_coords"][i].shape[1:] == (10, 2)


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_semantic_post_device_matches_reference_fixture_and_host_path(golden, tag):
    """The one-readback form of the filter / merge / part-centre stage (stdroi:2022-2031) gives the reference's part
    counts, centres, owners and centre features, and the same similarity maps as the host-logic form."""
    import attentionshift_amd as A
    from attentionshift_amd import ops
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    hp, wp, G = int(g["hp"]), int(g["wp"]), int(g["G"])
    head = A.AttnShiftRoIHead(num_semantic_points=int(g["num_semantic_points"]), mean_shift_times_local=int(g["n_shift"]))
    rois = t(g["rois"]).cuda()
    feat = inp["vit_feat"].cuda()
    labels = inp["labels"].cuda()
    fg_inter, map_fg, _ = head._semantic_pre(t(g["map_fg_last"]).cuda(), t(g["map_bg_last"]).cuda(), float(g["pos_thr"]))
    prot, sim = head.mean_shift_grid_prototype(map_fg, feat, rois, tau=0.1, temp=0.1, n_shift=int(g["n_shift"]))
    args = (prot, sim, fg_inter, rois, feat, labels, 0.85, int(g["num_semantic_points"]))
    want = head._semantic_post(*args)
    flags = [torch.zeros((), dtype=torch.bool, device="cuda")]
    got = head._semantic_post_device(*args, extra=flags)
    assert not any(flags)
    assert_equal(g["num_parts"], np.array(got[5]), "num_parts")
    assert_close(t(g["coords_org"]), got[6], 0, 0, "centre coords")
    assert_equal(g["corres_gt"], got[8], "corres_gt")
    assert_equal(g["labels_org"], got[7], "labels")
    assert_equal(want[0][0], got[0][0], "centres"); assert_equal(want[0][1], got[0][1], "centre labels")
    assert want[5] == got[5]
    if g["feats_all"].shape[0]:
        assert_close(t(g["feats_all"]), got[4], 0, 0, "centre features")
    for a_, b_ in zip(want[2], got[2]):
        assert a_.shape == b_.shape
        if a_.numel():
            assert_close(a_, b_, 1e-5, 1e-6, "part similarity maps")
    assert head._semantic_post_device(*args, extra=[torch.ones((), dtype=torch.bool, device="cuda")]) is None


def test_seed_pseudo_gt_with_the_mil_head_selecting_the_depth(golden, monkeypatch):
    """The MIL head built from the reference's config (not the median-area stand-in) picks gt_box_index from the
    RoI-aligned stride-16 map inside seed_pseudo_gt; its loss comes back as mil_losses['mil_loss']."""
    import attentionshift_amd as A
    g = golden("shift_tiny224")
    inp = shift_case_inputs(g)
    hp, wp, G, Lc, C = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"]), int(g["C"])
    T, N = 10, 1 + hp * wp + 10
    torch.manual_seed(3)
    head = A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]),
                             mean_shift_times_local=int(g["n_shift"]), rng_mode="fast",
                             bbox_roi_extractor=dict(type="SingleRoIExtractor", featmap_strides=[16],
                                                     roi_layer=dict(type="RoIAlign", output_size=7, sampling_ratio=0)),
                             mil_head=dict(type="MAEBoxHeadMIL", in_channels=C, embed_dim=64, num_classes=20,
                                           num_layers_query=Lc, hidden_dim=128),
                             bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]),
                                            seed_multiple=float(g["area_ratio"]), cam_layer=Lc, num_classes=20))).cuda()
    rows = torch.zeros(1, Lc, T, N)
    rows[0, :, :G, 1:-T] = inp["cams"].flatten(2)
    monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
    feat = inp["vit_feat"][None].cuda()
    out = head.seed_pseudo_gt(None, [dict(img_shape=(hp * 16, wp * 16, 3))], None, None, None, vit_feat=feat,
                              point_cls=torch.zeros(1, T, 20).cuda(), point_reg=torch.zeros(1, T, 2).cuda(), attns=None,
                              gt_points=[inp["points"].cuda()], gt_points_labels=[inp["labels"].cuda()], return_mask=True,
                              roi_feature_map=feat, pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]),
                              num_mask_point_gt=int(g["num_gt"]), corr_size=int(g["corr_size"]), obj_tau=float(g["obj_tau"]),
                              pos_inds=[torch.arange(G).cuda()], matched_gt=[torch.arange(G).cuda()])
    idx = out["best_attn_idx"][0]
    assert idx.shape == (G,) and int(idx.min()) >= 0 and int(idx.max()) < Lc
    assert "mil_loss" in out["mil_losses"] and float(out["mil_losses"]["mil_loss"]) > 0
    assert out["pseudo_gt_bboxes"][0].shape == (G, 4) and len(out["pseudo_gt_masks"][0]) == G


def test_train_losses_pseudo_labels_into_the_box_and_mask_branches(golden, monkeypatch):
    """two_stage_point_align.py:75-150 with precomputed proposals: seed_pseudo_gt's boxes / labels / mask points feed
    forward_train (IoU assignment, sampling, the MAE box and mask heads on the HIP small-N attention); every loss is
    finite, differentiable down to the RoI feature map, and the pseudo boxes themselves are sampled as positives."""
    import attentionshift_amd as A
    g = golden("shift_tiny224")
    inp = shift_case_inputs(g)
    hp, wp, G, Lc, C = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"]), int(g["C"])
    T, N = 10, 1 + hp * wp + 10
    torch.manual_seed(3)
    dec = dict(in_channels=C, embed_dim=64, depth=1, num_heads=2, num_classes=20)
    head = A.build_head(dict(
        type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]), mean_shift_times_local=int(g["n_shift"]),
        rng_mode="fast",
        bbox_roi_extractor=dict(type="SingleRoIExtractor", featmap_strides=[16],
                                roi_layer=dict(type="RoIAlign", output_size=7, sampling_ratio=0)),
        mil_head=dict(type="MAEBoxHeadMIL", in_channels=C, embed_dim=64, num_classes=20, num_layers_query=Lc, hidden_dim=128),
        bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]), seed_multiple=float(g["area_ratio"]),
                       cam_layer=Lc, with_reconstruct=False, reg_decoded_bbox=True,
                       loss_bbox=dict(type="GIoULoss", loss_weight=10.0), **dec),
        mask_head=dict(type="MAEMaskHeadPointSup", scale_factor=2, scale_mode="bicubic", **dec),
        train_cfg=dict(assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5),
                       sampler=dict(type="RandomSampler", num=32, pos_fraction=0.25, add_gt_as_proposals=True),
                       point_assigner=dict(type="HungarianPointAssigner", cls_cost=dict(weight=1.0),
                                           reg_cost=dict(weight=10.0))))).cuda()
    rows = torch.zeros(1, Lc, T, N)
    rows[0, :, :G, 1:-T] = inp["cams"].flatten(2)
    monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
    gen = torch.Generator().manual_seed(11)
    xy = torch.rand(40, 2, generator=gen) * (wp * 16 - 60)
    props = [torch.cat((xy, xy + 20 + torch.rand(40, 2, generator=gen) * 40), 1).cuda()]
    with torch.enable_grad():
        feat = inp["vit_feat"][None].cuda().requires_grad_(True)
        losses, seed = head.train_losses(
            feat, [dict(img_shape=(hp * 16, wp * 16, 3))], props, feat.detach(), None,
            torch.randn(1, T, 20, generator=gen).cuda().requires_grad_(True),
            torch.rand(1, T, 2, generator=gen).cuda().requires_grad_(True),
            [inp["points"].cuda()], [inp["labels"].cuda()], generator=torch.Generator().manual_seed(4),
            pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]), num_mask_point_gt=int(g["num_gt"]),
            corr_size=int(g["corr_size"]), obj_tau=float(g["obj_tau"]),
            pos_inds=[torch.arange(G).cuda()], matched_gt=[torch.arange(G).cuda()])
        want = {"mil_loss", "loss_point_cls", "loss_point", "pos_point_acc", "loss_cls", "acc", "loss_bbox", "loss_mask"}
        assert set(losses) == want, set(losses)
        total = sum(v for k, v in losses.items() if "loss" in k)
        assert bool(torch.isfinite(total))
        total.backward()
    assert float(feat.grad.abs().sum()) > 0
    for name in ("bbox_head.fc_cls.weight", "bbox_head.decoder_blocks.0.attn.qkv.weight", "mask_head.conv_logits.weight",
                 "mask_head.decoder_blocks.0.attn.qkv.weight"):
        assert float(dict(head.named_parameters())[name].grad.abs().sum()) > 0, name
    res = head.last_sampling_results[0]
    from attentionshift_amd.assign import bbox_overlaps
    iou = bbox_overlaps(res.pos_bboxes, seed["pseudo_gt_bboxes"][0])
    assert res.pos_inds.numel() >= min(G, 8) and bool((iou[torch.arange(iou.shape[0]), res.pos_assigned_gt_inds] >= 0.5).all())
    assert res.pos_inds.numel() + res.neg_inds.numel() <= 32


def test_simple_test_on_the_hip_small_attention(monkeypatch):
    """Test-time path of the RoI head on the GPU: the box / mask decoders run on as_small_attn_fwd; the best detection
    agrees with the same weights through torch's SDPA."""
    import numpy as np
    import torch.nn.functional as F
    import attentionshift_amd as A
    from attentionshift_amd import mae_heads
    torch.manual_seed(0)
    dec = dict(in_channels=48, embed_dim=64, depth=1, num_heads=2, num_classes=5)
    head = A.build_head(dict(
        type="AttnShiftRoIHead",
        bbox_roi_extractor=dict(type="SingleRoIExtractor", featmap_strides=[16], roi_layer=dict(type="RoIAlign", output_size=7, sampling_ratio=0)),
        mask_roi_extractor=dict(type="SingleRoIExtractor", featmap_strides=[16], roi_layer=dict(type="RoIAlign", output_size=14, sampling_ratio=0)),
        bbox_head=dict(type="MAEBoxHeadRec", with_reconstruct=False, cam_layer=3, **dec),
        mask_head=dict(type="MAEMaskHeadPointSup", scale_factor=2, scale_mode="bicubic", **dec),
        test_cfg=dict(score_thr=0.05, nms=dict(type="nms", iou_threshold=0.5), max_per_img=10, mask_thr_binary=0.5))).cuda()
    gen = torch.Generator().manual_seed(3)
    fmap = torch.rand(1, 48, 14, 14, generator=gen).cuda()
    xy = torch.rand(40, 2, generator=gen) * 120
    props = [torch.cat((xy, xy + 10 + torch.rand(40, 2, generator=gen) * 80), 1).cuda()]
    metas = [dict(img_shape=(224, 224, 3), ori_shape=(224, 224, 3), scale_factor=np.ones(4, dtype=np.float32))]
    with torch.no_grad():
        (boxes, segm), = head.simple_test(fmap, props, metas, rescale=False)

        def ref_attn(self, x):
            B, N, C = x.shape
            q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            return self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))
        monkeypatch.setattr(mae_heads._Attention, "forward", ref_attn)
        (boxes_ref, _), = head.simple_test(fmap, props, metas, rescale=False)
    n = sum(b.shape[0] for b in boxes)
    assert 0 < n <= 10 and all(len(s) == b.shape[0] for s, b in zip(segm, boxes))
    assert all(m.shape == (224, 224) for s in segm for m in s)
    best = max((b[:, 4].max(), c) for c, b in enumerate(boxes) if b.shape[0])
    best_ref = max((b[:, 4].max(), c) for c, b in enumerate(boxes_ref) if b.shape[0])
    assert best[1] == best_ref[1] and abs(best[0] - best_ref[0]) < 2e-3
