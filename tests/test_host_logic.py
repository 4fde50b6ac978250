"""Host-side logic of the plugin (registry, config schema, matching, sampling, part selection) on CPU.

The HIP ops are replaced IN THE TEST by oracle-backed stand-ins (monkeypatch) so the torch glue around
them is pinned to the reference fixtures without a GPU.  The product itself never imports the oracle.
"""
import os

import numpy as np
import pytest
import torch

import attnshift_oracle as O
from helpers import assert_close, assert_equal, shift_case_inputs, t

import attentionshift_amd as A
from attentionshift_amd import roi_head as RH

torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_surface_and_both_head_names():
    assert "VisionTransformerDet" in A.BACKBONES
    assert A.HEADS.get("AttnShiftRoIHead") is A.HEADS.get("StandardRoIHeadMaskPointSampleDeformAttnReppoints")
    head = A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=5, mean_shift_times_local=10,
                             bbox_head=dict(type="MAEBoxHeadRec", seed_thr=0.2, seed_multiple=0.5, cam_layer=7),
                             mil_head=dict(type="MAEBoxHeadMIL", num_layers_query=7)))
    assert head.bbox_head.cam_layer == 7 and head.num_semantic_points == 5 and head.with_mil
    with pytest.raises(KeyError):
        A.build_head(dict(type="NoSuchHead"))
    with pytest.raises(KeyError):
        A.HEADS.register_module(name="AttnShiftRoIHead", module=A.AttnShiftRoIHead)


def test_config_schema_roundtrip(tmp_path):
    base = tmp_path / "base.py"
    base.write_text("model = dict(type='FasterRCNNPointSupAlign', backbone=dict(type='VisionTransformerDet', img_size=224, "
                    "patch_size=16, embed_dim=192, depth=12, num_heads=3), roi_head=dict(type='AttnShiftRoIHead', "
                    "mean_shift_times_local=10, bbox_head=dict(cam_layer=7)))\noptimizer = dict(type='AdamW', lr=1e-4)\n")
    child = tmp_path / "child.py"
    child.write_text("_base_ = ['base.py']\nmodel = dict(backbone=dict(embed_dim=768, num_heads=12), "
                     "roi_head=dict(bbox_head=dict(_delete_=True, cam_layer=5)))\n")
    cfg = A.Config.fromfile(str(child))
    assert cfg.model.backbone.embed_dim == 768 and cfg.model.backbone.depth == 12
    assert cfg.model.roi_head.bbox_head == dict(cam_layer=5)
    cfg.merge_from_dict({"model.roi_head.mean_shift_times_local": 5})
    assert cfg.model.roi_head.mean_shift_times_local == 5
    bb = A.build_backbone(dict(cfg.model.backbone, mlp_ratio=4., qkv_bias=True, last_feat=True, return_attention=True,
                               compute_dtype=torch.float32))
    assert bb.embed_dim == 768 and len(bb.blocks) == 12


def test_backbone_state_dict_keys_match_reference(golden):
    g = golden("backbone_small")
    names = g["param_names"].tolist()
    bb = A.VisionTransformerDet(img_size=64, patch_size=16, embed_dim=128, depth=4, num_heads=2, mlp_ratio=4.,
                                qkv_bias=True, out_indices=(0, 1, 2, 3), last_feat=True, point_tokens_num=10,
                                num_classes=5, return_attention=True)
    assert sorted(bb.state_dict().keys()) == sorted(names)
    for n, s in zip(names, g["param_shapes"].tolist()):
        want = tuple(int(v) for v in s.split(",")) if s else ()
        assert tuple(bb.state_dict()[n].shape) == want, n


def test_hungarian_matching_orders_objects_by_token_index():
    gen = torch.Generator().manual_seed(0)
    T, ncls = 12, 5
    gt_points = torch.tensor([[30., 40.], [200., 120.], [90., 210.]])
    gt_labels = torch.tensor([1, 3, 0])
    pred = torch.rand(T, 2, generator=gen)
    pred[7] = gt_points[0] / 224
    pred[2] = gt_points[1] / 224
    pred[9] = gt_points[2] / 224
    cls = torch.randn(T, ncls, generator=gen) * 0.01
    pos, gt = RH.hungarian_point_match(pred, cls, gt_points, gt_labels, (224, 224, 3), reg_weight=10.0)
    assert pos.tolist() == [2, 7, 9] and gt.tolist() == [1, 0, 2]


def fake_crop_threshold_erode(maps, crops, thr, relative, k):
    """Test-side torch stand-in for ops.crop_threshold_erode (the HIP op needs a GPU)."""
    M, H, W = maps.shape
    out = torch.zeros(M, H, W, dtype=torch.uint8)
    counts = torch.zeros(M, dtype=torch.int32)
    for m in range(M):
        x0, y0, x1, y1 = (0, 0, W, H) if crops is None else [int(v) for v in crops[m]]
        sub = maps[m][y0:y1, x0:x1]
        if sub.numel() == 0:
            continue
        b = (sub > (sub.max() * thr if relative else thr)).float()
        if k > 1:
            b = O.erode(b, k)
        out[m][y0:y1, x0:x1] = b.to(torch.uint8)
        counts[m] = int(b.sum())
    return out, counts


def fake_rank_select(mask, ranks):
    """Test-side stand-in for ops.rank_select: literally mask[m].nonzero()[rank]."""
    out = torch.full(ranks.shape, -1, dtype=torch.long)
    for m in range(mask.shape[0]):
        nz = mask[m].nonzero()[:, 0]
        ok = ranks[m] < nz.numel()
        out[m][ok] = nz[ranks[m][ok].long()]
    return out


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_sampling_and_mask_points_share_the_reference_rng_stream(golden, tag, monkeypatch):
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    hp, wp, G = int(g["hp"]), int(g["wp"]), int(g["G"])
    monkeypatch.setattr(RH.ops, "crop_threshold_erode", fake_crop_threshold_erode)
    monkeypatch.setattr(RH.ops, "rank_select", fake_rank_select)
    cams = O.upsample_bilinear(inp["cams"], hp * 16, wp * 16)
    attn_sel = cams[t(g["best_idx"]), torch.arange(G)]
    torch.manual_seed(int(g["seed"]) + 1)
    nm = RH._minmax_maps(attn_sel)
    bg = RH.sample_point_grid(nm, 20, 0.1, False)
    fg = RH.sample_point_grid(nm, 20, 0.2, True, inp["points"])
    supp = RH.sample_point_grid(nm.mean(0, keepdim=True), 20, 0.1, False)
    assert_equal(t(g["points_bg"]), bg, "bg points")
    assert_equal(t(g["points_fg"]), torch.cat((fg, supp)), "fg points")
    # the slow per-object path (rare branches) draws the same stream
    torch.manual_seed(int(g["seed"]) + 1)
    assert_equal(bg, RH._sample_point_grid_slow(nm, 20, 0.1, False), "slow path == fast path")
    torch.manual_seed(int(g["seed"]) + 1)
    for _ in range(3):                                   # consume the three sampling draws, then the mask points
        pass
    RH.sample_point_grid(nm, 20, 0.1, False); RH.sample_point_grid(nm, 20, 0.2, True, inp["points"])
    RH.sample_point_grid(nm.mean(0, keepdim=True), 20, 0.1, False)
    coords, labels = RH.mask_sample_points(t(g["map_fg_last"]), t(g["map_bg_last"]), t(g["rois"]), float(g["pos_thr"]),
                                           float(g["neg_thr"]), int(g["num_gt"]), int(g["corr_size"]))
    assert_equal(t(g["mask_coords"]), coords, "mask point coords")
    assert_equal(t(g["mask_labels"]), labels, "mask point labels")


def test_down16_is_bit_identical_to_interpolate():
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(3, 224, 224, generator=gen)
    ref = torch.nn.functional.interpolate(x[None], (14, 14), mode="bilinear")[0]
    assert_close(ref, RH._down16(x), 1e-6, 1e-7, "bilinear /16 (general floats: last-bit rounding order only)")
    b = (x > 0.5).float()        # the path only thresholds the down-sampled BINARY erosion map: exact
    assert_equal(torch.nn.functional.interpolate(b[None], (14, 14), mode="bilinear")[0], RH._down16(b), "binary /16")


def test_merge_plan_matches_reference_greedy_loop():
    gen = torch.Generator().manual_seed(8)
    base = torch.randn(4, 16, generator=gen)
    prot = torch.cat([base[i:i + 1] + 0.05 * torch.randn(3, 16, generator=gen) for i in range(4)])   # 4 groups of 3
    prot = prot[torch.randperm(12, generator=gen)]
    keep = torch.ones(12, dtype=torch.bool)
    keep[5] = False
    ref = O.merge_parts([prot[keep]], 0.85)[0]
    got = RH.merge_parts(prot[None], keep[None], 0.85)[0]
    assert_close(ref, got, 1e-6, 1e-6, "merged prototypes")


@pytest.mark.parametrize("tag", ["tiny224", "mid320"])
def test_semantic_center_host_logic_with_oracle_backed_shift(golden, tag, monkeypatch):
    g = golden(f"shift_{tag}")
    inp = shift_case_inputs(g)
    hp, wp, G = int(g["hp"]), int(g["wp"]), int(g["G"])

    def fake_cosine_shift(feat, box_patch, obj_img, prot, n_shift, hp_, wp_, tau0=0.1, temp=0.1, return_trace=False):
        inbox = O.box_mask(box_patch.float(), (hp_, wp_)).flatten(1)
        p, s = O.cosine_shift(prot.clone(), feat[0][None] * inbox[..., None], feat[0], tau0, temp, n_shift)
        return p.reshape(prot.shape), s.reshape(prot.shape[0], prot.shape[1], -1)

    def fake_refine_similarity(feat, seeds, boxes_patch, num_obj, refine_times, tau, is_select, hp_, wp_):
        assert refine_times == 0 and not is_select          # part_similarity only needs the plain cosine map
        cos = torch.nn.functional.cosine_similarity(seeds[:, None, :], feat[None, :, :], dim=-1)
        return cos[None], seeds.clone()

    def fake_semantic_prestage(map_fg, thr, k=11, up=16):
        fg_inter, _bg, fg_bin = O.semantic_prestage(map_fg, map_fg, (map_fg.shape[-2] // up, map_fg.shape[-1] // up), thr)
        return fg_inter, fg_bin.to(torch.uint8), fg_bin.flatten(1).sum(1).int()

    def fake_part_stats(maps, rois, owner, stride=16):       # stdroi:222-262 per part, plain torch
        peak = maps.flatten(1).max(1)[0][:, None, None]
        at = (maps >= peak).float()
        cnt = at.sum(dim=[-2, -1])
        ys = torch.arange(maps.shape[1], dtype=torch.float32)[None, :, None]
        xs = torch.arange(maps.shape[2], dtype=torch.float32)[None, None, :]
        cy, cx = (at * ys).sum(dim=[-2, -1]) / cnt, (at * xs).sum(dim=[-2, -1]) / cnt
        c = (torch.stack((cx, cy), dim=1) + 0.5) * stride
        box = rois[owner.long()]
        inside = (c[:, 0] >= box[:, 0]) & (c[:, 0] <= box[:, 2]) & (c[:, 1] >= box[:, 1]) & (c[:, 1] <= box[:, 3])
        return c, torch.stack((cy.long(), cx.long()), dim=1), (maps > 0.9).sum(dim=[-2, -1]), inside

    monkeypatch.setattr(RH.ops, "cosine_shift", fake_cosine_shift)
    monkeypatch.setattr(RH.ops, "part_stats", fake_part_stats)
    monkeypatch.setattr(RH.ops, "filter_parts", lambda sim, fg, sim_thr=0.8, pos_thr=0.85: RH.filter_parts(sim, fg, pos_thr))
    monkeypatch.setattr(RH.ops, "semantic_prestage", fake_semantic_prestage)
    monkeypatch.setattr(RH.ops, "refine_similarity", fake_refine_similarity)
    monkeypatch.setattr(RH.ops, "crop_threshold_erode", fake_crop_threshold_erode)
    monkeypatch.setattr(RH.ops, "rank_select", fake_rank_select)
    head = A.AttnShiftRoIHead(num_semantic_points=int(g["num_semantic_points"]), mean_shift_times_local=int(g["n_shift"]))
    res = head.get_semantic_centers(t(g["map_fg_last"]), t(g["map_bg_last"]), t(g["rois"]), inp["vit_feat"],
                                    pos_thr=float(g["pos_thr"]), refine_times=int(g["n_shift"]), gt_labels=inp["labels"],
                                    num_semantic_points=int(g["num_semantic_points"]))
    centers, split, sim_parts, feat_split, feats, num_parts, coords_org, labels_org, corres = res
    assert_equal(g["num_parts"], np.array(num_parts), "num_parts")
    assert_close(t(g["coords_org"]), coords_org, 0, 0, "centre coords")
    assert_equal(g["corres_gt"], corres, "corres_gt")
    assert_equal(g["labels_org"], labels_org, "labels")
    if g["feats_all"].shape[0]:
        assert_close(t(g["feats_all"]), feats, 0, 0, "centre features")
    for i, n in enumerate(g["n_sim_parts"].tolist()):
        if n:
            assert_close(t(g[f"sim_parts{i}"]), sim_parts[i], 1e-4, 1e-5, f"sim_parts{i}")


def test_hungarian_matching_equals_the_reference_assigner(golden):
    """Fixture = the reference's HungarianPointAssigner.assign with FocalLossCost + PointL1Cost(10) executed on seeded
    cases (tools/gen_golden_hungarian.py): more tokens than points, more points than tokens, one point, no point."""
    g = golden("hungarian")
    for i in range(int(g["n"])):
        tt = lambda k: torch.from_numpy(g[f"{k}{i}"])
        pos, gt = RH.hungarian_point_match(tt("pred"), tt("cls"), tt("pts"), tt("labels"), tuple(int(v) for v in g[f"shape{i}"]),
                                           cls_weight=1.0, reg_weight=10.0)
        want = tt("gt_inds")                                          # 0 = unmatched, k + 1 = matched to point k
        assert torch.equal(pos, torch.nonzero(want > 0).flatten()), i
        assert torch.equal(gt, want[want > 0] - 1), i


def test_median_area_selector_batched_equals_per_image():
    """The depth selector of the attribute-only configuration computes all images in one pass (same launches as one image);
    per image it must give what the per-image loop gives, also for ties (stable order) and unequal object counts."""
    from attentionshift_amd.roi_head import median_area_selector
    g = torch.Generator().manual_seed(3)
    boxes = [torch.rand(3, 7, 4, generator=g) * 100, torch.rand(5, 7, 4, generator=g) * 100, torch.rand(1, 7, 4, generator=g)]
    boxes[1][2] = boxes[1][2, :1]                       # seven identical boxes: a 7-way tie
    for b in boxes:
        b[..., 2:] += b[..., :2]
    batched = median_area_selector(boxes)
    single = [median_area_selector([b])[0] for b in boxes]
    assert all(torch.equal(a, b) for a, b in zip(batched, single))
    assert int(batched[1][2]) == 3                      # stable sort: the middle of 0..6
    mixed = median_area_selector([boxes[0], boxes[1][:, :5]])           # different depth counts: per-image path
    assert torch.equal(mixed[0], single[0]) and mixed[1].shape == (5,)



def test_bench_gpus_flag_launches_or_fails_loudly():
    """bench.py --gpus N: without a launcher N > 1 re-executes under torch.distributed.run with one
    rank per GPU on 127.0.0.1; under a launcher the world size must equal --gpus; a node with fewer GPUs than ranks is an
    error unless AS_BENCH_SHARE_GPUS asks for the plumbing check.  (The re-exec itself runs in test_distributed_cpu.)"""
    import bench
    assert bench.launch_plan(1, {}, [], 0) == ("run", None)
    what, cmd = bench.launch_plan(4, {}, ["--gpus", "4", "--steps", "3"], 8, script="/x/bench.py")
    assert what == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]
    assert bench.launch_plan(2, {}, [], 1)[0] == "error"                                   # 2 ranks, 1 GPU: fail loudly
    assert bench.launch_plan(2, {"AS_BENCH_SHARE_GPUS": "1"}, [], 1)[0] == "spawn"         # ... unless sharing is asked for
    assert bench.launch_plan(2, {"WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2"}, [], 2) == ("run", None)
    assert bench.launch_plan(1, {"WORLD_SIZE": "2"}, [], 2)[0] == "error"                  # launcher and flag disagree
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8"}, [], 1)[0] == "error"
    assert bench.launch_plan(0, {}, [], 1)[0] == "error"


def test_bench_gpus_2_on_a_box_without_two_gpus_exits_2_instead_of_measuring_one_rank():
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AS_BENCH_SHARE_GPUS")}
    p = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert "--gpus 2" in p.stderr and p.stdout.strip() == ""
