"""BASELINE config-2 size (64x64 patches, C=768, G=3, 7 roll-out layers, 5 shift iterations) parity of Part B on the
GPU against the reference-run fixture tests/golden/shift_cfg2.npz (tools/gen_golden.py --cfg2-only) and the oracle.

What "bit-exact argmax" means at this size.  The reference evaluates cos(prototype, patch) as a broadcast product summed
by torch's reduction; ANY other summation order (a matmul on the CPU, an MFMA chain on the GPU) changes the last bits,
and with 12 288 decisions per iteration a few of them are decided by those bits (tests/test_oracle_golden.py pins this
on the CPU: the oracle's own matmul form flips 1..16 patches per iteration against the reference).  So the bar is:
every assignment of every iteration equals the reference arithmetic's argmax EVALUATED FROM THE KERNEL'S OWN STATE of
that iteration, except patches whose two candidates are tied to within the fp32 evaluation noise or whose softmax
weights both underflow (helpers.check_shift_decisions).  Everything else (boxes, sampled points, mask points, masks,
part counts) is compared bit for bit.
"""
import os

import numpy as np
import pytest
import torch

import attnshift_oracle as O
from helpers import assert_close, assert_equal, check_shift_decisions, shift_case_inputs, shift_state_inputs, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def dev(x):
    return x.cuda().contiguous()


@pytest.fixture(scope="module")
def ops():
    from attentionshift_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def cfg2(golden):
    g = golden("shift_cfg2")
    return g, shift_case_inputs(g)


def unpack_masks(g):
    H, W = int(g["hp"]) * 16, int(g["wp"]) * 16
    return np.unpackbits(g["pseudo_masks_packed"], axis=-1)[..., :W].reshape(-1, H, W)


def _rows_close(ref, got, what):
    """Unnormalised prototypes = sum of w * feature with w = softmax(cos / (temp * tau)): at tau ~ 1e-3 a cosine's fp32
    round-off (1e-7) moves a weight by 1e-3 relative, so the SCALE of a row carries that noise (the consumers normalise
    it away); compared to 1e-3 of each row's magnitude."""
    scale = ref.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    assert_close(ref / scale, got / scale, 0, 1e-3, what)


def test_cosine_shift_every_iteration_is_the_reference_argmax(ops, cfg2):
    """B4 at config-2 size: each of the 5 iterations of as_cosine_shift is checked from the kernel's own state
    (prototypes after k iterations = a call with n_shift=k; tau from the trace) against ONE iteration of the reference
    arithmetic (oracle, faithful broadcast cosine): argmax equal on every determined decision, new prototypes / tau of
    the clusters no coin flip touched within 1e-3.  Then the whole run against the reference's own trajectory."""
    g, inp = cfg2
    hp, wp, G, S = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["n_shift"])
    feats, tok, prot0, box_patch = shift_state_inputs(g, inp)
    obj_img = torch.zeros(G, dtype=torch.int32)
    run = lambda k, trace=False: ops.cosine_shift(dev(tok[None]), dev(box_patch), dev(obj_img), dev(prot0), k, hp, wp,
                                                  return_trace=trace)
    pout, sim, assign, tau = run(S, True)
    states = [prot0] + [run(k)[0].cpu() for k in range(1, S)] + [pout.cpu()]
    flips_total = 0
    record = []                                              # per-iteration counts -> gpurun_out/r06_shift_flips.json
    for k in range(S):
        tau_k = 0.1 if k == 0 else tau[k - 1].cpu()[..., None]        # python float at iteration 0, as the reference
        step = O.cosine_shift_step(states[k], feats, tau_k, faithful=True)
        n, near, under = check_shift_decisions(step, assign[k], states[k], feats, tau_k, what=f"iteration {k}")
        flips_total += n
        # clusters whose membership is identical on both sides: same sums up to rounding
        same = torch.ones(G, 20, dtype=torch.bool)
        bad = assign[k].long().cpu() != step["win"]
        for gi, ni in zip(*bad.nonzero(as_tuple=True)):
            same[gi, assign[k][gi, ni].long()] = False
            same[gi, step["win"][gi, ni]] = False
        _rows_close(step["prot"][same], states[k + 1][same], f"prototypes after iteration {k}")
        assert_close(step["tau"][..., 0][same], tau[k].cpu()[same], 1e-3, 2e-6, f"tau after iteration {k}")
        print(f"[cfg2] iteration {k}: {n} coin-flip patches (near ties {near}, underflow {under}) of {assign[k].numel()}")
        record.append(dict(iteration=k, coin_flip_patches=int(n), near_ties=int(near), underflow=int(under),
                           assignments=int(assign[k].numel())))
    # against the reference's own trajectory (fixture): identical unless a coin flip moved a patch
    ref_assign = t(g["ref_assign"]).long()
    diff = [(assign[k].long().cpu() != ref_assign[k]).sum().item() for k in range(S)]
    print(f"[cfg2] patches assigned differently from the reference run, per iteration: {diff}; coin flips {flips_total}")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):                               # (the relaxed argmax bar stays visible)
        import json
        with open(os.path.join(out_dir, "r06_shift_flips.json"), "w") as f:
            json.dump({"case": "tests/golden/shift_cfg2.npz (64 x 64 patches, C = 768, G = 3, 20 seeds, 5 iterations)",
                       "bar": "argmax equal on every DETERMINED decision; a patch may differ only where the fp64 margin between its two "
                              "best clusters is inside the fp32 noise of the cosine (tests/helpers.py check_shift_decisions)",
                       "per_iteration": record, "differs_from_reference_run_per_iteration": [int(d) for d in diff],
                       "bound_per_iteration": 16}, f, indent=1)
    if flips_total == 0:
        assert sum(diff) == 0, "no coin flips, so the whole trajectory must equal the reference's"
        _rows_close(t(g["ref_prot"]), pout.reshape(-1, pout.shape[-1]).cpu(), "prototypes vs reference")
        assert_close(t(g["ref_sim"]).flatten(1), sim.reshape(-1, hp * wp).clamp(min=0), 1e-3, 1e-5, "sim vs reference")
    assert max(diff) <= 16, diff        # the bound the CPU matmul-form oracle itself meets against the reference
    direct = O.cos_matrix(pout.cpu(), tok)
    assert_close(direct, sim, 1e-3, 1e-5, "final sim == cos(returned prototypes, unmasked features)")


def _build_head(g, rng_mode):
    import attentionshift_amd as A
    return A.build_head(dict(type="AttnShiftRoIHead", num_semantic_points=int(g["num_semantic_points"]),
                             mean_shift_times_local=int(g["n_shift"]), rng_mode=rng_mode,
                             bbox_head=dict(type="MAEBoxHeadRec", seed_thr=float(g["cam_thr"]),
                                            seed_multiple=float(g["area_ratio"]), cam_layer=int(g["Lc"]), num_classes=20)))


def _run_head(head, g, inp, monkeypatch, images=1):
    hp, wp, G, Lc = int(g["hp"]), int(g["wp"]), int(g["G"]), int(g["Lc"])
    T, N = 10, 1 + hp * wp + 10
    rows = torch.zeros(images, Lc, T, N)
    rows[:, :, :G, 1:-T] = inp["cams"].flatten(2)
    monkeypatch.setattr(head, "rollout_cams", lambda attns, n, pos_inds=None: rows.cuda())
    best = t(g["best_idx"]).cuda()
    head.layer_selector = lambda boxes, labels, fmap: [best] * images
    torch.manual_seed(int(g["seed"]) + 1)
    ar = torch.arange(G).cuda()
    return head.seed_pseudo_gt(None, [dict(img_shape=(hp * 16, wp * 16, 3))] * images, None, None, None,
                               vit_feat=inp["vit_feat"][None].repeat(images, 1, 1, 1).cuda(),
                               point_cls=torch.zeros(images, T, 20).cuda(), point_reg=torch.zeros(images, T, 2).cuda(),
                               attns=None, gt_points=[inp["points"].cuda()] * images,
                               gt_points_labels=[inp["labels"].cuda()] * images, return_mask=True,
                               pos_mask_thr=float(g["pos_thr"]), neg_mask_thr=float(g["neg_thr"]),
                               num_mask_point_gt=int(g["num_gt"]), corr_size=int(g["corr_size"]), obj_tau=float(g["obj_tau"]),
                               pos_inds=[ar] * images, matched_gt=[ar] * images)


def _masks_equal_up_to_threshold_edge(ref_masks, got_masks, map_fg, pos_thr, what):
    """B6: uint8 masks of a float threshold.  The GPU's maps differ from the CPU's in the last bits (different summation
    order of the patch-grid cosines), so a pixel may differ only if its value sits on the threshold to within that
    noise: |map - peak * thr| <= 2e-6 (maps are in [0, 1]).  Everything else must be identical."""
    diff = ref_masks != got_masks
    n = int(diff.sum())
    if n:
        peak = map_fg.flatten(1).max(1)[0][:, None, None]
        edge = ((map_fg - peak * pos_thr).abs() <= 2e-6).numpy()
        assert not (diff & ~edge).any(), f"{what}: {int((diff & ~edge).sum())} pixels differ away from the threshold"
        assert n <= 16, f"{what}: {n} threshold-edge pixels differ"
    return n


def test_seed_pseudo_gt_chain_full_size_reference_rng(golden, cfg2, monkeypatch):
    """The whole chain B1..B6 of AttnShiftRoIHead.seed_pseudo_gt at config-2 size, the reference's RNG stream, against
    the reference-run fixture: boxes, mask points, masks, part counts bit-exact; float maps / features 1e-3."""
    g, inp = cfg2
    G = int(g["G"])
    head = _build_head(g, "reference")
    head.capture = []
    out = _run_head(head, g, inp, monkeypatch)
    # the draws were made ON THE DEVICE from torch's engine state (csrc/mt19937.hip), not on the host path
    assert head.rng_stats == dict(device_calls=1, host_redos=0), head.rng_stats
    assert_equal(t(g["rois"]), out["pseudo_gt_bboxes"][0], "pseudo boxes (B1)")
    cap = {k: v for d in head.capture for k, v in d.items()}
    assert_equal(t(g["points_fg"]), cap["points_fg"], "sampled fg points (B2)")
    assert_equal(t(g["points_bg"]), cap["points_bg"], "sampled bg points (B2)")
    sub = int(g["map_sub"])
    assert_close(t(g["map_fg_sub"])[-1], out["map_cos_fg"][0][:, ::sub, ::sub], 1e-3, 1e-5, "map_cos_fg (B2)")
    assert_close(t(g["fg_feat"]), out["inst_fg_feat"][0].flatten(1), 1e-3, 1e-4, "inst_fg_feat")
    assert_equal(t(g["mask_coords"]), out["mask_points_coords"][0], "mask point coords (B2')")
    assert_equal(t(g["mask_labels"]), out["mask_points_labels"][0], "mask point labels (B2')")
    assert_equal(t(g["seed_coords"]), cap["seeds"], "grid seeds (B4 input)")
    assert_equal(t(g["fg_inter"]), cap["fg_inter"], "patch-grid foreground (B3)")
    # B6: the oracle's full-resolution map (bit-identical to the reference's, tests/test_oracle_golden.py) locates the
    # threshold-edge pixels
    boxes, cams = O.cam_boxes_from_rollout(inp["cams"], inp["points"], float(g["cam_thr"]), float(g["area_ratio"]))
    attn_sel = cams[t(g["best_idx"]), torch.arange(G)]
    m_fg, _m_bg, _, _ = O.cosine_refined_maps(attn_sel, inp["vit_feat"], t(g["rois"]), t(g["points_fg"]), t(g["points_bg"]), 2,
                                             float(g["obj_tau"]))
    n_edge = _masks_equal_up_to_threshold_edge(unpack_masks(g), out["pseudo_gt_masks"][0], m_fg[-1], float(g["pos_thr"]),
                                               "pseudo masks (B6)")
    print(f"[cfg2] pseudo masks: {n_edge} threshold-edge pixels differ of {out['pseudo_gt_masks'][0].size}")
    assert_equal(g["num_parts"], np.array(out["num_parts"][0]), "num_parts (B5)")
    assert_equal(g["corres_gt"], out["corres_gts"][0], "corres_gts")
    assert_close(t(g["coords_org"]), out["semantic_centers_org"][0][0], 0, 0, "part centres (B5)")


def test_seed_pseudo_gt_chain_full_size_fast_rng(golden, cfg2, monkeypatch):
    """The headline mode of bench.py (rng_mode='fast': draws on the device, one readback per image) at config-2 size,
    two images in one call.  Its draws differ from the reference's stream by design, so the DEVICE-DRAWN sample points
    are fed back into the oracle and every downstream output is compared with the oracle's on those samples."""
    g, inp = cfg2
    G, hp, wp = int(g["G"]), int(g["hp"]), int(g["wp"])
    head = _build_head(g, "fast")
    head.capture = []
    out = _run_head(head, g, inp, monkeypatch, images=2)
    assert len(head.capture) == 4, "two images: sample points + seeds each"
    pos_thr, neg_thr = float(g["pos_thr"]), float(g["neg_thr"])
    boxes, cams = O.cam_boxes_from_rollout(inp["cams"], inp["points"], float(g["cam_thr"]), float(g["area_ratio"]))
    rois = t(g["rois"])
    attn_sel = cams[t(g["best_idx"]), torch.arange(G)]
    norm = O.minmax_maps(attn_sel)
    for i in range(2):
        assert_equal(rois, out["pseudo_gt_bboxes"][i], "pseudo boxes (B1)")
        pts = [d for d in head.capture if "points_fg" in d][i]
        seeds = [d for d in head.capture if d.get("image") == i][0]
        pfg, pbg = pts["points_fg"].cpu(), pts["points_bg"].cpu()
        # the draws respect the candidate sets of stdroi:343-371 (fg >= 0.2, bg < 0.1 of the min-max normalised CAM)
        val = lambda m, p: m[p[..., 1].long(), p[..., 0].long()]
        for o in range(G):
            assert (val(norm[o], pfg[o]) >= 0.2).all() and (val(norm[o], pbg[o]) < 0.1).all(), "draws outside the candidate set"
        assert (val(norm.mean(0), pfg[G]) < 0.1).all(), "shared background draws outside the candidate set"
        m_fg, m_bg, f_fg, f_bg = O.cosine_refined_maps(attn_sel, inp["vit_feat"], rois, pfg, pbg, 2, float(g["obj_tau"]))
        assert_close(m_fg[-1], out["map_cos_fg"][i], 1e-3, 1e-5, "map_cos_fg on the device's samples")
        assert_close(f_fg, out["inst_fg_feat"][i].flatten(1), 1e-3, 1e-4, "inst_fg_feat")
        _masks_equal_up_to_threshold_edge(O.pseudo_masks(m_fg[-1], pos_thr), out["pseudo_gt_masks"][i], m_fg[-1], pos_thr,
                                          "pseudo masks (B6)")
        # mask points: distinct, labelled by the candidate set they were drawn from (stdroi:433-461)
        coords, labels = out["mask_points_coords"][i].cpu(), out["mask_points_labels"][i].cpu()
        for o in range(G):
            x0, y0, x1, y1 = (int(v) for v in rois[o])
            crop_fg, crop_bg = m_fg[-1][o, y0:y1, x0:x1], m_bg[-1][o, y0:y1, x0:x1]
            cand_fg = O.erode((crop_fg > crop_fg.max() * pos_thr).float()[None], int(g["corr_size"]))[0] > 0
            cand_bg = crop_bg > crop_bg.max() * neg_thr
            cx, cy = coords[o, :, 0].long() - x0, coords[o, :, 1].long() - y0
            # a drawn point may fall outside the CPU-evaluated candidate set only if the pixel that decides it sits on
            # the threshold to within the maps' last-bit noise: the drawn pixel itself for a background point, a pixel of
            # the drawn pixel's erosion window for a foreground point
            edge_fg = (crop_fg - crop_fg.max() * pos_thr).abs() <= 2e-6
            edge_bg = (crop_bg - crop_bg.max() * neg_thr).abs() <= 2e-6
            rad = int(g["corr_size"]) // 2
            for j in range(coords.shape[1]):
                y, x = int(cy[j]), int(cx[j])
                if labels[o, j]:
                    ok = bool(cand_fg[y, x]) or bool(edge_fg[max(y - rad, 0):y + rad + 1, max(x - rad, 0):x + rad + 1].any())
                else:
                    ok = bool(cand_bg[y, x]) or bool(edge_bg[y, x])
                assert ok, f"mask point {j} of object {o} is not a candidate of its label"
            assert len({(int(a), int(b)) for a, b in coords[o]}) == coords.shape[1], "mask points must be distinct"
        fg_inter, _bg, fg_bin = O.semantic_prestage(m_fg[-1], m_bg[-1], (hp, wp), pos_thr)
        assert_equal(fg_inter, seeds["fg_inter"], "patch-grid foreground (B3)")
        assert_equal(O.grid_seed_coords(fg_bin, rois), seeds["seeds"], "grid seeds")
        res = O.semantic_centers(m_fg[-1], m_bg[-1], rois, inp["vit_feat"], pos_thr, int(g["n_shift"]), inp["labels"],
                                 num_semantic_points=int(g["num_semantic_points"]))
        assert_equal(np.array(res["num_parts"]), np.array(out["num_parts"][i]), "num_parts (B5)")
        assert_equal(res["corres_gt"], out["corres_gts"][i], "corres_gts")
    # two identical images in one call give identical deterministic stages
    assert_equal(out["pseudo_gt_bboxes"][0], out["pseudo_gt_bboxes"][1], "image 0 == image 1 boxes")


def test_reference_rng_device_draws_equal_host_draws_full_size(golden, cfg2, monkeypatch):
    """Reference-RNG mode at config-2 size, two images in one call: the draws made on the device from torch's own engine
    state (the default) and the draws made on the host from the global generator (AS_REF_RNG_HOST=1: three count readbacks
    per image) give the same sampled seeds, mask points, centres and masks bit for bit -- and leave torch's global
    generator in the same state (the next torch.rand agrees)."""
    g, inp = cfg2

    def run(host):
        if host:
            monkeypatch.setenv("AS_REF_RNG_HOST", "1")
        else:
            monkeypatch.delenv("AS_REF_RNG_HOST", raising=False)
        head = _build_head(g, "reference")
        head.capture = []
        out = _run_head(head, g, inp, monkeypatch, images=2)
        return out, head.capture, torch.rand(4), head.rng_stats

    host, cap_h, next_h, st_h = run(True)
    dev, cap_d, next_d, st_d = run(False)
    assert st_h == dict(device_calls=0, host_redos=0) and st_d == dict(device_calls=1, host_redos=0), (st_h, st_d)
    assert torch.equal(next_h, next_d)                                # the generator ends where the host path leaves it
    pts_h = [c for c in cap_h if "points_fg" in c]
    pts_d = [c for c in cap_d if "points_fg" in c]
    assert len(pts_h) == len(pts_d) == 2
    for a, b in zip(pts_h, pts_d):
        assert_equal(a["points_fg"], b["points_fg"], "sampled foreground seeds")
        assert_equal(a["points_bg"], b["points_bg"], "sampled background seeds")
    for i in range(2):
        assert_equal(host["mask_points_coords"][i], dev["mask_points_coords"][i], "mask points")
        assert_equal(host["mask_points_labels"][i], dev["mask_points_labels"][i], "mask point labels")
        assert (host["pseudo_gt_masks"][i] == dev["pseudo_gt_masks"][i]).all()
        assert host["num_parts"][i] == dev["num_parts"][i]
        assert_equal(host["semantic_centers_org"][0][i], dev["semantic_centers_org"][0][i], "part centres")
    assert_equal(t(g["mask_coords"]), dev["mask_points_coords"][0], "image 0's mask points vs the reference fixture")
