"""Full-width, full-depth backbone parity at the headline size (VERDICT r05 "missing" #2 / "weak" #1).

ONE image through VisionTransformerDet.forward as BASELINE config 2 runs it -- ViT-B: D = 768, 12 heads, 12 blocks, 1024 x 1024
-> N = 1 + 4096 + 100 = 4197 tokens -- against the oracle's backbone_forward on the host (reference
mmdet/models/backbones/visual_transformer_det.py:221-275 over models/vision_transformer.py:109-124; 12 dense [12, N, N]
softmaxes, ~1-2 s per block).  Compared: last_feat, the four tapped block outputs (org_feats) and the FPN maps made of them,
the point head's class / coordinate outputs, and the 7-layer roll-out rows of the point tokens.

  * compute_dtype=float32 (exact-fp32 MFMA path): north_star's 1e-3 of the output range on every output, end to end;
  * compute_dtype=bfloat16 (the path bench.py times):
      - every one of the 12 blocks on its own, fed the ORACLE's input of that block ("teacher-forced"): the block's update
        x_out - x_in within 3e-2 of the range of the oracle's update (the bar of tests/test_gpu_path.py), i.e. each layer of
        the timed path computes the reference's function at the headline width and token count;
      - end to end (free running): the errors are RECORDED, not held to 3e-2: this synthetic network (random weights with
        peaked attention, synthetic.det_state_dict) amplifies a perturbation of its tokens ~x100-1000 over 12 blocks -- measured
        here on the fp32 path with a known relative perturbation of 1e-4 -- so bf16 operand rounding (2^-9 per operand) ends
        up at 4e-2 of the range on the first tap and 2.6e-1 on the last (profiles/r06_backbone_fullsize_err.md).  The bound
        asserted end to end is the one that follows from the measured gain.
    The observed max / mean errors of all legs are written to gpurun_out/r06_backbone_fullsize_err.json.
"""
import json
import os

import pytest
import torch

import attnshift_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(img=1024, patch=16, embed_dim=768, depth=12, heads=12, point_tokens=100, num_classes=20, cam_layer=7,
           out_indices=(3, 5, 7, 11))


@pytest.fixture(scope="module")
def case():
    import attentionshift_amd as A
    from attentionshift_amd import synthetic
    c = CFG

    def make(dtype):
        bb = A.build_backbone(dict(type="VisionTransformerDet", img_size=c["img"], patch_size=c["patch"], embed_dim=c["embed_dim"],
                                   depth=c["depth"], num_heads=c["heads"], mlp_ratio=4., qkv_bias=True, drop_path_rate=0.,
                                   out_indices=c["out_indices"], last_feat=True, point_tokens_num=c["point_tokens"],
                                   num_classes=c["num_classes"], return_attention=True, compute_dtype=dtype))
        return bb

    bb32 = make(torch.float32)
    sd = synthetic.det_state_dict({k: tuple(v.shape) for k, v in bb32.state_dict().items()})
    img = synthetic.images(1, c["img"], c["img"], seed=11)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    trace = []
    try:
        ref = O.backbone_forward(img, sd, patch_size=c["patch"], depth=c["depth"], num_heads=c["heads"],
                                 out_indices=c["out_indices"], point_tokens_num=c["point_tokens"], trace=trace)
        ref["rollout_rows"] = O.rollout_rows(ref["attns"][-c["cam_layer"]:], c["point_tokens"])
        ref.pop("attns")                                     # 12 x 70 MB of dense head-mean attention: not needed any more
    finally:
        torch.set_num_threads(threads)
    return dict(make=make, sd=sd, img=img, ref=ref, trace=trace, report={})


def errors(ref, got):
    ref, got = ref.double().cpu(), got.double().cpu()
    scale = float(ref.abs().max()) + 1e-30
    d = (ref - got).abs()
    return float(d.max()) / scale, float(d.mean()) / scale


def run(case, dtype):
    from attentionshift_amd import ops
    bb = case["make"](dtype)
    bb.load_state_dict(case["sd"])
    bb = bb.cuda().eval()
    out = bb(case["img"].cuda())
    ref = case["ref"]
    res = {}
    for k in ("last_feat", "point_tokens", "outputs_class", "outputs_coord", "org_feats"):
        res[k] = errors(ref[k], out[k])
    for i, f in enumerate(out["feature"]):
        res[f"feature{i}"] = errors(ref["feature"][i], f)
    rows = ops.rollout_rows(out["attns"][-CFG["cam_layer"]:], CFG["point_tokens"])
    res["rollout_rows"] = errors(ref["rollout_rows"], rows)
    assert all(torch.isfinite(out[k].float()).all() for k in ("last_feat", "outputs_coord"))
    return res


def write_report(case):
    d = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(d):
        return
    with open(os.path.join(d, "r06_backbone_fullsize_err.json"), "w") as f:
        json.dump({"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in CFG.items()},
                   "metric": "max|ref - got| / max|ref| and mean|ref - got| / max|ref| against oracle.backbone_forward (fp32, host)",
                   "paths": case["report"]}, f, indent=1)


def test_fp32_path_meets_1e_3_at_vit_b_1024(case):
    res = run(case, torch.float32)
    case["report"]["float32"] = {k: {"max": v[0], "mean": v[1]} for k, v in res.items()}
    write_report(case)
    print("fp32 path:", {k: f"{v[0]:.2e}" for k, v in res.items()})
    for k, (mx, _) in res.items():
        assert mx < 1e-3, (k, mx)


def perturbation_gain(case, eps=1e-4):
    """How much this network amplifies a perturbation of its input tokens: the fp32 path twice, the second time with the
    image perturbed by eps of its range; returns (relative perturbation of last_feat) / eps."""
    bb = case["make"](torch.float32)
    bb.load_state_dict(case["sd"])
    bb = bb.cuda().eval()
    img = case["img"].cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = torch.randn(img.shape, device="cuda", generator=g) * (eps * float(img.abs().max()))
    a = bb(img)["last_feat"].double()
    b = bb(img + noise)["last_feat"].double()
    return float((a - b).abs().max() / a.abs().max()) / eps, float((a - b).abs().mean() / a.abs().max()) / eps


def test_bf16_blocks_teacher_forced_and_end_to_end_error_growth_at_vit_b_1024(case):
    trace = case["trace"]
    bb = case["make"](torch.bfloat16)
    bb.load_state_dict(case["sd"])
    bb = bb.cuda().eval()
    per_block = []
    for i, blk in enumerate(bb.blocks):                      # each block of the timed path on the oracle's own input
        x_in = trace[i].cuda()
        x_out, _, _ = bb._block(blk, x_in, None, False, True)
        upd_ref = (trace[i + 1] - trace[i]).double()
        upd = (x_out.double().cpu() - trace[i].double())
        scale = float(upd_ref.abs().max())
        per_block.append((float((upd - upd_ref).abs().max()) / scale, float((upd - upd_ref).abs().mean()) / scale))
    gain_max, gain_mean = perturbation_gain(case)
    res = run(case, torch.bfloat16)
    case["report"]["bfloat16"] = {k: {"max": v[0], "mean": v[1]} for k, v in res.items()}
    case["report"]["bfloat16_blocks_teacher_forced"] = [{"block": i, "max": m, "mean": a} for i, (m, a) in enumerate(per_block)]
    case["report"]["fp32_perturbation_gain_last_feat"] = {"eps": 1e-4, "gain_of_max": gain_max, "gain_of_mean": gain_mean}
    write_report(case)
    print("bf16 per-block (teacher-forced) max:", [f"{m:.2e}" for m, _ in per_block])
    print("bf16 end to end:", {k: f"{v[0]:.2e}" for k, v in res.items()})
    print(f"fp32 perturbation gain over 12 blocks: max x{gain_max:.0f}, mean x{gain_mean:.0f}")
    for i, (mx, _) in enumerate(per_block):
        assert mx < 3e-2, (i, mx)
    # end to end: bounded by what the measured amplification makes of one block's bf16 noise (mean error of a block x gain of the
    # mean, with a factor 4 of slack: twelve blocks inject noise, the later ones are amplified less); finite everywhere
    block_noise = max(a for _, a in per_block)
    assert res["last_feat"][1] < 4.0 * max(gain_mean, 1.0) * block_noise + 3e-2, (res["last_feat"], gain_mean, block_noise)
    assert res["rollout_rows"][1] < 3e-2 and res["outputs_coord"][1] < 1e-1
