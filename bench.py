"""Headline benchmark of the hot path (BASELINE.json: images/sec at 1024^2, ViT-B; kernel rooflines).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One STEP = one pass of the hot path over one batch (BASELINE config 2 per GPU: MAE-ViT-Base, 1024x1024,
2 images, 3 point-labelled objects per image, 7 roll-out layers, 5 mean-shift iterations, bf16):
    VisionTransformerDet.forward (12 blocks: QKV GEMM + flash SDPA + proj, MLP, FPN taps, point head)
  + AttnShiftRoIHead.seed_pseudo_gt (roll-out rows, CAM boxes via CCL, cosine refinement, instance maps,
    mean-shift token clustering, part centres, pseudo masks -> host numpy, exactly as the reference hands
    them to the mask head).
Inputs are synthetic and resident in HBM before the timed region.  Because randomly initialised weights give
near-uniform attention, the attention-shift stage consumes the seeded CAM rows / feature blobs of SURVEY 8d
(attentionshift_amd/synthetic.py); the roll-out itself is still computed from the real attention of the pass.
Images shard over GPUs with no data-path collective in this (forward / no-grad) path: weak scaling.

Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel (flash SDPA forward, MFMA-bound) timed with
HIP events on its launch stream inside the timed region; `roofline_affinity` = the mean-shift token-affinity
call (HBM-bound, algorithmic bytes of SURVEY 8d); `cpu_baseline` = the CPU oracle port timed on this box's
host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0

CFG = dict(img=1024, patch=16, embed_dim=768, depth=12, heads=12, batch=2, objects=3, cam_layer=7, n_shift=5,
           point_tokens=100, num_classes=20)


def build(device, rng_mode="fast", train=False, ranks=None):
    import attentionshift_amd as A
    from attentionshift_amd import synthetic

    torch.manual_seed(0)
    bb = A.build_backbone(dict(type="VisionTransformerDet", img_size=CFG["img"], patch_size=CFG["patch"],
                               embed_dim=CFG["embed_dim"], depth=CFG["depth"], num_heads=CFG["heads"], mlp_ratio=4.,
                               qkv_bias=True, drop_path_rate=0.05, out_indices=(3, 5, 7, 11), learnable_pos_embed=True,
                               use_checkpoint=True, last_feat=True, point_tokens_num=CFG["point_tokens"],
                               num_classes=CFG["num_classes"], return_attention=True, compute_dtype=torch.bfloat16))
    bb = bb.to(device)
    bb = bb.train() if train else bb.eval()

    hp = wp = CFG["img"] // CFG["patch"]
    G, Lc, T, B = CFG["objects"], CFG["cam_layer"], CFG["point_tokens"], CFG["batch"]
    rank = int(os.environ.get("RANK", "0"))                # every rank works on its own images (weak scaling)
    shift = [synthetic.shift_inputs(1234 + rank * B + b, hp, wp, CFG["embed_dim"], G, Lc) for b in range(B)]
    cams = torch.stack([s["cams"].flatten(2) for s in shift]).to(device)          # [B, Lc, G, Np]
    # vit_feat reaches seed_pseudo_gt the way the reference's caller builds it (two_stage_point_align.py:77): a
    # permuted VIEW [B, C, hp, wp] of the token-major last_feat [B, 1 + Np, C] with the cls row dropped
    last = torch.zeros(B, 1 + hp * wp, CFG["embed_dim"])
    last[:, 1:] = torch.stack([s["vit_feat"].flatten(1).t() for s in shift])
    vit_feat = last.to(device).permute(0, 2, 1)[..., 1:].unflatten(-1, (hp, wp))
    if os.environ.get("AS_BENCH_CHW", "0") == "1":          # A/B: channel-major contiguous copy instead of the view
        vit_feat = vit_feat.contiguous()
    gt_points = [s["points"].to(device) for s in shift]
    gt_labels = [s["labels"].to(device) for s in shift]

    class BenchHead(A.AttnShiftRoIHead):
        def rollout_cams(self, attns, num_proposals):
            rows = super().rollout_cams(attns, num_proposals)                   # real roll-out of this pass
            rows[:, :, :G, 1:-num_proposals] = cams                              # seeded, non-degenerate CAMs
            return rows

    head = BenchHead(num_semantic_points=5, mean_shift_times_local=CFG["n_shift"], rng_mode=rng_mode,
                     bbox_head=dict(type="MAEBoxHeadRec", seed_thr=0.2, seed_multiple=0.5, cam_layer=Lc,
                                    num_classes=CFG["num_classes"]),
                     mil_head=dict(type="MAEBoxHeadMIL", num_layers_query=Lc))
    img = synthetic.images(B, CFG["img"], CFG["img"], seed=rank).to(device)
    metas = [dict(img_shape=(CFG["img"], CFG["img"], 3)) for _ in range(B)]
    pos_inds = [torch.arange(G, device=device) for _ in range(B)]

    def pseudo_labels(out):
        return head.seed_pseudo_gt(out["feature"], metas, None, None, None, vit_feat=vit_feat, img=img,
                                   point_cls=out["outputs_class"], point_reg=out["outputs_coord"], attns=out["attns"],
                                   gt_points=gt_points, gt_points_labels=gt_labels, return_mask=True, pos_mask_thr=0.35,
                                   neg_mask_thr=0.8, num_mask_point_gt=10, corr_size=21, obj_tau=0.9,
                                   pos_inds=pos_inds, matched_gt=pos_inds)

    if not train:
        def step():
            return pseudo_labels(bb(img))

        step.head = head
        return step

    # DDP training step (BASELINE configs[2] shape per GPU): backbone forward under autograd (HIP attention fwd), the
    # no-grad attention shift on its outputs, backward (HIP attention bwd), bucketed RCCL gradient all-reduce
    # overlapped with backward, AdamW.  The detection losses need proposals from the RPN, which is not part of this build,
    # so the scalar that is differentiated is a fixed surrogate over every backbone output the heads consume.
    from attentionshift_amd.dist import GradAllReducer, convert_sync_batchnorm
    if ranks is not None and ranks.world > 1:
        convert_sync_batchnorm(bb, ranks)              # as mmdet/apis/train.py:95 does before wrapping the model in DDP
    params = [p for p in bb.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.05, fused=True)
    reducer = GradAllReducer(params, ranks)

    def train_step():
        out = bb(img)
        with torch.no_grad():
            labels = pseudo_labels(out)
        loss = out["outputs_class"].float().square().mean() + out["outputs_coord"].float().mean()
        loss = loss + out["last_feat"].float().square().mean() + out["org_feats"].float().mean()
        for f in out["feature"]:
            loss = loss + f.float().square().mean()
        loss.backward()
        reducer.finish()
        opt.step()
        opt.zero_grad(set_to_none=False)
        return labels

    train_step.head = head
    return train_step


def sdpa_traffic():
    """HBM bytes per as_sdpa_fwd call at this shape from the PMC passes kept under profiles/ (FETCH_SIZE + WRITE_SIZE,
    separate rocprofv3 --pmc runs, gfx950 correction applied there); None if the summary is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_sdpa_traffic.json")) as f:
            return float(json.load(f)["per_as_sdpa_fwd_call_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def shift_traffic():
    """HBM bytes per as_cosine_shift call (all 16 launches) from the PMC passes kept under profiles/ (same shape as the
    bench: 2 images x 3 objects, S=5); None if the summary is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_shift_traffic.json")) as f:
            return float(json.load(f)["per_call_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline():
    """The CPU oracle (a port of the reference's PyTorch path) on this box's host cores, bounded sample:
    ONE image; 2 of the 12 ViT-B blocks at N=4197 incl. the dense head-mean attention the reference keeps
    (extrapolated x6), the 7-layer row roll-out, and the full attention-shift chain (G=3, S=5)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import attnshift_oracle as O
    from attentionshift_amd import synthetic

    # many-core hosts thrash on the small ops of this path: cap the pool and report what was used
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cores = torch.get_num_threads()
    D, h, T = CFG["embed_dim"], CFG["heads"], CFG["point_tokens"]
    hp = wp = CFG["img"] // CFG["patch"]
    N = 1 + hp * wp + T
    shapes = {}
    for i in range(2):
        p = f"blocks.{i}."
        shapes.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "attn.qkv.weight": (3 * D, D),
                       p + "attn.qkv.bias": (3 * D,), p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,),
                       p + "norm2.weight": (D,), p + "norm2.bias": (D,), p + "mlp.fc1.weight": (4 * D, D),
                       p + "mlp.fc1.bias": (4 * D,), p + "mlp.fc2.weight": (D, 4 * D), p + "mlp.fc2.bias": (D,)})
    sd = synthetic.det_state_dict(shapes)
    with torch.no_grad():
        x = torch.randn(1, N, D, generator=torch.Generator().manual_seed(0))
        t0 = time.time()
        attns = []
        for i in range(2):
            x, p = O.block(x, sd, f"blocks.{i}.", h)
            attns.append(p.mean(1))
        t_blocks = time.time() - t0
        t0 = time.time()
        O.rollout_rows([attns[i % 2] for i in range(CFG["cam_layer"])], T)
        t_roll = time.time() - t0
        inp = synthetic.shift_inputs(1234, hp, wp, D, CFG["objects"], CFG["cam_layer"])
        torch.manual_seed(1)
        t0 = time.time()
        boxes, cams_up = O.cam_boxes_from_rollout(inp["cams"], inp["points"], 0.2, 0.5)
        best = torch.zeros(CFG["objects"], dtype=torch.long)
        rois = boxes[torch.arange(CFG["objects"]), best]
        attn_sel = cams_up[best, torch.arange(CFG["objects"])]
        fg, bg = O.sample_refine_inputs(attn_sel, inp["points"])
        m_fg, m_bg, _, _ = O.cosine_refined_maps(attn_sel, inp["vit_feat"], rois, fg, bg, 2, 0.9)
        O.mask_sample_points(m_fg[-1], m_bg[-1], rois, 0.35, 0.8, 10, 21)
        O.semantic_centers(m_fg[-1], m_bg[-1], rois, inp["vit_feat"], 0.35, CFG["n_shift"], inp["labels"], num_semantic_points=5)
        O.pseudo_masks(m_fg[-1], 0.35)
        t_shift = time.time() - t0
    per_image = t_blocks * (CFG["depth"] / 2) + t_roll + t_shift
    return dict(value=round(1.0 / per_image, 4), unit="images/sec", cores=cores, kind="port",
                sample=(f"1 image: 2/12 ViT-B blocks at N={N} with dense head-mean attention ({t_blocks:.1f}s, x6), "
                        f"7-layer row roll-out ({t_roll:.1f}s), full attention-shift chain G=3 S=5 ({t_shift:.1f}s)"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=5, help="steps of the extra training-step leg (0 = skip)")
    ap.add_argument("--train-multi", action="store_true",
                    help="also run the training leg when N > 1 (RCCL gradient all-reduce); off by default so that an "
                         "untested-fabric problem in the extra leg can never take the headline scaling run down")
    a = ap.parse_args()

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from attentionshift_amd.dist import Ranks
    ranks = Ranks(backend="nccl", device=device)          # "nccl" is RCCL on ROCm
    world, rank = ranks.world, ranks.rank

    from attentionshift_amd import ops
    # the host side of this path is a single Python thread; a 256-thread intra-op pool only adds spin-wait noise
    torch.set_num_threads(int(os.environ.get("AS_HOST_THREADS", "8")))
    # Sampling draws: "fast" = O(k) rejection draws for the first k entries of a random permutation (same distribution
    # as the reference's torch.randperm(n)[:k]); "reference" = the literal torch.randperm(n) stream, which costs O(n) host
    # work for the 1e5..1e6 candidate pixels of a 1024^2 crop.  The headline number uses AS_RNG_MODE (default fast); the
    # reference-stream rate is measured right after and reported next to it.
    rng_mode = os.environ.get("AS_RNG_MODE", "fast")
    step = build(device, rng_mode)
    with torch.no_grad():
        for _ in range(a.warmup):
            step()
        if os.environ.get("AS_BENCH_EVENTS", "1") == "1":
            ops.enable_timing(["sdpa_fwd", "cosine_shift"])
        ranks.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ranks.barrier()
        elapsed = time.perf_counter() - t0
    timing = ops.collect_timing()
    ops.disable_timing()
    elapsed = ranks.max_over_ranks(elapsed)
    other = "reference" if rng_mode == "fast" else "fast"
    step.head.rng_mode = other
    with torch.no_grad():
        step()
        ranks.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ranks.barrier()
        elapsed_other = ranks.max_over_ranks(time.perf_counter() - t0)

    # extra leg: the DDP training step (forward + attention shift + backward + gradient all-reduce + AdamW)
    train_rec = None
    if a.train_steps > 0 and (world == 1 or a.train_multi):
        del step
        torch.cuda.empty_cache()
        tstep = build(device, rng_mode, train=True, ranks=ranks)
        for _ in range(2):
            tstep()
        ops.enable_timing(["attn_bwd"])
        ranks.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.train_steps):
            tstep()
        torch.cuda.synchronize()
        ranks.barrier()
        t_train = ranks.max_over_ranks(time.perf_counter() - t0)
        ttiming = ops.collect_timing()
        ops.disable_timing()
        n_bwd, ms_bwd = ttiming.get("attn_bwd", (0, float("nan")))
        train_rec = {"images_per_sec": round(world * CFG["batch"] * a.train_steps / t_train, 3),
                     "ms_per_step": round(t_train / a.train_steps * 1e3, 3), "steps": a.train_steps,
                     "attn_bwd_ms_per_layer": round(ms_bwd, 4), "attn_bwd_launches_timed": n_bwd,
                     "what": "backbone fwd (autograd, HIP attention fwd) + no-grad attention shift + bwd (HIP attention "
                             "bwd) + bucketed RCCL grad all-reduce overlapped with bwd + fused AdamW; surrogate loss "
                             "over all backbone outputs; batch 2/GPU"}

    if rank == 0:
        B, N, h = CFG["batch"], 1 + (CFG["img"] // CFG["patch"]) ** 2 + CFG["point_tokens"], CFG["heads"]
        n_sdpa, ms_sdpa = timing.get("sdpa_fwd", (0, float("nan")))
        flops_sdpa = 4.0 * B * h * N * N * 64                      # QK^T + PV per launch (one layer, one batch)
        ach = flops_sdpa / (ms_sdpa * 1e-3) / 1e12
        n_cs, ms_cs = timing.get("cosine_shift", (0, float("nan")))
        Np, C, S, G, P = (CFG["img"] // CFG["patch"]) ** 2, CFG["embed_dim"], CFG["n_shift"], CFG["objects"], 20
        imgs_per_call = max(1, round(B * a.steps / max(n_cs, 1)))         # the head batches a step's images into one call
        bytes_cs = imgs_per_call * ((2 * S + 1) * Np * C * 4 + G * P * Np * 4)   # SURVEY 8d, per call
        gbps = bytes_cs / (ms_cs * 1e-3) / 1e9
        rec = {
            "metric": "images/sec (1024^2, ViT-B) hot path: backbone attention fwd + attention-shift pseudo-labels",
            "value": round(world * B * a.steps / elapsed, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "rng_mode": rng_mode, f"images_per_sec_{other}_rng": round(world * B * a.steps / elapsed_other, 3),
            "config": {"workload": "BASELINE configs[1]: MAE-ViT-Base 1024x1024, batch 2/GPU, 3 objects/img, "
                                   "7 roll-out layers, 5 shift iters, forward + no-grad attention shift",
                       "global_batch": world * B, "parallelism": f"dp{world} (image sharding, no data-path collective)"},
            "roofline": {"kernel": "as_sdpa_fwd (bf16): sdpa_fwd_glds_kernel<false> on 32 of 33 q-tiles, concurrently "
                                   "sdpa_fwd_glds_kernel<true> + sdpa_combine_kernel for the key-split last q-tile on a "
                                   "helper stream; one timed 'launch' = the whole call",
                         "bound": "mfma", "achieved": round(ach, 2),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                         "traffic": sdpa_traffic(), "launches_timed": n_sdpa, "ms_per_launch": round(ms_sdpa, 4),
                         "flops_per_launch": flops_sdpa},
            "roofline_affinity": {"kernel": "as_cosine_shift (similarity / assign / aggregate x S + final similarity)", "bound": "hbm",
                                  "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                  "frac": round(gbps / PEAK_HBM_GBPS, 4), "traffic": shift_traffic() if imgs_per_call == 2 else None,
                                  "calls_timed": n_cs, "images_per_call": imgs_per_call,
                                  "ms_per_call": round(ms_cs, 4), "algorithmic_bytes_per_call": bytes_cs},
        }
        if train_rec is not None:
            rec["train"] = train_rec
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
