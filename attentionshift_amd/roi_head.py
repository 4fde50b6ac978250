"""AttnShiftRoIHead -- the no-grad attention-shift pseudo-label generator on MI355X.

Mirror of the reference RoI head's `seed_pseudo_gt` (same registry names, constructor kwargs, call
signature, output dict keys and error behaviour):
    reference mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py (`stdroi` below)
      class :1325-1390 (registered as StandardRoIHeadMaskPointSampleDeformAttnReppoints, requested by
      configs/mae/attnshift_voc12aug.py:60 as AttnShiftRoIHead -- both names are registered here)
      seed_pseudo_gt :2209-2415, caller mmdet/models/detectors/two_stage_point_align.py:75-126

Stage map -> where it runs:
  A3  roll-out rows of the matched point tokens      ops.rollout_rows   (csrc/rollout.hip)
  B1  upsample + CAM boxes via connected components  ops.cam_boxes      (csrc/ccl.hip)
  B2  cosine-affinity refinement, instance maps      ops.refine_similarity / ops.instance_maps
  B4  mean-shift token clustering                    ops.cosine_shift   (csrc/cosine_shift.hip)
  B2', B3, B5, B6 (point sampling, erosion, part filtering/merging, masks): fused HIP kernels for the data-parallel
      parts (ops.mask_candidates / semantic_prestage / filter_parts / part_stats / merge_plan / rank_select) around
      small data-dependent host logic; rng_mode "reference" draws from torch's global CPU generator in exactly the
      reference's order (equal seeds give equal samples), "fast" draws on the device with one readback per image.
The consumers of the pseudo labels sit on the same class: the MIL / box / mask sub-heads are built when
their configs carry construction arguments (mil_head.py, mae_heads.py), `forward_train` (stdroi:2513-2727) and
`simple_test` (:3192-3221) compose them; with attribute-only configs the one value the pseudo-label path needs from
the MIL head -- which roll-out depth to use per object -- comes from `layer_selector` (a callable; default: the depth
whose CAM box has the median area).
"""
import types

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .registry import HEADS

STRIDE = 16


class _StageClock:
    """Optional wall-clock log of the stages of seed_pseudo_gt (AS_STAGE_LOG=1): a device sync at every mark, so
    only for diagnosis."""

    def __init__(self):
        import os
        self.on = os.environ.get("AS_STAGE_LOG", "0") == "1"
        self.acc = {}
        self.t = None

    def start(self):
        if self.on:
            torch.cuda.synchronize()
            import time
            self.t = time.perf_counter()

    def mark(self, name):
        if self.on:
            import time
            torch.cuda.synchronize()
            now = time.perf_counter()
            self.acc[name] = self.acc.get(name, 0.0) + (now - self.t)
            self.t = now

    def report(self, steps=1):
        return {k: round(v / steps * 1e3, 3) for k, v in self.acc.items()}


CLOCK = _StageClock()


class _HostDrawsNeeded(Exception):
    """reference-RNG mode, device draws: an image needs one of the reference's host-side refill branches."""

# Per-thread CPU generator for the sampling draws: None = torch's global generator (the reference's stream).  When the
# images of a batch are processed concurrently (seed_pseudo_gt, rng_mode "fast") every image gets its own generator,
# seeded from the global one in image order, so a run is still reproducible from torch.manual_seed().
import threading
_TLS = threading.local()


def _gen():
    return getattr(_TLS, "gen", None)


_COPY_STREAMS = {}
_COPY_STREAMS_LOCK = __import__("threading").Lock()


def _to_host_issue(t, side_stream=False):
    """Start the device -> pinned-host copy of `t`; returns (pinned tensor, event).  With `side_stream` the copy runs on
    a per-device copy stream behind an event of the current stream, so the kernels queued after it on the current stream
    do not wait for the PCIe transfer (the 3 MB pseudo-mask stack of an image takes ~0.1 ms)."""
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    cur = torch.cuda.current_stream()
    if side_stream:
        with _COPY_STREAMS_LOCK:                          # (the per-image worker threads of parallel_images share it)
            cs = _COPY_STREAMS.get(t.device)
            if cs is None:
                cs = _COPY_STREAMS[t.device] = torch.cuda.Stream(device=t.device)
        cs.wait_stream(cur)
        with torch.cuda.stream(cs):
            host.copy_(t, non_blocking=True)
        t.record_stream(cs)
        cur = cs
    else:
        host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(cur)
    return host, ev


def _to_host_finish(pending):
    host, ev = pending
    ev.synchronize()
    return host.numpy()


def _to_host_numpy(t):
    """Device tensor -> numpy through a PINNED staging tensor (torch's caching host allocator recycles it): a 3 MB
    pseudo-mask stack goes over PCIe at DMA speed instead of the ~2 GB/s of a pageable `.cpu()` (1.7 ms -> ~0.1 ms).
    The returned array keeps its buffer alive."""
    if not t.is_cuda:
        return t.numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


def to_device(values, device, dtype=None):
    """Host values (numpy array / CPU tensor / list) -> tensor on `device` WITHOUT a host sync: through a pinned staging
    tensor and a non-blocking copy (a pageable `torch.as_tensor(..., device=cuda)` waits for the stream to drain)."""
    t = torch.as_tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda" or t.numel() == 0:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def read_back(pieces):
    """Device tensors -> list of fp32 CPU tensors of the same shapes with ONE transfer and one host sync (values must be
    exact in fp32: small integers, bools, fp32 numbers)."""
    if not pieces:
        return []
    if not pieces[0].is_cuda:
        return [p_.detach().float().cpu() for p_ in pieces]
    flat = torch.from_numpy(_to_host_numpy(torch.cat([p_.detach().reshape(-1).float() for p_ in pieces])))
    out, off = [], 0
    for p_ in pieces:
        out.append(flat[off:off + p_.numel()].view(p_.shape))
        off += p_.numel()
    return out



# --------------------------------------------------------------------------------------------------
# host-side matching (stdroi:2237-2257; HungarianPointAssigner mmdet/core/bbox/assigners/
# hungarian_point_assigner.py:54-113, FocalLossCost / PointL1Cost match_cost.py:52-104)
# --------------------------------------------------------------------------------------------------
def hungarian_point_match(point_pred, cls_pred, gt_points, gt_labels, img_shape, cls_weight=1.0, reg_weight=10.0,
                          alpha=0.25, gamma=2.0, eps=1e-12):
    """Returns (pos_inds ascending [G'], matched gt index per pos_ind [G'])."""
    if gt_points.shape[0] == 0 or point_pred.shape[0] == 0:
        z = torch.zeros(0, dtype=torch.long, device=point_pred.device)
        return z, z
    cost = point_match_cost(point_pred, cls_pred, gt_points, gt_labels, img_shape, cls_weight, reg_weight, alpha, gamma, eps)
    rows, cols = hungarian_rows_cols(cost.detach().cpu().numpy())
    dev = point_pred.device
    return to_device(rows, dev, torch.long), to_device(cols, dev, torch.long)


def point_match_cost(point_pred, cls_pred, gt_points, gt_labels, img_shape, cls_weight=1.0, reg_weight=10.0, alpha=0.25,
                     gamma=2.0, eps=1e-12):
    """[T, G] matching cost (FocalLossCost + PointL1Cost, match_cost.py:52-104) on the inputs' device; no host sync."""
    img_h, img_w = img_shape[:2]
    p = cls_pred.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    # torch.cdist(p=1) of 2-d points, written out (two terms: the same fp32 sum) with the scale applied per axis
    gx, gy = gt_points[:, 0] / img_w, gt_points[:, 1] / img_h
    l1 = (point_pred[:, None, 0] - gx[None, :]).abs() + (point_pred[:, None, 1] - gy[None, :]).abs()
    return (pos[:, gt_labels] - neg[:, gt_labels]) * cls_weight + l1 * reg_weight


def hungarian_rows_cols(cost_np):
    """scipy's assignment on a host cost matrix -> (token indices ascending, their GT index), numpy int64."""
    from scipy.optimize import linear_sum_assignment
    rows, cols = linear_sum_assignment(cost_np)
    order = np.argsort(rows)
    return rows[order].astype(np.int64), cols[order].astype(np.int64)


# --------------------------------------------------------------------------------------------------
# small torch helpers (device agnostic; mirrored one-to-one from the cited reference lines)
# --------------------------------------------------------------------------------------------------
def _fill_in(idx, want):
    """stdroi:1147-1155 for row lists: cyclic repeat up to `want` rows."""
    assert idx.shape[0] != 0, "cannot fill from an empty selection"
    while idx.shape[0] < want / 2:
        idx = idx.repeat(want // idx.shape[0], *([1] * (idx.dim() - 1)))
    return torch.cat((idx, idx[: want - idx.shape[0]]), dim=0)


def _erode(x, k):
    """stdroi:145-146, 1182-1187 (min-pool via negated max-pool)."""
    shp = x.shape
    return (-F.max_pool2d(-x.reshape(1, -1, shp[-2], shp[-1]), k, 1, k // 2)).reshape(shp)


def _centre4(x, dy, dx):
    return x[..., dy::STRIDE, dx::STRIDE]


def _down16(x, presampled=False):
    """F.interpolate(x, size/16, mode='bilinear') for an exact factor 16: source coordinate 16*i + 7.5, i.e. the
    mean of pixels (16i+7, 16i+8) along each axis with weights 0.5/0.5 (products exact, one rounding per add,
    identical to ATen's two fma steps).  presampled: x = [4, ...] stacked (7,7),(7,8),(8,7),(8,8) samples."""
    if presampled:
        a, b, c, d = x[0], x[1], x[2], x[3]
    else:
        a, b, c, d = _centre4(x, 7, 7), _centre4(x, 7, 8), _centre4(x, 8, 7), _centre4(x, 8, 8)
    return 0.5 * (0.5 * a + 0.5 * b) + 0.5 * (0.5 * c + 0.5 * d)


def _minmax_maps(a):
    """stdroi:329-333 norm_attns."""
    flat = a.flatten(1)
    lo, hi = flat.min(1)[0][:, None, None], flat.max(1)[0][:, None, None]
    return (a - lo) / (hi - lo)


def rank_select(mask_flat, ranks):
    """k-th set pixel of each row of a [G, H*W] 0/1 mask, in raster order (= the order of .nonzero()):
    ranks [G,K] long (0-based) -> flat indices [G,K].  No host sync and no compaction pass (ops.rank_select)."""
    m = mask_flat if mask_flat.dtype == torch.uint8 else mask_flat.to(torch.uint8)
    if m.shape[1] % 16:                                    # the kernel reads 16-byte groups: zero-pad the tail
        m = F.pad(m, (0, 16 - m.shape[1] % 16))
    return ops.rank_select(m.contiguous(), ranks)


def rank_select_xy(mask_flat, ranks, W, yx=False):
    """rank_select returning the (x, y) [or (y, x)] grid coordinates of the selected pixels, int64 [G,K,2]; ranks beyond
    the population give pixel 0 (ops.rank_select_xy)."""
    m = mask_flat if mask_flat.dtype == torch.uint8 else mask_flat.to(torch.uint8)
    if m.shape[1] % 16:                                    # the kernel reads 16-byte groups: zero-pad the tail
        m = F.pad(m, (0, 16 - m.shape[1] % 16))
    return ops.rank_select_xy(m.contiguous(), ranks, W, yx)


def sample_point_grid(maps, num_points, thr, is_pos, gt_points=None):
    """stdroi:343-371.  Random draws come from torch's global CPU generator exactly like the reference
    (`torch.randint(n, shape)` per object, in object order); candidate counts cost ONE host sync for all
    objects and the drawn ranks are resolved on the device by rank_select (no .nonzero())."""
    G, H, W = maps.shape
    mask = (maps >= thr) if is_pos else (maps < thr)
    counts = mask.flatten(1).sum(1).tolist()
    if min(counts) < num_points:
        return _sample_point_grid_slow(maps, num_points, thr, is_pos, gt_points)
    ranks = []
    for n in counts:
        n_draw = len(range(0, n, n // num_points))
        ranks.append((torch.randint(n, (n_draw,), generator=_gen()) % n)[:num_points])
    flat = rank_select(mask.flatten(1), torch.stack(ranks).to(maps.device))
    return torch.stack((flat % W, flat // W), dim=-1)          # (x, y) = coords.flip(-1)


def sample_point_grid_multi(specs, num_points):
    """Several sample_point_grid calls with ONE host sync: specs = [(maps, thr, is_pos, gt_points), ...].  The draws
    are made spec by spec, object by object -- the same stream order as consecutive sample_point_grid calls."""
    masks = [((m >= thr) if pos else (m < thr)) for (m, thr, pos, _) in specs]
    if all(mk[0].numel() % 16 == 0 for mk in masks):
        counts = ops.mask_count(torch.cat([mk.flatten(1) for mk in masks])).tolist()      # one launch, one sync
    else:
        counts = torch.cat([mk.flatten(1).sum(1) for mk in masks]).tolist()
    out, off = [], 0
    for (maps, thr, pos, gtp), mask in zip(specs, masks):
        G, H, W = maps.shape
        cs = counts[off:off + G]
        off += G
        if min(cs) < num_points:
            out.append(_sample_point_grid_slow(maps, num_points, thr, pos, gtp))
            continue
        ranks = []
        for n in cs:
            n_draw = len(range(0, n, n // num_points))
            ranks.append((torch.randint(n, (n_draw,), generator=_gen()) % n)[:num_points])
        flat = rank_select(mask.flatten(1), torch.stack(ranks).to(maps.device))
        out.append(torch.stack((flat % W, flat // W), dim=-1))
    return out


def sample_points_from_cams(cams_lr, map_idx, minmax, gt_points, num_points, thr_bg=0.1, thr_fg=0.2):
    """The three seed samplings of stdroi:1003-1007 (background / foreground / shared background on norm_attns of
    the selected CAMs) from the LOW-resolution maps: one fused launch builds the candidate masks and counts
    (ops.cam_sample_masks; the upsampled maps are never written), one host sync reads the counts, the draws follow
    the reference's order (spec by spec, object by object) and rank_select resolves them on the device.
    cams_lr [M,hp,wp], map_idx [G] int32 (rows of cams_lr), minmax [M,2].  Returns (pts_bg, pts_fg, pts_supp)."""
    G = map_idx.shape[0]
    masks, counts_dev = ops.cam_sample_masks(cams_lr, map_idx, minmax, thr_bg, thr_fg, STRIDE)
    counts = counts_dev.tolist()                                                     # the one sync
    H, W = masks.shape[-2:]
    nm_cache = []

    def nm():                                               # only the rare short-of-candidates branches need the maps
        if not nm_cache:
            idx = map_idx.long()
            dummy = torch.zeros(G, 2, device=cams_lr.device)
            up = ops.cam_boxes(cams_lr[idx].contiguous(), dummy, 0.5, 0.5, STRIDE, True)[2]
            lo, hi = minmax[idx, 0][:, None, None], minmax[idx, 1][:, None, None]
            nm_cache.append((up - lo) / (hi - lo))
        return nm_cache[0]

    out = []
    for (lo_, hi_, thr, pos, gtp, mean) in ((0, G, thr_bg, False, None, False), (G, 2 * G, thr_fg, True, gt_points, False),
                                           (2 * G, 2 * G + 1, thr_bg, False, None, True)):
        cs = counts[lo_:hi_]
        if min(cs) < num_points:
            maps = nm().mean(0, keepdim=True) if mean else nm()
            out.append(_sample_point_grid_slow(maps, num_points, thr, pos, gtp))
            continue
        ranks = []
        for n in cs:
            n_draw = len(range(0, n, n // num_points))
            ranks.append((torch.randint(n, (n_draw,), generator=_gen()) % n)[:num_points])
        flat = ops.rank_select(masks[lo_:hi_].flatten(1), torch.stack(ranks).to(masks.device))
        out.append(torch.stack((flat % W, flat // W), dim=-1))
    return out


# ---- fast-RNG mode: draws on the device, no host sync -------------------------------------------------------------
# rng_mode="fast" does not reproduce the reference's generator stream, only its distributions, so the draws need not
# happen on the host: the candidate counts stay on the device, uniform numbers come from a device generator and
# rank = floor(u * n).  The rare branches that need host logic (fewer candidates than points, tiny candidate sets) only
# raise a device FLAG; the caller reads it with a readback it needs anyway and redoes that image on the synchronous
# path (same distributions, different draws).

def sample_points_from_cams_nosync(cams_lr, map_idx, minmax, num_points, gen, thr_bg=0.1, thr_fg=0.2, flag=None, u=None):
    """sample_points_from_cams without the count readback.  Returns (pts_bg, pts_fg, pts_supp, flag) with flag a
    device flag: some candidate set is smaller than num_points (the reference's refill branches, stdroi:354-364).
    `flag` (a zeroed int32 [1] slot of the caller): the ranks are then derived inside the selection kernel from the
    populations it counts anyway (ops.rank_draw_xy: one launch pair instead of the eight tensor ops below).
    `u` [2G+1, num_points]: the uniform numbers, if the caller drew them already (one draw for the batch)."""
    G = map_idx.shape[0]
    in_kernel = flag is not None and (cams_lr.shape[-1] * cams_lr.shape[-2] * STRIDE * STRIDE) % 16 == 0
    masks, counts = ops.cam_sample_masks(cams_lr, map_idx, minmax, thr_bg, thr_fg, STRIDE, want_counts=not in_kernel)
    W = masks.shape[-1]
    if u is None:
        u = torch.rand(2 * G + 1, num_points, device=masks.device, generator=gen)
    if in_kernel:
        # also the token index of every drawn pixel, rows rotated to [fg objects, shared background, bg objects]: the
        # gather index of the seed features (seed_features' `// 16`, clamp and cat chain)
        pts, pidx = ops.rank_draw_xy(masks.flatten(1), num_points, W, u=u, flag=flag, patch=(None, STRIDE, W // STRIDE, 0, G))
        return pts[:G], pts[G:], pidx, flag.reshape(())
    n = counts.float()[:, None]
    ranks = torch.minimum((u * n).to(torch.int32), (counts[:, None] - 1).clamp(min=0))
    pts = rank_select_xy(masks.flatten(1), ranks, W)                           # (x, y) of every drawn candidate
    return pts[:G], pts[G:2 * G], pts[2 * G:], (counts < num_points).any()


def sample_points_from_cams_mt(cams_lr, map_idx, minmax, num_points, mt_state, thr_bg=0.1, thr_fg=0.2):
    """sample_points_from_cams in the REFERENCE's generator stream without the count readback: the candidate counts stay
    on the device and ops.mt_sample_ranks advances torch's own mt19937 engine (`mt_state`, attentionshift_amd/mt19937.py)
    by exactly the words the reference's `torch.randint(n, (n_draw,))` calls consume, spec by spec, object by object.
    Returns (pts_bg, pts_fg, pts_supp, flag); flag: a candidate set smaller than num_points (host path)."""
    G = map_idx.shape[0]
    masks, counts = ops.cam_sample_masks(cams_lr, map_idx, minmax, thr_bg, thr_fg, STRIDE)
    W = masks.shape[-1]
    ranks, flag = ops.mt_sample_ranks(mt_state, counts.to(torch.int32).contiguous(), num_points)
    pts = rank_select_xy(masks.flatten(1), ranks, W)
    return pts[:G], pts[G:2 * G], pts[2 * G:], flag[0] != 0


def mask_points_mt(pend, num_gt, mt_state):
    """mask_points_finish in the reference's generator stream without the count readback: ranks = torch.randperm(n)[:num_gt]
    per object from the device copy of torch's engine (ops.mt_perm_ranks).  flag: an object with fewer than num_gt
    candidates (the reference's fill-in / empty branches, stdroi:449-455: host path)."""
    pos, neg = pend["pos"], pend["neg"]
    G, H, W = pend["shape"]
    counts2 = pend["counts"].to(torch.int32).contiguous()
    ranks, flag = ops.mt_perm_ranks(mt_state, counts2, num_gt)
    n_pos = counts2[:, :1]
    is_pos = ranks < n_pos
    zero = torch.zeros_like(ranks)
    xy_pos = rank_select_xy(pos.flatten(1), torch.where(is_pos, ranks, zero), W)
    xy_neg = rank_select_xy(neg.flatten(1), torch.where(is_pos, zero, ranks - n_pos), W)
    coords = torch.where(is_pos[..., None], xy_pos, xy_neg).float()
    return coords, is_pos, flag[0] != 0


def mask_points_nosync(pend, num_gt, gen, int_flag=False, flag=None, u=None):
    """mask_points_finish without the count readback: num_gt DISTINCT uniform ranks among the n = n_pos + n_neg
    candidates of each object (the first num_gt entries of a random permutation, stdroi:447) = the first num_gt
    distinct values of 32 uniform draws.  flag: an object with fewer than 4*num_gt candidates (the host path's
    randperm / refill / empty branches) or, with probability < 1e-12, too few distinct draws."""
    pos, neg = pend["pos"], pend["neg"]
    G, H, W = pend["shape"]
    if u is None:                                          # (else: the caller's share of one draw for the batch)
        u = torch.rand(G, 32, device=pos.device, generator=gen)
    # counts [G, 2] = (n_pos, n_neg), read in place; `flag`: a zeroed int32 [1] slot of the caller's flag vector
    rank_pos, rank_neg, is_pos, flag = ops.draw_distinct(pend["counts"], u, num_gt, flag=flag)
    xy_pos = rank_select_xy(pos.flatten(1), rank_pos, W)
    xy_neg = rank_select_xy(neg.flatten(1), rank_neg, W)
    coords = torch.where(is_pos[..., None], xy_pos, xy_neg).float()
    return coords, is_pos, (flag[0] if int_flag else flag[0] != 0)


def grid_seed_nosync(mask, count_dev, n_points=20, flag=None, patch=None):
    """grid_seed_finish without the count readback, for objects with at least n_points positives (stdroi:1790-1792:
    every (n // n_points)-th positive in raster order).  flag: an object with fewer (refill / box-centre branches).
    With `flag` (a zeroed int32 [1] slot) ranks and flag come out of the selection kernel itself (ops.rank_draw_xy)."""
    G, hp, wp = mask.shape
    if flag is not None and (hp * wp) % 16 == 0 and mask.dtype == torch.uint8:
        if patch is not None:                              # (out rows, base): also the seeds' token ids, for ONE batched gather
            return ops.rank_draw_xy(mask.flatten(1), n_points, wp, flag=flag, yx=True,
                                    patch=(patch[0], 1, wp, patch[1], 0))[0], flag.reshape(())
        return ops.rank_draw_xy(mask.flatten(1), n_points, wp, flag=flag, yx=True), flag.reshape(())
    step = (count_dev // n_points).clamp(min=1)
    ranks = torch.arange(n_points, device=mask.device, dtype=torch.int32)[None, :] * step[:, None].int()
    return rank_select_xy(mask.flatten(1), ranks, wp, yx=True), (count_dev < n_points).any()


def _sample_point_grid_slow(maps, num_points, thr, is_pos, gt_points=None):
    """The rare branches of stdroi:354-364 (fewer candidates than points), one object at a time."""
    out = []
    for g, m in enumerate(maps):
        factor = 1.0
        coords = ((m >= thr) if is_pos else (m < thr)).nonzero()
        n = coords.shape[0]
        if n < num_points:
            if is_pos:
                out.append(torch.cat((coords, gt_points[g].repeat(num_points - n, 1)), dim=0))
                continue
            while n < num_points:
                factor *= 2
                coords = (m < thr * factor).nonzero()
                n = coords.shape[0]
        n_draw = len(range(0, n, n // num_points))
        pick = (torch.randint(n, (n_draw,), generator=_gen()) % n).to(coords.device)
        out.append(coords[pick][:num_points])
    return torch.stack(out).flip(-1)


def seed_features(point_xy, feat_chw):
    """stdroi:335-338: mean feature under the sampled pixels.  point_xy [G',K,2] (x,y)."""
    C, hp, wp = feat_chw.shape
    py = (point_xy[..., 1].long() // STRIDE).clamp(0, hp)
    px = (point_xy[..., 0].long() // STRIDE).clamp(0, wp)
    return feat_chw.permute(1, 2, 0)[py, px].mean(dim=1)


def feature_tokens(vit_feat):
    """[B,C,hp,wp] (any strides) -> fp32 [B,Np,C] whose per-image slices are contiguous: the layout every kernel of
    this path reads.  The reference's caller builds vit_feat as a permuted VIEW of the token-major last_feat
    (two_stage_point_align.py:77); for such an input this is a view too (no copy).  Anything else is transposed once
    per batch here instead of once per consumer."""
    B, C = vit_feat.shape[:2]
    t = vit_feat.flatten(2).transpose(1, 2)
    if t.dtype != torch.float32:
        return t.to(torch.float32, memory_format=torch.contiguous_format)
    if t.stride(2) != 1 or t.stride(1) != C:
        t = t.contiguous()
    return t


def candidate_masks(map_fg, map_bg, crops, pos_thr, neg_thr, corr_size):
    """Candidate pixels of stdroi:442-443 for every object at once, as full-size byte masks that are zero
    outside each object's crop: fg = erode(map_fg > cropmax*pos_thr, corr_size), bg = map_bg > cropmax*neg_thr."""
    pos, cp = ops.crop_threshold_erode(map_fg, crops, pos_thr, True, corr_size)
    neg, cn = ops.crop_threshold_erode(map_bg, crops, neg_thr, True, 1)
    return pos, neg, cp, cn


def first_of_randperm(n, k, mode="reference"):
    """The first k entries of a uniformly random permutation of range(n), from torch's global CPU generator.
    "reference": literally torch.randperm(n)[:k] (stdroi:447) -- the same stream as the reference, but O(n) host
    work (tens of ms for the 1e5..1e6 candidate pixels of a 1024^2 crop).
    "fast": the same distribution (k distinct indices in uniformly random order) by rejection from O(k) draws."""
    if mode == "reference" or n <= 4 * k:
        return torch.randperm(n, generator=_gen())[:k]
    seen, out = set(), []
    while len(out) < k:
        for v in torch.randint(n, (2 * k,), generator=_gen()).tolist():
            if v not in seen:
                seen.add(v)
                out.append(v)
                if len(out) == k:
                    break
    return torch.tensor(out, dtype=torch.long)


def mask_sample_points(map_fg, map_bg, rois, pos_thr, neg_thr, num_gt, corr_size, rng_mode="reference"):
    """stdroi:1980-1993 + 433-461 for all objects: coords [G,num_gt,2] (x,y) float, labels [G,num_gt] bool.
    One host sync (candidate counts); `torch.randperm(n)` per object from the global CPU generator in object
    order like the reference; the drawn ranks index the concatenation [fg candidates, bg candidates] in raster
    order inside the crop, resolved on the device by rank_select."""
    return mask_points_finish(mask_points_issue(map_fg, map_bg, rois, pos_thr, neg_thr, corr_size), num_gt, rng_mode)


def mask_points_issue(map_fg, map_bg, rois, pos_thr, neg_thr, corr_size):
    """Device half of mask_sample_points: candidate masks and their counts are queued, nothing is read back."""
    crops = rois.int().contiguous()                      # stdroi:1981: rois[i].int().tolist()
    pos, neg, cp, cn = candidate_masks(map_fg.contiguous(), map_bg.contiguous(), crops, pos_thr, neg_thr, corr_size)
    return dict(pos=pos, neg=neg, cp=cp, crops=crops, counts=torch.stack((cp, cn), dim=1), shape=tuple(map_fg.shape))


def mask_points_and_pseudo_issue(map_fg, map_bg, rois, pos_thr, neg_thr, corr_size, mask_thr, crops=None):
    """mask_points_issue plus the pseudo mask of stdroi:2357 from ONE fused call (ops.mask_candidates): the three
    thresholds share their maxima pass and their thresholding pass.  Returns (pending dict, pseudo mask uint8)."""
    if crops is None:
        crops = rois.int().contiguous()                  # stdroi:1981: rois[i].int().tolist()
    pos, neg, pseudo, counts = ops.mask_candidates(map_fg.contiguous(), map_bg.contiguous(), crops, pos_thr, neg_thr,
                                                   mask_thr, corr_size)
    return dict(pos=pos, neg=neg, cp=counts[0], crops=crops, counts=counts[:2].t(), shape=tuple(map_fg.shape)), pseudo


def mask_points_finish(pend, num_gt, rng_mode="reference"):
    """Host half of mask_sample_points: ONE sync (the counts), the draws, and the rank lookups."""
    pos, neg, cp, crops = pend["pos"], pend["neg"], pend["cp"], pend["crops"]
    G, H, W = pend["shape"]
    dev = pos.device
    counts = pend["counts"].tolist()
    ranks, empty = [], []
    for g in range(G):
        n = counts[g][0] + counts[g][1]
        pick = first_of_randperm(n, num_gt, rng_mode)
        if pick.shape[0] < num_gt:
            if pick.shape[0] == 0:
                empty.append(g)
                pick = torch.zeros(num_gt, dtype=torch.long)
            else:
                pick = _fill_in(pick, num_gt)     # NB: the reference's 1-D repeat bug (2-D index) is not reproduced
        ranks.append(pick)
    ranks = torch.stack(ranks).to(dev)
    n_pos = cp.long()[:, None]
    is_pos = ranks < n_pos
    idx_pos = rank_select(pos.flatten(1), torch.where(is_pos, ranks, torch.zeros_like(ranks)))
    idx_neg = rank_select(neg.flatten(1), torch.where(is_pos, torch.zeros_like(ranks), ranks - n_pos))
    flat = torch.where(is_pos, idx_pos, idx_neg).clamp(max=H * W - 1)
    coords = torch.stack((flat % W, flat // W), dim=-1).float()
    labels = is_pos
    for g in empty:                                       # stdroi:451-455: -1 points, later ignored
        x0, y0 = crops[g, 0].float(), crops[g, 1].float()
        coords[g, :, 0] = -1 + x0
        coords[g, :, 1] = -1 + y0
        labels[g] = False
    return coords, labels


def grid_seed_coords(maps, rois, thr=0.35, n_points=20):
    """stdroi:1784-1810: (y,x) patch coords of n_points grid-strided positives per object."""
    return grid_seed_finish(*grid_seed_issue(maps, thr), rois, n_points)


def grid_seed_issue(maps, thr=0.35):
    mask = maps >= thr
    return mask, mask.flatten(1).sum(1)


def grid_seed_finish(mask, count_dev, rois, n_points=20):
    G, hp, wp = mask.shape
    counts = count_dev.tolist()                              # the one host sync (was one .nonzero() per object)
    ranks = []
    for n in counts:                                         # which of the n positives (raster order) are taken
        if n >= n_points:
            r = torch.arange(0, n, step=n // n_points)[:n_points]
        elif n > 0:
            r = _fill_in(torch.arange(n), n_points)
        else:
            r = torch.zeros(n_points, dtype=torch.long)      # placeholder, replaced by the box centre below
        ranks.append(r)
    flat = rank_select(mask.flatten(1), torch.stack(ranks).to(mask.device))
    coords = torch.stack((flat // wp, flat % wp), dim=-1)    # (y, x), as .nonzero() rows
    for g, n in enumerate(counts):
        if n == 0:
            coords[g] = ((rois[g][:2] + rois[g][2:]) // (2 * STRIDE)).long().view(1, 2).flip(1).repeat(n_points, 1)
    return coords


_CONSTS = {}
_CONSTS_LRU = None          # count-dependent entries (one per distinct tuple of per-image object counts), bounded
_CONSTS_LRU_MAX = 64


def _const_tensor(key, device, make, make_device=None):
    """Small index tensors (arange, owners of padded slots, per-object metadata), reused by later calls and streams.

    Keys that depend only on fixed sizes (make_device None): built once on the host, copied, the creating stream drained once
    so that no other stream can read early; kept for the life of the process.
    Keys that depend on the BATCH's per-image object counts (make_device given): on real data almost every step brings a new
    tuple, so a miss must cost neither a host sync nor a pageable copy, and the cache must not grow with the dataset
    (ADVICE r05).  make_device() builds the tensor with device-side fills on the CURRENT stream (scalars travel as kernel
    arguments); the entry carries an event that later users on other streams wait for ON THE DEVICE until it has completed
    once; at most _CONSTS_LRU_MAX such entries are kept (least recently used dropped)."""
    global _CONSTS_LRU
    k = (key, str(device))
    if make_device is None or torch.device(device).type != "cuda":
        t = _CONSTS.get(k)
        if t is None:
            t = make().to(device)
            if t.is_cuda:
                torch.cuda.current_stream(t.device).synchronize()
            _CONSTS[k] = t
        return t
    if _CONSTS_LRU is None:
        import collections
        _CONSTS_LRU = collections.OrderedDict()
    ent = _CONSTS_LRU.get(k)
    if ent is None:
        t = make_device()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        ent = [t, ev]
        _CONSTS_LRU[k] = ent
        while len(_CONSTS_LRU) > _CONSTS_LRU_MAX:
            _CONSTS_LRU.popitem(last=False)
        return t
    _CONSTS_LRU.move_to_end(k)
    if ent[1] is not None:
        if ent[1].query():
            ent[1] = None                                    # landed: visible to every stream from now on
        else:
            torch.cuda.current_stream(ent[0].device).wait_event(ent[1])
    return ent[0]


def _token_blocks_ok(t):
    """[B,Np,C] fp32 whose per-image blocks are contiguous rows of C floats, images a whole number of rows apart (a
    contiguous tensor, or the reference caller's view of last_feat without the cls row): what as_cosine_shift_strided reads
    in place and what _token_rows can index as one 2-D row table."""
    B, Np, C = t.shape
    return (t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) == C and t.data_ptr() % 16 == 0
            and (B == 1 or (t.stride(0) >= Np * C and t.stride(0) % C == 0 and t.stride(0) % 4 == 0)))


def _token_rows(t):
    """Row table [(B - 1) * rows_per_image + Np, C] over the storage of a _token_blocks_ok tensor: token n of image i is row
    i * _token_row_stride(t) + n."""
    B, Np, C = t.shape
    rpi = t.stride(0) // C if B > 1 else Np
    return t.as_strided(((B - 1) * rpi + Np, C), (C, 1))


def _token_row_stride(t):
    return t.stride(0) // t.shape[2] if t.shape[0] > 1 else t.shape[1]


def _unit(x):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-8)


def filter_parts(sim, fg_inter, pos_thr=0.85):
    """stdroi:265-275 filter_maps."""
    support = (sim > 0.8).to(sim.dtype)
    score = (fg_inter[:, None] * support).sum(dim=[-2, -1]) / support.sum(dim=[-2, -1]).clamp(1e-6)
    return score >= pos_thr


def merge_plan(keep, link):
    """Greedy upper-triangular grouping of stdroi:278-294 on the host (numpy, a few hundred bytes).
    keep [P] bool, link [P,P] bool (cos >= thr) -> list of member-index lists over the ORIGINAL prototype ids."""
    idx = np.flatnonzero(keep)
    sub = np.triu(link[np.ix_(idx, idx)]).astype(bool)
    groups = []
    for i in range(len(idx)):
        row = sub[i].copy()
        if row.any():
            groups.append(idx[row].tolist())
        sub[row] = False                                   # sim_triu[weight > 0] *= 0
    return groups


def merge_parts(prot, keep, thr, extra=None):
    """stdroi:278-294 merge_maps for all objects: prot [G,P,C], keep [G,P] ->
    (list over objects of merged prototypes [m_g, C] or []).  One host sync for the whole image; `extra` (a list that
    holds device bool scalars) rides on the same transfer and is replaced by its host values."""
    G, P, C = prot.shape
    u = _unit(prot)
    link = (u @ u.transpose(1, 2)) >= thr
    packed = torch.cat((keep[:, None, :], link), dim=1)                      # [G, 1+P, P]
    if extra:
        tail = torch.stack([e.reshape(()) for e in extra]).to(packed.dtype)
        both = torch.cat((packed.flatten(), tail)).cpu().numpy()
        extra[:] = [bool(v) for v in both[packed.numel():]]
        host = both[:packed.numel()].reshape(G, 1 + P, P)
    else:
        host = packed.cpu().numpy()
    plans = [merge_plan(host[g, 0], host[g, 1:]) for g in range(G)]
    m_max = max((len(p) for p in plans), default=0)
    if m_max == 0:
        return [[] for _ in range(G)]
    wgt = np.zeros((G, m_max, P), dtype=np.float32)
    for g, plan in enumerate(plans):
        for i, members in enumerate(plan):
            wgt[g, i, members] = 1.0
    wgt = torch.from_numpy(wgt).to(prot.device)
    merged = torch.bmm(wgt, prot) / (wgt.sum(-1, keepdim=True) + 1e-8)        # matmul(weight, prot) / (sum + 1e-8)
    return [merged[g, :len(plans[g])] if plans[g] else [] for g in range(G)]


def part_similarity(prot_list, feat_chw):
    """stdroi:297-301 cal_similarity for every object: one similarity pass per <= 32 merged prototypes (norms of both
    operands inside the kernel; no normalised copy of the feature map)."""
    C, hp, wp = feat_chw.shape
    sizes = [0 if isinstance(p, list) else p.shape[0] for p in prot_list]
    if sum(sizes) == 0:
        return [torch.zeros(0, 0) for _ in prot_list]
    allp = torch.cat([p for p in prot_list if not isinstance(p, list)]).contiguous()
    feat_tok = feat_chw.flatten(1).t().contiguous()                      # a view when feat_chw views token-major storage
    sim = torch.cat([ops.refine_similarity(feat_tok, allp[o:o + 32], None, 0, 0, 1.0, False, hp, wp)[0][0]
                     for o in range(0, allp.shape[0], 32)]).reshape(-1, hp, wp)
    out, off = [], 0
    for n in sizes:
        out.append(sim[off:off + n] if n else torch.zeros(0, 0))
        off += n
    return out


def part_centers(maps, rois, obj_label, feat_chw, num_max_keep=50, num_max_obj=3):
    """stdroi:222-262 get_center_coord_with_feat (same eight outputs, same order).  The per-part statistics
    (peak centroid, >0.9 area, inside-box test) are computed for all parts at once on the device and read
    back in one transfer; the visiting order / cap logic then runs on those few numbers."""
    dev = rois[0].device if len(rois) else feat_chw.device
    split = [0 for _ in range(len(maps))]
    sizes = [int(m.shape[0]) for m in maps]
    empty = ([torch.zeros(0, 2, dtype=rois[0].dtype, device=dev), torch.zeros(0, dtype=obj_label[0].dtype, device=dev)],
             [], [], [], split, torch.zeros(0, 2, dtype=rois[0].dtype, device=dev),
             torch.zeros(0, dtype=obj_label[0].dtype, device=dev), torch.zeros(0, dtype=torch.long, device=dev))
    if sum(sizes) == 0:
        return empty
    allm = torch.cat([m for m in maps if m.shape[0] > 0])                     # [M, hp, wp]
    owner = torch.cat([torch.full((n,), g, dtype=torch.long) for g, n in enumerate(sizes)]).to(dev)
    c, cyx, area, inside = ops.part_stats(allm, rois, owner, STRIDE)         # one launch for all parts
    cy, cx = cyx[:, 0], cyx[:, 1]
    host = torch.stack((area.float(), inside.float()), dim=1).cpu().numpy()    # the one sync
    chosen, off = [], 0
    for g, n in enumerate(sizes):
        if n == 0:
            continue
        order = np.argsort(-host[off:off + n, 0], kind="stable")               # argsort(descending, stable)
        for i in range(n):
            if i > num_max_obj:
                break
            j = off + int(order[i])
            if host[j, 1] > 0:
                chosen.append(j)
                split[g] += 1
        off += n
    if not chosen:
        return empty
    sel = torch.as_tensor(chosen, device=dev, dtype=torch.long)
    coords, labels = c[sel], obj_label[owner[sel]]
    feats = feat_chw[:, cy[sel], cx[sel]].t()
    coords_org, labels_org = coords.clone(), labels.clone()
    coord_split, feats_split = list(coords.split(split, dim=0)), list(feats.split(split, dim=0))
    if coords.shape[0] > num_max_keep:
        pick = torch.randperm(coords.shape[0], device=coords.device)[:num_max_keep]
        coords, labels = coords[pick], labels[pick]
    return ([coords, labels], coord_split, feats_split, feats, split, coords_org, labels_org, owner[sel])


def median_area_selector(boxes_per_img, labels_per_img=None, roi_feature_map=None):
    """Stand-in for the trainable MIL head's choice (mae_bbox_head_mil.py:140-169): per object the roll-out
    depth whose CAM box has the median area.  boxes [G,Lc,4] -> index [G]."""
    if len({b.shape[1] for b in boxes_per_img}) == 1:        # one pass for the whole batch (same launches as one image)
        b = torch.cat(list(boxes_per_img)) if len(boxes_per_img) > 1 else boxes_per_img[0]
        area = (b[..., 2] - b[..., 0]).clamp(min=0) * (b[..., 3] - b[..., 1]).clamp(min=0)
        pick = area.argsort(dim=1, stable=True)[:, (b.shape[1] - 1) // 2]
        return list(pick.split([x.shape[0] for x in boxes_per_img]))
    out = []
    for b in boxes_per_img:
        area = (b[..., 2] - b[..., 0]).clamp(min=0) * (b[..., 3] - b[..., 1]).clamp(min=0)
        order = area.argsort(dim=1, stable=True)
        out.append(order[:, (b.shape[1] - 1) // 2])
    return out


# --------------------------------------------------------------------------------------------------
@HEADS.register_module(name=["AttnShiftRoIHead", "StandardRoIHeadMaskPointSampleDeformAttnReppoints"])
class AttnShiftRoIHead(nn.Module):
    def __init__(self, mil_head=None, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None,
                 mask_head=None, shared_head=None, mae_head=None, bbox_rec_head=None, train_cfg=None, test_cfg=None,
                 visualize=False, epoch=0, epoch_semantic_centers=0, num_semantic_points=3, semantic_to_token=False,
                 pca_dim=128, mean_shift_times_local=10, reppoints_head=None, num_reppoints_head=1,
                 layer_selector=None, rng_mode="reference", parallel_images=False):
        super().__init__()
        self.train_cfg = _ns(train_cfg)
        self.test_cfg = _ns(test_cfg)
        self.sub_cfgs = dict(mil_head=mil_head, bbox_roi_extractor=bbox_roi_extractor, bbox_head=bbox_head,
                             mask_roi_extractor=mask_roi_extractor, mask_head=mask_head, shared_head=shared_head,
                             mae_head=mae_head, bbox_rec_head=bbox_rec_head, reppoints_head=reppoints_head)
        bh = dict(bbox_head or {})
        # the three attributes seed_pseudo_gt reads off self.bbox_head (stdroi:2261, 2287-2288)
        self.bbox_head = types.SimpleNamespace(cam_layer=bh.get("cam_layer", 7), seed_thr=bh.get("seed_thr", 0.2),
                                               seed_multiple=bh.get("seed_multiple", 0.5),
                                               num_classes=bh.get("num_classes", 20))
        # the trainable box / mask heads are built when their configs carry construction arguments (in_channels); the
        # attribute-only form keeps the pseudo-label path usable on its own
        self.mask_head = None
        if isinstance(bbox_head, dict) and bbox_head.get("type") == "MAEBoxHeadRec" and "in_channels" in bbox_head:
            from .mae_heads import MAEBoxHeadRec
            self.bbox_head = MAEBoxHeadRec(**{k: v for k, v in bbox_head.items() if k != "type"})
        if isinstance(mask_head, dict) and mask_head.get("type") == "MAEMaskHeadPointSup" and "in_channels" in mask_head:
            from .mae_heads import MAEMaskHeadPointSup
            self.mask_head = MAEMaskHeadPointSup(**{k: v for k, v in mask_head.items() if k != "type"})
        self.with_mil = mil_head is not None
        self.with_mask = mask_head is not None
        self.with_bbox = bbox_head is not None
        # the depth selector: the caller's, else the MIL head of the config when it carries its construction
        # arguments (it then needs `roi_feature_map` in seed_pseudo_gt), else the median-area stand-in
        self.mil_head = None
        if layer_selector is None and isinstance(mil_head, dict) and mil_head.get("type") == "MAEBoxHeadMIL" \
                and "in_channels" in mil_head:
            from .mil_head import MAEBoxHeadMIL, MILLayerSelector
            rl = dict((bbox_roi_extractor or {}).get("roi_layer", {}))
            self.mil_head = MAEBoxHeadMIL(**{k: v for k, v in mil_head.items() if k != "type"})
            self._mil_selector = MILLayerSelector(self.mil_head, rl.get("output_size", 7),
                                                  (bbox_roi_extractor or {}).get("featmap_strides", [STRIDE])[0],
                                                  rl.get("sampling_ratio", 0))
            layer_selector = self._select_with_mil
        self.layer_selector = layer_selector or median_area_selector
        assert rng_mode in ("reference", "fast")
        self.rng_mode = rng_mode          # "reference": the reference's exact torch RNG stream; "fast": O(k) draws
        # optional: one host thread + HIP stream per image (only with rng_mode "fast").  Off by default: after the host
        # syncs of a chain were merged the chains are interpreter-bound, and two GIL-sharing threads measure within
        # +-10 % of the sequential loop depending on the host (tools/phase_times.py)
        self.parallel_images = parallel_images
        self.batch_mean_shift = True              # one as_cosine_shift call for all images of a batch
        self.rollout_matched_only = True          # roll out only the matched point tokens' rows (the only ones consumed)
        self.image_streams = True                 # one HIP stream per image in the single-threaded fast-RNG path
        self.device_draws = True                  # draws on the device, no readback before the merge plan: fast mode from a
        #                                           device generator, reference mode from torch's own engine (csrc/mt19937.hip)
        self.rng_stats = dict(device_calls=0, host_redos=0)   # reference mode: calls drawn on the device / repeated on the host
        self.part_slots = 8                       # merged-part slots per object carried by the one-readback merge stage
        # sync-free path: the batched mean shift waits for the images' grid seeds only, so each image's mask-point / pseudo-mask
        # kernels (queued behind the seeds on the image's stream) run under it.  False: wait for the whole image streams
        # (bench.py times the affinity kernels alone that way)
        self.overlap_mask_work = True
        self._dev_gens = {}
        self._pool, self._streams = None, []
        # parity tests set this to a list: every image's sampled refinement points and grid seeds are appended to it
        # (the fast RNG mode draws on the device, so a checker needs the draws to re-run the chain on the same samples)
        self.capture = None
        self.ranks = None                         # dist.Ranks of a multi-rank run: loss normalisers are averaged over ranks
        self.visualize = visualize
        self.epoch, self.epoch_semantic_centers = epoch, epoch_semantic_centers
        self.num_semantic_points = num_semantic_points
        self.semantic_to_token, self.pca_dim = semantic_to_token, pca_dim
        self.mean_shift_times_local = mean_shift_times_local
        self.num_reppoints_head = num_reppoints_head

    def _select_with_mil(self, boxes_per_img, labels_per_img, roi_feature_map):
        """stdroi:2308-2312: the MIL head on the RoI-aligned stride-16 features; without a feature map (callers that
        only exercise the pseudo-label path) the median-area stand-in."""
        if roi_feature_map is None:
            return median_area_selector(boxes_per_img, labels_per_img, roi_feature_map)
        return self._mil_selector(boxes_per_img, labels_per_img, roi_feature_map)

    def _roi_extract(self, x, rois, which="bbox_roi_extractor"):
        """SingleRoIExtractor with one stride-16 level (attnshift_voc12aug.py:64-68): RoIAlign on x[0]."""
        from .mil_head import roi_align
        cfg = dict(self.sub_cfgs.get(which) or {})
        rl = dict(cfg.get("roi_layer", {}))
        fmap = x[0] if isinstance(x, (list, tuple)) else x
        return roi_align(fmap.float(), rois, rl.get("output_size", 7), 1.0 / cfg.get("featmap_strides", [STRIDE])[0],
                         rl.get("sampling_ratio", 0), True)

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                      vit_feat=None, img=None, point_init=None, point_cls=None, point_reg=None, imgs_whwh=None,
                      attns=None, gt_points=None, gt_points_labels=None, mask_point_labels=None, mask_point_coords=None,
                      semantic_centers=None, semantic_centers_split=None, sc_corres_gts=None, generator=None, **kwargs):
        """stdroi:2513-2727, the shipped configuration (no RepPoints heads, no MAE reconstruction head): the point-token
        loss, IoU assignment + random sampling of the proposals against the PSEUDO boxes, the box branch on the 7x7
        RoI features and the point-supervised mask branch on the positives' features.  `gt_bboxes` / `gt_labels` /
        `mask_point_*` / `semantic_centers_split` are what seed_pseudo_gt returned (two_stage_point_align.py:95-135)."""
        from . import assign as A
        from .mask_targets import mask_point_targets, point_sample
        from .point_loss import point_matches, point_token_loss
        if not isinstance(self.bbox_head, nn.Module):
            raise RuntimeError("forward_train needs the box head built from its config (bbox_head with in_channels)")
        num_imgs = len(img_metas)
        rcnn = self.train_cfg
        losses = {}
        pa = _get(rcnn, "point_assigner", None)
        asg = dict(_get(rcnn, "assigner", None) or {})
        smp = dict(_get(rcnn, "sampler", None) or {})
        with_mask = self.mask_head is not None and mask_point_coords is not None
        shapes = [m["img_shape"] for m in img_metas]
        # Everything whose SHAPE depends on data is decided on the host from ONE readback: the Hungarian cost matrices of
        # the point tokens, the proposals' IoU assignment (the sampler permutes `randperm(count)`) and the mask points'
        # labels (the negatives of an object are ragged).  The device work in between is queued without a sync.
        pieces, what = [], []
        if pa:
            cls_cost, reg_cost = _get(pa, "cls_cost", {}).get("weight", 1.0), _get(pa, "reg_cost", {}).get("weight", 1.0)
            for i in range(num_imgs):
                if gt_points[i].shape[0] and point_reg.shape[1]:
                    pieces.append(point_match_cost(point_reg[i].detach(), point_cls[i].detach(), gt_points[i],
                                                   gt_points_labels[i], shapes[i], cls_weight=cls_cost, reg_weight=reg_cost))
                    what.append(("cost", i))
        assigned = []
        for i in range(num_imgs):                                              # :2624-2636
            assigned.append(A.max_iou_assign(proposal_list[i][:, :4], gt_bboxes[i], asg.get("pos_iou_thr", 0.5),
                                             asg.get("neg_iou_thr", 0.5), asg.get("min_pos_iou", 0.5),
                                             asg.get("match_low_quality", False))[0])
            pieces.append(assigned[i])
            what.append(("assigned", i))
        if with_mask:
            for i in range(num_imgs):
                if len(semantic_centers_split[i]):
                    pieces.append(mask_point_labels[i])
                    what.append(("mask_labels", i))
        host = {k: v for k, v in zip(what, read_back(pieces))}
        if pa:                                                                 # :2568-2602
            bh = self.bbox_head
            matches = point_matches(point_cls, point_reg, gt_points, gt_points_labels, shapes, cls_cost, reg_cost,
                                    costs_host=[host.get(("cost", i)) for i in range(num_imgs)])
            losses.update(point_token_loss(
                point_cls, point_reg, gt_points, gt_points_labels, shapes,
                num_classes=bh.num_classes, loss_point_weight=bh.loss_point_cfg.get("loss_weight", 10.0),
                loss_cls_weight=bh.loss_point_cls_cfg.get("loss_weight", 1.0),
                gamma=bh.loss_point_cls_cfg.get("gamma", 2.0), alpha=bh.loss_point_cls_cfg.get("alpha", 0.25),
                point_pos_weight=_get(rcnn, "point_pos_weight", 1), cls_cost=cls_cost, reg_cost=reg_cost, matches=matches,
                ranks=getattr(self, "ranks", None)))        # reduce_mean of the matched-token count (stdroi:3430-3514)
        sampling_results = [A.random_sample(proposal_list[i][:, :4], gt_bboxes[i], gt_labels[i], assigned[i],
                                            smp.get("num", 512), smp.get("pos_fraction", 0.25),
                                            smp.get("add_gt_as_proposals", True), generator,
                                            assigned_host=host[("assigned", i)]) for i in range(num_imgs)]
        # ---- box branch (:2974-3020) ----
        rois = torch.cat([torch.cat((r.bboxes.new_full((r.bboxes.shape[0], 1), float(i)), r.bboxes), dim=1)
                          for i, r in enumerate(sampling_results)])
        bbox_feats = self._roi_extract(x, rois)
        cls_score, bbox_pred, _rec = self.bbox_head(bbox_feats)
        targets = self.bbox_head.get_targets(sampling_results, _get(rcnn, "pos_weight", -1))
        # the rows of the positives: each image's block of `rois` starts with them
        offs = np.cumsum([0] + [r.bboxes.shape[0] for r in sampling_results])
        pos_rows = np.concatenate([offs[i] + np.arange(r.pos_bboxes.shape[0]) for i, r in enumerate(sampling_results)])
        pos_index = to_device(pos_rows.astype(np.int64), rois.device)
        losses.update(self.bbox_head.loss(cls_score, bbox_pred, rois, *targets, pos_index=pos_index,
                                          num_weighted=int(rois.shape[0])))
        # ---- mask branch (:3094-3160): the positives' BOX features through the mask head, BCE at the mask points ----
        if with_mask:
            # the positives' rows gathered in the layout RoIAlign wrote them in ([R, 7, 7, C]; bbox_feats is its NCHW-shaped
            # view): the gather, its backward scatter and the sum of the two branches' gradients stay token-major
            tok = bbox_feats.permute(0, 2, 3, 1)
            mask_in = tok[pos_index].permute(0, 3, 1, 2) if tok.is_contiguous() else bbox_feats[pos_index]
            mask_pred = self.mask_head(mask_in)
            sites, mask_t = mask_point_targets([r.pos_bboxes for r in sampling_results],
                                               [r.pos_assigned_gt_inds for r in sampling_results],
                                               mask_point_coords, mask_point_labels, semantic_centers_split,
                                               labels_host=[host.get(("mask_labels", i)) for i in range(num_imgs)])
            pos_labels = torch.cat([r.pos_gt_labels for r in sampling_results])
            if mask_pred.shape[0]:
                losses.update(self.mask_head.loss(point_sample(mask_pred, sites, align_corners=False), mask_t, pos_labels))
            else:
                losses.update(self.mask_head.loss(mask_pred, mask_t, pos_labels))
        self.last_sampling_results = sampling_results
        return losses

    def train_losses(self, roi_feature_map, img_metas, proposal_list, vit_feat, attns, point_cls, point_reg, gt_points,
                     gt_points_labels, generator=None, **seed_kw):
        """The RoI-head half of the detector's training step with precomputed proposals
        (two_stage_point_align.py:75-150, the `proposal_list = proposals` branch): pseudo labels from the attention
        shift, then the box / mask / point / MIL losses against them.  Returns (losses, seed_pseudo_gt's dict)."""
        seed = self.seed_pseudo_gt(None, img_metas, None, None, None, vit_feat=vit_feat, point_cls=point_cls,
                                   point_reg=point_reg, attns=attns, gt_points=gt_points,
                                   gt_points_labels=gt_points_labels, roi_feature_map=roi_feature_map, return_mask=True,
                                   **seed_kw)
        losses = dict(seed["mil_losses"])
        losses.update(self.forward_train(
            roi_feature_map, img_metas, proposal_list, seed["pseudo_gt_bboxes"], seed["pseudo_gt_labels"], None,
            seed["pseudo_gt_masks"], vit_feat=vit_feat, point_cls=point_cls, point_reg=point_reg, attns=attns,
            gt_points=gt_points, gt_points_labels=gt_points_labels, mask_point_coords=seed["mask_points_coords"],
            mask_point_labels=seed["mask_points_labels"], semantic_centers=seed["semantic_centers"],
            semantic_centers_split=seed["semantic_centers_split"], sc_corres_gts=seed.get("corres_gts"),
            generator=generator))
        return losses, seed

    def simple_test(self, x, proposal_list, img_metas, proposals=None, rescale=False):
        """stdroi:3192-3221 + test_mixins.py:52-170, 262-340: per image, the box head on the proposals' RoI features ->
        softmax scores + decoded boxes -> class-aware NMS; then the mask head on the DETECTIONS' RoI features (the mask
        extractor's 14x14 grid) -> pasted binary masks.  Returns [(bbox_results, segm_results)] per image, or
        [bbox_results] without a mask head."""
        from . import inference as I
        if not isinstance(self.bbox_head, nn.Module):
            raise RuntimeError("simple_test needs the box head built from its config (bbox_head with in_channels)")
        cfg = self.test_cfg
        nms_cfg = dict(_get(cfg, "nms", None) or {})
        bh = self.bbox_head
        out = []
        for i, meta in enumerate(img_metas):
            props = proposal_list[i][:, :4]
            rois = torch.cat((props.new_full((props.shape[0], 1), float(i)), props), dim=1)
            if rois.shape[0]:
                cls_score, bbox_pred, _ = bh(self._roi_extract(x, rois))
            else:
                cls_score, bbox_pred = props.new_zeros(0, bh.num_classes + 1), props.new_zeros(0, 4 * bh.num_classes)
            dets, labels = I.get_det_bboxes(rois, cls_score, bbox_pred, meta["img_shape"], meta.get("scale_factor", 1.0),
                                            rescale, _get(cfg, "score_thr", 0.05), nms_cfg.get("iou_threshold", 0.5),
                                            _get(cfg, "max_per_img", 100), bh.target_means, bh.target_stds)
            boxes_res = I.bbox2result(dets, labels, bh.num_classes)
            if self.mask_head is None:
                out.append(boxes_res)
                continue
            mboxes = dets[:, :4] * dets.new_tensor(meta.get("scale_factor", 1.0)) if rescale else dets[:, :4]
            mrois = torch.cat((mboxes.new_full((mboxes.shape[0], 1), float(i)), mboxes), dim=1)
            if mrois.shape[0]:
                mask_pred = self.mask_head(self._roi_extract(x, mrois, "mask_roi_extractor"))
            else:
                mask_pred = mboxes.new_zeros(0, self.mask_head.num_classes, 1, 1)
            # the INPUT-scale boxes go in (test_mixins.py:293-331): get_seg_masks divides by scale_factor once itself
            segm = I.get_seg_masks(mask_pred, mboxes, labels, self.mask_head.num_classes, meta["ori_shape"],
                                   meta.get("scale_factor", 1.0), rescale, _get(cfg, "mask_thr_binary", 0.5),
                                   self.mask_head.class_agnostic)
            out.append((boxes_res, segm))
        return out

    # ---- stage helpers ----------------------------------------------------------------------------
    def rollout_cams(self, attns, num_proposals, pos_inds=None):
        """A3 (stdroi:2261): [B, Lc, T, N] rows of the roll-out for the T point tokens.  With `pos_inds` (per-image index
        lists of the matched point tokens) only THOSE rows are rolled out below the top layer -- the only ones stdroi:2272
        + the pos_inds gather ever read; a row of the product depends on that row alone, so the values are unchanged --
        and the result is [B, Lc, Gmax, N] with image i's rows in pos_inds[i] order (padded by repeating its last index)."""
        Lc = self.bbox_head.cam_layer
        states = attns[-Lc:]
        if not isinstance(states[0], ops.AttnLayerState):
            raise TypeError("attns must be the AttnLayerState handles returned by the MI355X VisionTransformerDet "
                            "(dense [B,N,N] attention maps are never materialised on this path)")
        self._rows_matched_only = False
        if pos_inds is None or min(int(p.numel()) for p in pos_inds) == 0:
            return ops.rollout_rows(states, num_proposals)
        self._rows_matched_only = True                       # tells seed_pseudo_gt how to index the result
        gmax = max(int(p.numel()) for p in pos_inds)
        sel = torch.stack([torch.cat((p, p[-1:].expand(gmax - p.numel()))) if p.numel() < gmax else p for p in pos_inds])
        return ops.rollout_rows(states, num_proposals, rows=sel.to(states[0].q.device).long())

    def refine_maps(self, attn_sel, feat_chw, rois, gt_points, refine_times, obj_tau, minmax=None, cam_src=None,
                    draw_gen=None, flags_out=None, last_level_only=False, mt_state=None, flag_slot=None, box_patch=None,
                    draw_u=None):
        """B2 (stdroi:1000-1019).  attn_sel [G,H,W], feat [C,hp,wp]; minmax [G,2] = per-map (min,max) if the
        caller already has them (as_cam_boxes does).  cam_src = (cams_lr [M,hp,wp], map_idx [G] int32, minmax [M,2])
        replaces attn_sel: the seed sampling then reads the low-resolution CAMs and the upsampled maps are never
        materialised.  Returns map_fg, map_bg [R+1,G,H,W], points_fg, points_bg, fg_feat, bg_feat."""
        C, hp, wp = feat_chw.shape
        seed_idx = None
        if cam_src is not None and mt_state is not None:          # reference stream, drawn on the device: no readback
            G = cam_src[1].shape[0]
            pts_bg, pts_fg, pts_supp, short = sample_points_from_cams_mt(cam_src[0], cam_src[1], cam_src[2], 20, mt_state)
            flags_out.append(short)
        elif cam_src is not None and draw_gen is not None:        # fast-RNG mode: no readback, flag instead
            G = cam_src[1].shape[0]
            pts_bg, pts_fg, pts_supp, short = sample_points_from_cams_nosync(cam_src[0], cam_src[1], cam_src[2], 20, draw_gen,
                                                                             flag=flag_slot, u=draw_u)
            flags_out.append(short)
            if flag_slot is not None and pts_supp.dim() == 2:      # (pts_fg already holds the shared group; pts_supp = token ids)
                seed_idx, pts_supp = pts_supp, None
        elif cam_src is not None:
            G = cam_src[1].shape[0]
            pts_bg, pts_fg, pts_supp = sample_points_from_cams(cam_src[0], cam_src[1], cam_src[2], gt_points, 20)
        else:
            G = attn_sel.shape[0]
            if minmax is None:
                nm = _minmax_maps(attn_sel)
            else:                                               # norm_attns (:329-333) with known extrema
                lo, hi = minmax[:, 0][:, None, None], minmax[:, 1][:, None, None]
                nm = (attn_sel - lo) / (hi - lo)
            pts_bg, pts_fg, pts_supp = sample_point_grid_multi(
                [(nm, 0.1, False, None), (nm, 0.2, True, gt_points), (nm.mean(0, keepdim=True), 0.1, False, None)], 20)
        if pts_supp is not None:
            pts_fg = torch.cat((pts_fg, pts_supp), dim=0)
        if self.capture is not None:
            self.capture.append(dict(points_fg=pts_fg, points_bg=pts_bg))
        CLOCK.mark("  sampling")
        feat_tok = feat_chw.flatten(1).t().contiguous()
        if box_patch is None:
            box_patch = (rois // STRIDE).to(torch.int32).contiguous()
        # foreground (G + 1 seeds, selection group) and background (G seeds) sets refined by ONE call
        if seed_idx is not None:                                  # token ids straight from the selection kernel
            seeds = feat_tok[seed_idx].mean(dim=1)
        else:
            seeds = seed_features(torch.cat((pts_fg, pts_bg), dim=0), feat_chw).contiguous()
        n_fg = pts_fg.shape[0]
        sims, seeds_out = ops.refine_similarity(feat_tok, seeds, box_patch, G, refine_times, obj_tau, n_fg, hp, wp)
        sim_fg, sim_bg = sims[:, :n_fg], sims[:, n_fg:]
        fg_feat, bg_feat = seeds_out[:n_fg], seeds_out[n_fg:]
        CLOCK.mark("  refine_similarity")
        if last_level_only:                                     # the chain below only consumes the last refinement level
            sim_fg, sim_bg = sim_fg[-1:], sim_bg[-1:]
        map_fg, map_bg = ops.instance_maps(sim_fg.contiguous(), sim_bg.contiguous(), G, hp, wp, STRIDE)
        CLOCK.mark("  instance_maps")
        return map_fg, map_bg, pts_fg, pts_bg, fg_feat[:, :, None, None], bg_feat[:, :, None, None]

    def get_mask_sample_points_roi_best_attn_feat_refine(self, attn, rois, attn_idx, vit_feat, pos_thr=0.6, neg_thr=0.6,
                                                         num_gt=20, corr_size=21, refine_times=2, obj_tau=0.85,
                                                         gt_points=None, minmax=None):
        """stdroi:1966-1993 (same argument meaning and return order)."""
        G = attn.shape[1]
        ar = torch.arange(G, device=attn.device)
        attn_sel = attn[attn_idx, ar].contiguous()
        mm = None if minmax is None else minmax[attn_idx, ar]
        map_fg, map_bg, pts_a, pts_b, f_fg, f_bg = self.refine_maps(attn_sel, vit_feat, rois, gt_points, refine_times,
                                                                    obj_tau, mm)
        coords, labels = mask_sample_points(map_fg[-1], map_bg[-1], rois, pos_thr, neg_thr, num_gt, corr_size,
                                            self.rng_mode)
        return coords, labels, map_fg, map_bg, pts_a, pts_b, f_fg, f_bg

    def mean_shift_grid_prototype(self, maps, vit_feat, rois=None, thr=0.35, n_shift=5, output_size=(4, 4), tau=0.1,
                                  temp=0.1, n_points=20):
        """stdroi:1778-1840 (rois given).  Returns (prototypes [G*P,C], sim [G*P,hp,wp] clamped at 0)."""
        if rois is None:
            raise NotImplementedError("the rois=None branch is never taken by seed_pseudo_gt")
        C, hp, wp = vit_feat.shape
        coords = grid_seed_coords(maps, rois, thr, n_points)
        feat_tok = vit_feat.flatten(1).t().contiguous()
        prot = vit_feat.permute(1, 2, 0)[coords[..., 0], coords[..., 1]].contiguous()
        box_patch = (rois // STRIDE).to(torch.int32).contiguous()
        obj_img = torch.zeros(maps.shape[0], dtype=torch.int32, device=vit_feat.device)
        pout, sim = ops.cosine_shift(feat_tok[None], box_patch, obj_img, prot, n_shift, hp, wp, tau, temp)
        return pout.flatten(0, 1), sim.reshape(-1, hp, wp).clamp(0)

    def mean_shift_batch(self, coords_list, feats_list, rois_list, n_shift, tau=0.1, temp=0.1, feat_tok=None,
                         box_patch_list=None, seed_ids=None, clamp=True, box_patch_all=None):
        """mean_shift_grid_prototype for every image of the batch in ONE as_cosine_shift call (coords_list[i] =
        grid_seed_coords of image i; the objects carry their
        image index; the kernels are batched over objects, and a call's latency does not depend on how many objects it
        holds).  Returns per-image (prototypes [G_i*P,C], sim [G_i*P,hp,wp] clamped at 0) exactly as the per-image
        method does (batched == per-image bitwise, tests/test_gpu_kernels.py)."""
        C, hp, wp = feats_list[0].shape
        sizes = tuple(int(c.shape[0]) for c in coords_list)
        dev = feats_list[0].device
        if feat_tok is None or not _token_blocks_ok(feat_tok):
            feat_tok = torch.stack([f.flatten(1).t() for f in feats_list]).contiguous()
            seed_ids = None                                  # (the ids index the caller's token tensor)
        if seed_ids is not None:                             # token ids of the seeds (as_rank_draw_xy): one gather
            prot = _token_rows(feat_tok)[seed_ids]
        else:
            prot = torch.cat([feat.permute(1, 2, 0)[coords[..., 0], coords[..., 1]]
                              for coords, feat in zip(coords_list, feats_list)]).contiguous()
        if box_patch_all is not None:                        # the batch's patch boxes in one tensor already (image order)
            boxes = box_patch_all
        else:
            if box_patch_list is None:
                box_patch_list = [(rois // STRIDE).to(torch.int32) for rois in rois_list]
            boxes = torch.cat(box_patch_list).contiguous() if len(box_patch_list) > 1 else box_patch_list[0].contiguous()
        owners = _const_tensor(("owners", sizes), dev,
                               lambda: torch.cat([torch.full((n,), i, dtype=torch.int32) for i, n in enumerate(sizes)]),
                               lambda: torch.cat([torch.full((n,), i, dtype=torch.int32, device=dev) for i, n in enumerate(sizes)]))
        pout, sim = ops.cosine_shift(feat_tok, boxes, owners, prot, n_shift, hp, wp, tau, temp)
        out, off = [], 0
        for g in sizes:
            s_i = sim[off:off + g].reshape(-1, hp, wp)
            out.append((pout[off:off + g].flatten(0, 1), s_i.clamp(0) if clamp else s_i))
            off += g
        return out

    def get_semantic_centers(self, map_cos_fg, map_cos_bg, rois, vit_feat, pos_thr=0.35, refine_times=5, gt_labels=None,
                             merge_thr=0.85, num_semantic_points=3):
        """stdroi:1995-2031 (same nine outputs)."""
        hp, wp = vit_feat.shape[-2:]
        G, H, W = map_cos_fg.shape
        # :2011-2013.  erode_11(map > thr) at full resolution, then the bilinear /16 down-sampling, which for an
        # exact factor of 16 reads only the 2x2 centre pixels of each patch with weights 1/2 (bit-identical)
        fg_inter, map_fg, _seeds = self._semantic_pre(map_cos_fg, map_cos_bg, pos_thr)
        prot, sim = self.mean_shift_grid_prototype(map_fg, vit_feat, rois, tau=0.1, temp=0.1, n_shift=refine_times)
        CLOCK.mark("  sc:mean_shift")
        return self._semantic_post(prot, sim, fg_inter, rois, vit_feat, gt_labels, merge_thr, num_semantic_points)

    def _device_gen(self, device, reseed=False):
        """Device generator of the fast RNG mode.  `reseed` (once per seed_pseudo_gt call) draws its seed from torch's
        global CPU generator, so that equal global seeds give equal samples, as in the reference mode."""
        key = str(device)
        if key not in self._dev_gens:
            self._dev_gens[key] = torch.Generator(device=device)
            reseed = True
        if reseed:
            self._dev_gens[key].manual_seed(int(torch.randint(2 ** 62, (1,)).item()))
        return self._dev_gens[key]

    def _semantic_pre(self, map_cos_fg, map_cos_bg, pos_thr, float_map=True, want_counts=True):
        """First part of get_semantic_centers (stdroi:2011-2020): the patch-grid foreground maps, one fused launch
        (ops.semantic_prestage).  The reference also down-samples max_g(map_cos_bg) here (bg_inter, :2013), but its
        only consumer is commented out (filter_maps :267), so it is not computed.  Returns fg_inter, the binary
        patch map (float, as the reference), and (mask uint8, counts) of its positives for the grid seeds."""
        kw = {} if want_counts else dict(want_counts=False)   # (only the sync-free path asks the kernel to skip the counting)
        fg_inter, mask, counts = ops.semantic_prestage(map_cos_fg.contiguous(), pos_thr, 11, STRIDE, **kw)
        CLOCK.mark("  sc:prestage")
        return fg_inter, (mask.to(fg_inter.dtype) if float_map else None), (mask, counts)

    def _semantic_post(self, prot, sim, fg_inter, rois, vit_feat, gt_labels, merge_thr, num_semantic_points, extra=None):
        """Last part of get_semantic_centers (stdroi:2022-2031): filter / merge the shifted prototypes, part centres.
        `extra`: device flags read back with the merge plan (see merge_parts); returns None if any of them is set."""
        G = fg_inter.shape[0]
        P = sim.shape[0] // G
        keep = ops.filter_parts(sim.unflatten(0, (G, P)), fg_inter)                  # one launch (stdroi:263-271)
        merged = merge_parts(prot.unflatten(0, (G, P)), keep, merge_thr, extra)
        if extra and any(extra):
            return None
        CLOCK.mark("  sc:filter+merge")
        sim_parts = part_similarity(merged, vit_feat)
        CLOCK.mark("  sc:part_similarity")
        (centers, split, feat_split, feats, num_parts, coords_org, labels_org, corres) = part_centers(
            sim_parts, rois, gt_labels, vit_feat, num_max_obj=num_semantic_points)
        return centers, split, sim_parts, feat_split, feats, num_parts, coords_org, labels_org, corres

    def _semantic_post_device(self, prot, sim, fg_inter, rois, vit_feat, gt_labels, merge_thr, num_semantic_points,
                              extra=None, num_max_keep=50):
        """_semantic_post_issue + _semantic_post_finish back to back (one image at a time)."""
        return self._semantic_post_finish(self._semantic_post_issue(prot, sim, fg_inter, rois, vit_feat, gt_labels,
                                                                   merge_thr, num_semantic_points, extra), num_max_keep)

    def _semantic_post_issue(self, prot, sim, fg_inter, rois, vit_feat, gt_labels, merge_thr, num_semantic_points,
                             extra=None, flag_slot=None, flag_vec=None):
        """_semantic_post with ONE readback (fast-RNG path), first half: everything up to and including the START of
        that readback (a device -> pinned-host copy + event), so that a caller with several images can queue all of them
        before it waits for the first: the greedy merge plan (ops.merge_plan), the merged prototypes, their similarity
        maps and the per-part statistics are computed for all P group slots of every object (unused slots are zero
        prototypes); the visiting order / cap logic of stdroi:222-262 and the gathers behind it run on the device too
        (ops.part_select), so only the per-object counts and the group counts are read back -- together with the `extra`
        flags.  Returns the state _semantic_post_finish consumes."""
        G = fg_inter.shape[0]
        P = sim.shape[0] // G
        hp, wp = vit_feat.shape[-2:]
        dev = prot.device
        protg = prot.unflatten(0, (G, P))
        keep = ops.filter_parts(sim.unflatten(0, (G, P)), fg_inter)                  # one launch (stdroi:263-271)
        # objects rarely keep more than a few merged parts: carry `slots` group slots per object through the
        # similarity / statistics passes; an object with more raises a flag (synchronous path), like the other rare cases
        slots = min(P, self.part_slots) if extra is not None else P
        # cosine links, greedy grouping and the merged prototypes matmul(weight, prot) / (sum + 1e-8) in one launch
        # (as_merge_parts: stdroi:278-294 for every object; it replaces a chain of seventeen tensor ops)
        over = None
        if extra is not None:                             # (`flag_slot`: a zeroed int32 [1] slot of the caller's flag vector)
            over = flag_slot if flag_slot is not None else torch.zeros(1, dtype=torch.int32, device=dev)
        merged, ng32 = ops.merge_parts(protg, keep, merge_thr, slots, over)
        if extra is not None:
            extra.append(over.reshape(()))
        feat_tok = vit_feat.flatten(1).t().contiguous()
        allp = merged.flatten(0, 1)
        P = slots                                                                      # from here on: slots per object
        sims = [ops.refine_similarity(feat_tok, allp[o:o + 32], None, 0, 0, 1.0, False, hp, wp)[0][0] for o in range(0, G * P, 32)]
        sims = (sims[0] if len(sims) == 1 else torch.cat(sims)).reshape(G, P, hp, wp)
        # per-slot statistics exactly as part_centers computes them per part (one launch), then the visiting order / cap
        # logic of stdroi:222-262 and the gathers behind it in as_part_select: nothing here waits for the host
        slot_owner = _const_tensor(("slot_owner", G, P), dev,
                                   lambda: torch.arange(G, dtype=torch.int32).repeat_interleave(P))
        c, yx, area, inside = ops.part_stats_raw(sims.flatten(0, 1), rois, slot_owner, STRIDE)
        coords, coords_org, labels, labels_org, corres, feats, split = ops.part_select(
            area, inside, ng32, c, yx, gt_labels.long().contiguous(), feat_tok, G, P, wp, num_semantic_points)
        pieces = [split, ng32]
        lo = 0 if flag_vec is None else flag_vec.data_ptr()
        if extra and flag_vec is not None and all(e.dtype == torch.int32 and lo <= e.data_ptr() < lo + 4 * flag_vec.numel()
                                                  for e in extra[1:]):
            # every flag but the first (the caller's CAM check) is a slot of the caller's zero-filled vector: no stack
            pieces += [extra[0].reshape(1), flag_vec]
            extra[:] = [extra[0]] + [None] * flag_vec.numel()
        elif extra:
            pieces.append(torch.stack([e.reshape(()) for e in extra]).int())
        return dict(pending=_to_host_issue(torch.cat(pieces)), extra=extra, G=G, P=P, sims=sims, rois=rois, dev=dev,
                    gt_labels=gt_labels, picked=(coords, coords_org, labels, labels_org, corres, feats))

    def _semantic_post_finish(self, st, num_max_keep=50):
        """Second half: wait for the readback (the one sync of the image) and slice the padded device results by the
        per-object counts.  Returns None when a flag asks for the synchronous path."""
        host = _to_host_finish(st["pending"])
        extra, G, P, sims, rois, dev, gt_labels = (st[k] for k in ("extra", "G", "P", "sims", "rois", "dev", "gt_labels"))
        if extra:
            extra[:] = [bool(v) for v in host[2 * G + 1:]]
            if any(extra):
                return None
        split, total, ng = [int(v) for v in host[:G]], int(host[G]), host[G + 1:2 * G + 1]
        sim_parts = [sims[g, :int(ng[g])] if ng[g] else torch.zeros(0, 0) for g in range(G)]
        dt_c, dt_l = rois.dtype, gt_labels.dtype
        if total == 0:
            pc = ([torch.zeros(0, 2, dtype=dt_c, device=dev), torch.zeros(0, dtype=dt_l, device=dev)], [], [], [], [0] * G,
                  torch.zeros(0, 2, dtype=dt_c, device=dev), torch.zeros(0, dtype=dt_l, device=dev),
                  torch.zeros(0, dtype=torch.long, device=dev))
        else:
            coords, coords_org, labels, labels_org, corres, feats = (t[:total] for t in st["picked"])
            if dt_c != coords.dtype:
                coords, coords_org = coords.to(dt_c), coords_org.to(dt_c)
            if dt_l != labels.dtype:
                labels, labels_org = labels.to(dt_l), labels_org.to(dt_l)
            coord_split, feats_split = list(coords.split(split, dim=0)), list(feats.split(split, dim=0))
            if total > num_max_keep:
                pick = torch.randperm(total, device=coords.device)[:num_max_keep]
                coords, labels = coords[pick], labels[pick]
            pc = ([coords, labels], coord_split, feats_split, feats, split, coords_org, labels_org, corres)
        (centers, csplit, feat_split, feats, num_parts, coords_org, labels_org, corres) = pc
        return centers, csplit, sim_parts, feat_split, feats, num_parts, coords_org, labels_org, corres

    # ---- the hot-path entry point -------------------------------------------------------------------
    @torch.no_grad()
    def _run_images(self, fn, num_imgs):
        """Per-image chains B2..B6 of a batch.  The images are independent (the reference loops over them, stdroi:2318)
        and each chain is a sequence of small launches separated by host decisions (candidate counts, part merging), so
        with `parallel_images` the chains run on one host thread + one HIP stream per image: one image's host stalls
        are filled with the other's device work.  Sequential whenever the literal reference RNG stream is requested
        (its draws are ordered across images) or the stage clock is on."""
        parallel = (self.parallel_images and num_imgs > 1 and self.rng_mode == "fast" and not CLOCK.on
                    and torch.cuda.is_available())
        if not parallel:
            return [fn(i) for i in range(num_imgs)]
        if getattr(self, "_pool", None) is None or len(self._streams) < num_imgs:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=num_imgs, thread_name_prefix="attnshift-img")
            self._streams = [torch.cuda.Stream() for _ in range(num_imgs)]
        main = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        seeds = torch.randint(2 ** 31 - 1, (num_imgs,)).tolist()            # image order, from the global generator
        grad = torch.is_grad_enabled()

        def job(i):
            torch.cuda.set_device(dev)
            _TLS.gen = torch.Generator().manual_seed(seeds[i])
            try:
                with torch.set_grad_enabled(grad), torch.cuda.stream(self._streams[i]):
                    self._streams[i].wait_stream(main)                      # inputs were produced on the caller's stream
                    return fn(i)
            finally:
                _TLS.gen = None

        # the chains are short (a few ms) and alternate between Python and blocking device waits: with the interpreter's
        # default 5 ms switch interval one thread can hold the GIL for a whole chain while the other's device queue runs dry
        import sys
        old_iv = sys.getswitchinterval()
        sys.setswitchinterval(1e-4)
        try:
            futs = [self._pool.submit(job, i) for i in range(num_imgs)]
            res = [f.result() for f in futs]
        finally:
            sys.setswitchinterval(old_iv)
        for st in self._streams[:num_imgs]:
            main.wait_stream(st)                                            # results are consumed on the caller's stream
        return res

    def seed_pseudo_gt(self, *args, **kw):
        """stdroi:2209-2415 (signature: _seed_pseudo_gt).  In the reference-RNG mode the draws are first attempted on the
        device from torch's own engine state (no host round trip, same stream); if an image takes one of the reference's
        rare refill branches, the call is repeated on the host path -- the global generator has not been touched by the
        first attempt, so the repetition draws exactly what the reference draws."""
        ncap = len(self.capture) if self.capture is not None else 0
        try:
            return self._seed_pseudo_gt(*args, **kw)
        except _HostDrawsNeeded:
            # the abandoned attempt may still have work queued on the per-image streams (the exception left the closing
            # join out): order the redo -- and the release of the attempt's tensors -- behind it
            if torch.cuda.is_available():
                cur = torch.cuda.current_stream()
                for st in self._streams:
                    cur.wait_stream(st)
            self.rng_stats["host_redos"] += 1
            if self.capture is not None:
                del self.capture[ncap:]                     # what the abandoned attempt recorded
            return self._seed_pseudo_gt(*args, _mt_ok=False, **kw)

    def _seed_pseudo_gt(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                        vit_feat=None, img=None, point_init=None, point_cls=None, point_reg=None, imgs_whwh=None,
                        attns=None, gt_points=None, gt_points_labels=None, roi_feature_map=None, return_mask=False,
                        pos_mask_thr=0.6, neg_mask_thr=0.1, num_mask_point_gt=10, corr_size=21, point_adjuster=None,
                        edges=None, obj_tau=0.85, pos_inds=None, matched_gt=None, point_ready=None, _mt_ok=True):
        """stdroi:2209-2415.  Extra optional inputs `pos_inds` / `matched_gt` (per-image lists) bypass the
        Hungarian matching when the caller already has it (fixtures, benchmarks).  `point_ready`: the event behind which
        point_cls / point_reg are valid when the backbone computed them on its side stream (point_head_stream); only their
        shapes are read before it."""
        num_imgs = point_reg.size(0)
        num_proposals = point_cls.size(1)
        if pos_inds is None:
            if point_ready is not None:
                torch.cuda.current_stream().wait_event(point_ready)
            pa = getattr(self.train_cfg, "point_assigner", None) or {}
            pos_inds, matched_gt = [], []
            for i in range(num_imgs):
                pi, mg = hungarian_point_match(point_reg[i].detach(), point_cls[i], gt_points[i], gt_points_labels[i],
                                               img_metas[i]["img_shape"],
                                               cls_weight=_get(pa, "cls_cost", {}).get("weight", 1.0),
                                               reg_weight=_get(pa, "reg_cost", {}).get("weight", 1.0))
                pos_inds.append(pi)
                matched_gt.append(mg)
        gt_labels = [gt_points_labels[i][matched_gt[i]] for i in range(num_imgs)]
        point_targets = [gt_points[i][matched_gt[i]] for i in range(num_imgs)]

        patch_h, patch_w = vit_feat.shape[-2:]
        H, W = patch_h * STRIDE, patch_w * STRIDE
        Lc = self.bbox_head.cam_layer
        CLOCK.start()
        counts = [int(p.numel()) for p in pos_inds]
        self._rows_matched_only = False
        rows = (self.rollout_cams(attns, num_proposals, pos_inds) if self.rollout_matched_only
                else self.rollout_cams(attns, num_proposals))
        subset = self._rows_matched_only                                     # (an overridden rollout_cams may return all T rows)
        CLOCK.mark("rollout")                                                # [B, Lc, T | Gmax, N]
        # B1, batched over every (image, layer, object): one launch sequence for the whole batch
        cams_lr = torch.cat([(rows[i][:, :counts[i], 1:-num_proposals] if subset else
                              rows[i][:, pos_inds[i], 1:-num_proposals]).reshape(-1, patch_h, patch_w)
                             for i in range(num_imgs)]).contiguous()
        pts = torch.cat([point_targets[i].float().repeat(Lc, 1) for i in range(num_imgs)]).contiguous()
        if cams_lr.shape[0] == 0:
            raise RuntimeError("seed_pseudo_gt: no matched point tokens in the batch")
        if self.visualize:                                   # the reference keeps attn_maps_dealed only for plots
            boxes, status, cams_up, cam_minmax = ops.cam_boxes(cams_lr, pts, self.bbox_head.seed_thr,
                                                               self.bbox_head.seed_multiple, STRIDE, True)
        else:
            cams_up = None
            boxes, status, cam_minmax = ops.cam_boxes(cams_lr, pts, self.bbox_head.seed_thr,
                                                      self.bbox_head.seed_multiple, STRIDE, return_minmax=True)
        CLOCK.mark("cam_boxes")
        if hasattr(x, "launch"):
            # backbone.DeferredFPN: from here to the RoI feature extraction this path is a dependent chain of small launches
            # that leaves the device idle -- the FPN's GEMMs run under it on their own stream
            x.launch()
        # the reference raises here when a CAM has no foreground component (torch.stack of an empty list, stdroi:80).
        # The flag stays on the device and is checked at the first host sync the chain needs anyway (the seed counts):
        # reading it here would stall the host for the whole roll-out + CAM-box phase with nothing queued behind it.
        # the default selector (median box area over the roll-out depths) with its index arithmetic as ONE launch for the batch:
        # the chosen layer, its box, that box's row of the CAM stack, its patch box -- and the CAM check above
        # (csrc/refine.hip select_median_boxes)
        # Any other selector (the trained MIL head) still makes its choice itself -- from the per-image box lists -- and only
        # the indexing behind it runs in the kernel (`pick_in`).
        fused_idx = boxes.is_cuda and sum(counts) > 0 and os.environ.get("AS_HEAD_TENSOR_GLUE") != "1"
        fused_sel = fused_idx and self.layer_selector is median_area_selector
        flag_all = None
        if fused_idx:
            # one zero fill for every device flag of the call: 4 slots per image (raised by the selection kernels of the
            # sync-free path) + the CAM check
            flag_all = torch.zeros(4 * num_imgs + 1, dtype=torch.int32, device=boxes.device)
            bad_cam = flag_all[4 * num_imgs]
        else:
            bad_cam = (status <= 0).any()                    # 0: no component; -1: run table overflow
            if status.is_cuda:
                bad_cam = bad_cam.to(torch.int32)            # the dtype of the other device flags it is read back with
        gt_scale_bboxes, attn_maps_dealed, cam_off, off = [], [], [], 0
        for i in range(num_imgs):
            n = Lc * counts[i]
            if not fused_sel:
                gt_scale_bboxes.append(boxes[off:off + n].reshape(Lc, counts[i], 4).permute(1, 0, 2).contiguous())
            if cams_up is not None:
                attn_maps_dealed.append(cams_up[off:off + n].reshape(Lc, counts[i], H, W))
            cam_off.append(off)
            off += n

        CLOCK.mark("box_split")
        sel_rows = sel_patch = sel_int = sel_patch_all = None
        if fused_idx:
            def meta_on_device():                                               # rows [cam_off[i], counts[i], g]
                rows = []
                for i in range(num_imgs):
                    m = torch.empty(counts[i], 3, dtype=torch.int32, device=boxes.device)
                    m[:, 0] = cam_off[i]
                    m[:, 1] = counts[i]
                    m[:, 2] = torch.arange(counts[i], dtype=torch.int32, device=boxes.device)
                    rows.append(m)
                return torch.cat(rows) if len(rows) > 1 else rows[0]
            meta = _const_tensor(("select_meta", tuple(counts), Lc), boxes.device, lambda: torch.tensor(
                [[cam_off[i], counts[i], g] for i in range(num_imgs) for g in range(counts[i])], dtype=torch.int32),
                meta_on_device)
            pick_in = None
            if not fused_sel:
                gt_box_index = self.layer_selector(gt_scale_bboxes, gt_labels, roi_feature_map)
                pick_in = (torch.cat([g.long() for g in gt_box_index]) if num_imgs > 1 else gt_box_index[0].long())
                pick_in = pick_in.to(boxes.device).contiguous()
            pick, chosen, rows, patch, ints = ops.select_median_boxes(boxes, meta, Lc, STRIDE, status=status,
                                                                      bad=flag_all[4 * num_imgs:], pick_in=pick_in)
            if fused_sel:
                gt_box_index = list(pick.split(counts))
            pseudo_boxes = list(chosen.split(counts))
            sel_rows, sel_patch, sel_int = rows.split(counts), patch.split(counts), ints.split(counts)
            sel_patch_all = patch
        else:
            gt_box_index = self.layer_selector(gt_scale_bboxes, gt_labels, roi_feature_map)
            pseudo_boxes = [gt_scale_bboxes[i][_const_tensor(("arange", counts[i]), boxes.device,
                                                             lambda n=counts[i]: torch.arange(n)), gt_box_index[i]]
                            for i in range(num_imgs)]
        mil_losses = {}
        if self.mil_head is not None and roi_feature_map is not None and self._mil_selector.last_loss is not None:
            mil_losses["mil_loss"] = self._mil_selector.last_loss                # stdroi:2961

        out = dict(pseudo_gt_labels=gt_labels, pseudo_gt_bboxes=pseudo_boxes, mil_losses=mil_losses,
                   best_attn_idx=gt_box_index, map_cos_fg=[], mask_points_coords=[], mask_points_labels=[],
                   semantic_centers=[], semantic_centers_split=[], semantic_centers_feat_split=[],
                   semantic_centers_feat=[], num_parts=[], pseudo_gt_masks=[], corres_gts=[], inst_fg_feat=[],
                   inst_bg_feat=[])
        coords_sc_org, labels_sc_org, map_cos_bg_ret, sim_fg_ret = [], [], [], []
        CLOCK.mark("select")
        feat_tok = feature_tokens(vit_feat)                                   # [B, Np, C], per-image contiguous
        feats = [feat_tok[i].t().unflatten(1, (patch_h, patch_w)) for i in range(num_imgs)]   # [C,hp,wp] views

        def layer_rows(i):
            if sel_rows is not None:
                return sel_rows[i]
            ar = _const_tensor(("arange", counts[i]), boxes.device, lambda: torch.arange(counts[i]))
            return (cam_off[i] + gt_box_index[i] * counts[i] + ar).to(torch.int32)

        def phase_a(i):
            """Refinement (B2), then EVERYTHING that depends only on the refined maps is queued on the device -- the
            candidate masks of the mask points (B2'), the patch-grid foreground maps and seed counts (B3), the pseudo
            mask and its device->host copy (B6) -- before the first host sync of the chain, so that the device keeps
            working while the host waits for counts and draws.  (stdroi:1966-1993, 2011-2020, 2356-2358.)"""
            map_idx = layer_rows(i)                                                         # rows of cams_lr (layer-major)
            map_fg, map_bg, _pa, _pb, feats_fg, feats_bg = self.refine_maps(
                None, feats[i], pseudo_boxes[i], gt_points[i], 2, obj_tau, cam_src=(cams_lr, map_idx, cam_minmax),
                last_level_only=True)
            mp, mask_u8 = mask_points_and_pseudo_issue(map_fg[-1], map_bg[-1], pseudo_boxes[i], pos_mask_thr, neg_mask_thr,
                                                       corr_size, pos_mask_thr)                # B2' + B6 (stdroi:2356)
            fg_inter, _map_fg_patch, gs = self._semantic_pre(map_fg[-1], map_bg[-1], pos_mask_thr)   # gs: >= 0.35 of a 0/1 map
            pm = _to_host_issue(mask_u8, side_stream=True)
            return mp, gs, map_fg, map_bg, feats_fg, feats_bg, fg_inter, pm

        def phase_a_finish(i, r):
            mp, gs, map_fg, map_bg, feats_fg, feats_bg, fg_inter, pm = r
            coord_point, labels_point = mask_points_finish(mp, num_mask_point_gt, self.rng_mode)
            CLOCK.mark("refine+mask_points")
            seeds = grid_seed_finish(gs[0], gs[1], pseudo_boxes[i], 20)
            if self.capture is not None:
                self.capture.append(dict(image=i, seeds=seeds, fg_inter=fg_inter))
            return coord_point, labels_point, map_fg, map_bg, feats_fg, feats_bg, (fg_inter, seeds), pm

        def phase_a_nosync(i, mt_state=None, flag_slots=None, draws=None):
            """phase_a + phase_a_finish with every draw made on the device: nothing is read back.  Fast RNG mode: uniform
            numbers from the device generator; reference mode (`mt_state`): torch's own engine advanced on the device, the
            images strictly in order on one stream.  The last element is the list of device flags that ask for the
            synchronous path (rare refill branches)."""
            flags = []
            map_idx = layer_rows(i)
            # stdroi:1812 patch box, shared by B2 and B4
            bp = sel_patch[i] if sel_patch is not None else (pseudo_boxes[i] // STRIDE).to(torch.int32)
            map_fg, map_bg, _pa, _pb, feats_fg, feats_bg = self.refine_maps(
                None, feats[i], pseudo_boxes[i], gt_points[i], 2, obj_tau, cam_src=(cams_lr, map_idx, cam_minmax),
                draw_gen=None if mt_state is not None else self._device_gen(boxes.device), flags_out=flags,
                last_level_only=True, mt_state=mt_state, flag_slot=None if flag_slots is None else flag_slots[i, 0:1],
                box_patch=bp, draw_u=None if draws is None else draws[0])
            # what the batched mean shift (caller's stream) waits for comes FIRST -- the patch-grid foreground and the grid seeds
            # -- and is marked with an event; the mask candidates, mask points and the pseudo-mask copy (B2', B6: ~130 us of
            # launches nothing in the semantic chain reads) are queued behind it and run under the mean shift
            # (with a flag slot and a 16-divisible patch grid the selection kernel counts the seeds' candidates itself)
            fg_inter, _map_fg_patch, gs = self._semantic_pre(
                map_fg[-1], map_bg[-1], pos_mask_thr, float_map=False,
                want_counts=flag_slots is None or (patch_h * patch_w) % 16 != 0)
            seeds, f2 = grid_seed_nosync(gs[0], gs[1], 20, flag=None if flag_slots is None else flag_slots[i, 1:2],
                                         patch=None if seed_ids is None else (seed_ids[obj_off[i]:obj_off[i] + counts[i]],
                                                                              i * _token_row_stride(feat_tok)))
            seeds_ready = torch.cuda.current_stream().record_event() if boxes.is_cuda else None
            mp, mask_u8 = mask_points_and_pseudo_issue(map_fg[-1], map_bg[-1], pseudo_boxes[i], pos_mask_thr, neg_mask_thr,
                                                       corr_size, pos_mask_thr, crops=None if sel_int is None else sel_int[i])
            pm = _to_host_issue(mask_u8, side_stream=True)
            if mt_state is not None:
                coord_point, labels_point, f1 = mask_points_mt(mp, num_mask_point_gt, mt_state)
            else:
                coord_point, labels_point, f1 = mask_points_nosync(mp, num_mask_point_gt, self._device_gen(boxes.device),
                                                                   int_flag=flag_slots is not None,
                                                                   flag=None if flag_slots is None else flag_slots[i, 3:4],
                                                                   u=None if draws is None else draws[1])
            if self.capture is not None:
                self.capture.append(dict(image=i, seeds=seeds, fg_inter=fg_inter))
            flags += [f1, f2]
            return coord_point, labels_point, map_fg, map_bg, feats_fg, feats_bg, (fg_inter, seeds), pm, flags, bp, seeds_ready

        nosync = (self.rng_mode == "fast" and self.device_draws and not self.parallel_images and self.image_streams
                  and self.batch_mean_shift and torch.cuda.is_available() and not CLOCK.on and not self.visualize)
        multi = (self.rng_mode == "fast" and num_imgs > 1 and not self.parallel_images and self.image_streams
                 and torch.cuda.is_available() and not CLOCK.on)
        # reference RNG mode with the draws on the device (csrc/mt19937.hip): the same queue-everything-first structure
        mtdev = (_mt_ok and self.rng_mode == "reference" and self.device_draws and _gen() is None and self.image_streams
                 and self.batch_mean_shift and boxes.is_cuda and not CLOCK.on and not self.visualize
                 and os.environ.get("AS_REF_RNG_HOST") is None)
        mt_state = mt_blob = mt_final = flag_slots = seed_ids = None
        obj_off = [sum(counts[:i]) for i in range(num_imgs)]
        if mtdev:
            from . import mt19937 as _MT
            mt_blob = torch.get_rng_state()
            mt_state = to_device(_MT.unpack_state(mt_blob).copy(), boxes.device, torch.int32)
        if nosync or mtdev:
            # Fast RNG mode, device-side draws: the host queues the WHOLE chain of every image (refinement, candidate
            # masks, draws, seeds), the batched mean shift and the merge inputs without reading anything back, i.e. while
            # the device is still in the backbone; the first readback is the merge plan of stdroi:278-294, which also
            # carries the flags of the rare branches that need the synchronous path (and the deferred CAM check).
            if len(self._streams) < num_imgs:
                self._streams = [torch.cuda.Stream() for _ in range(num_imgs)]
            main = torch.cuda.current_stream()

            def on_stream(i, fn, *args):
                with torch.cuda.stream(self._streams[i]):
                    return fn(i, *args)

            if mtdev:
                # the engine state threads through the images IN ORDER (image i + 1's first draw follows image i's last):
                # one stream, no host decision in between
                ra = [phase_a_nosync(i, mt_state) for i in range(num_imgs)]
                mt_final = _to_host_issue(mt_state)
            else:
                gen = self._device_gen(boxes.device, reseed=True)
                # every uniform number of the call in ONE draw: per image [2G+1, 20] for the seed sampling (stdroi:346-369) and
                # [G, 32] for the mask points (stdroi:447)
                n_seed, n_mask = [(2 * c + 1) * 20 for c in counts], [c * 32 for c in counts]
                u_all = torch.rand(sum(n_seed) + sum(n_mask), device=boxes.device, generator=gen).split(n_seed + n_mask)
                draws = [(u_all[i].view(2 * counts[i] + 1, 20), u_all[num_imgs + i].view(counts[i], 32)) for i in range(num_imgs)]
                # device flags raised by the selection kernels themselves (too few candidates): one zero fill for the batch
                if os.environ.get("AS_HEAD_TENSOR_GLUE") != "1":     # (A/B switch: "1" = the tensor-op forms of the draws)
                    flag_slots = (flag_all[:4 * num_imgs].view(num_imgs, 4) if flag_all is not None else
                                  torch.zeros(num_imgs, 4, dtype=torch.int32, device=boxes.device))
                if flag_slots is not None and _token_blocks_ok(feat_tok) and (patch_h * patch_w) % 16 == 0:
                    # token ids (image offset included) of every object's grid seeds, written by the selection kernels: the
                    # mean shift's initial prototypes are then ONE gather for the batch
                    seed_ids = torch.empty(sum(counts), 20, dtype=torch.int64, device=boxes.device)
                for st in self._streams[:num_imgs]:
                    st.wait_stream(main)
                ra = [on_stream(i, phase_a_nosync, None, flag_slots, draws[i]) for i in range(num_imgs)]
                if self.overlap_mask_work:
                    for r in ra:                                 # the seeds only: the mask-point work behind them keeps running
                        main.wait_event(r[10])
                else:                                            # (measurement switch: the mean shift alone on the device)
                    for st in self._streams[:num_imgs]:
                        main.wait_stream(st)
            shifted = self.mean_shift_batch([r[6][1] for r in ra], feats, pseudo_boxes, self.mean_shift_times_local,
                                            feat_tok=feat_tok, box_patch_list=[r[9] for r in ra], seed_ids=seed_ids,
                                            clamp=False, box_patch_all=sel_patch_all)       # (the only consumer thresholds the maps at 0.8: no clamp(0))

            def chain_issue(i):
                prot, sim = shifted[i]
                extra = [bad_cam] + ra[i][8]
                return self._semantic_post_issue(prot, sim, ra[i][6][0], pseudo_boxes[i], feats[i], gt_labels[i], 0.85,
                                                 self.num_semantic_points, extra=extra,
                                                 flag_slot=None if flag_slots is None else flag_slots[i, 2:3],
                                                 flag_vec=None if flag_slots is None else flag_slots[i])

            def chain_finish(i, st):
                sc = self._semantic_post_finish(st)
                if sc is None:
                    if st["extra"][0]:
                        raise RuntimeError("seed_pseudo_gt: a CAM has no foreground component (constant attention map)")
                    if mtdev:                                    # the WHOLE call again on the host path, in stream order
                        raise _HostDrawsNeeded()
                    r = phase_a_finish(i, phase_a(i))            # a rare branch needs host logic: synchronous path
                    prot, sim = self.mean_shift_batch([r[6][1]], [feats[i]], [pseudo_boxes[i]],
                                                      self.mean_shift_times_local)[0]
                    sc = self._semantic_post(prot, sim, r[6][0], pseudo_boxes[i], feats[i], gt_labels[i], 0.85,
                                             self.num_semantic_points)
                    return r[:6] + (sc, _to_host_finish(r[7]))
                return ra[i][:6] + (sc, _to_host_finish(ra[i][7]))

            # every image's merge inputs and readback are queued before the host waits for the first one: image i+1's
            # device work and copy run while the host resolves image i's parts
            for st in self._streams[:num_imgs]:
                st.wait_stream(main)
            pend = [on_stream(i, chain_issue) for i in range(num_imgs)]
            results = [on_stream(i, chain_finish, pend[i]) for i in range(num_imgs)]
            for st in self._streams[:num_imgs]:
                main.wait_stream(st)
            ra = None
            if mtdev:                                            # hand the advanced engine back to torch's global generator
                torch.set_rng_state(_MT.pack_state(mt_blob, _to_host_finish(mt_final)))
                self.rng_stats["device_calls"] += 1
        elif multi:
            # One HIP stream per image, one host thread.  Every image's device work is queued first and only then are
            # the counts read back: a host sync waits for ITS image's stream only, so image i+1's refinement runs on
            # the device while the host draws and resolves image i's points (the draw ORDER across images changes,
            # which only the literal reference stream forbids), and the small kernels of different images overlap.
            if len(self._streams) < num_imgs:
                self._streams = [torch.cuda.Stream() for _ in range(num_imgs)]
            main = torch.cuda.current_stream()

            def on_stream(i, fn, *args):
                with torch.cuda.stream(self._streams[i]):
                    return fn(i, *args)

            for st in self._streams[:num_imgs]:
                st.wait_stream(main)                         # CAM maps / boxes were produced on the caller's stream
            issued = [on_stream(i, phase_a) for i in range(num_imgs)]
            ra = [on_stream(i, phase_a_finish, issued[i]) for i in range(num_imgs)]
            for st in self._streams[:num_imgs]:
                main.wait_stream(st)                         # the batched mean shift below consumes every image's seeds
        elif self.rng_mode == "fast" and num_imgs > 1 and not self.parallel_images:
            issued = [phase_a(i) for i in range(num_imgs)]
            ra = [phase_a_finish(i, issued[i]) for i in range(num_imgs)]
        else:
            ra = self._run_images(lambda i: phase_a_finish(i, phase_a(i)), num_imgs)
        if ra is not None and bool(bad_cam):
            raise RuntimeError("seed_pseudo_gt: a CAM has no foreground component (constant attention map)")
        if ra is None:
            pass                                             # the sync-free path has already produced `results`
        elif self.batch_mean_shift or num_imgs == 1:         # ONE mean-shift call for the whole batch
            shifted = self.mean_shift_batch([r[6][1] for r in ra], feats, pseudo_boxes, self.mean_shift_times_local,
                                            feat_tok=feat_tok)
        else:
            shifted = [self.mean_shift_batch([ra[i][6][1]], [feats[i]], [pseudo_boxes[i]], self.mean_shift_times_local)[0]
                       for i in range(num_imgs)]
        CLOCK.mark("semantic_centers")

        def image_chain(i):
            prot, sim = shifted[i]
            sc = self._semantic_post(prot, sim, ra[i][6][0], pseudo_boxes[i], feats[i], gt_labels[i], 0.85,
                                     self.num_semantic_points)
            mask_np = _to_host_finish(ra[i][7])
            CLOCK.mark("pseudo_masks")
            return ra[i][:6] + (sc, mask_np)

        if ra is None:
            pass
        elif multi:
            for st in self._streams[:num_imgs]:
                st.wait_stream(main)                         # shifted prototypes come from the caller's stream
            results = [on_stream(i, image_chain) for i in range(num_imgs)]
            for st in self._streams[:num_imgs]:
                main.wait_stream(st)
        else:
            results = self._run_images(image_chain, num_imgs)
        for res in results:
            coord_point, labels_point, map_fg, map_bg, feats_fg, feats_bg, sc, mask_np = res
            (centers, centers_split, sim_fg, feat_split, feat_centers, num_parts_obj, c_org, l_org, corres) = sc
            out["semantic_centers_feat_split"].append(feat_split)
            out["mask_points_coords"].append(coord_point)
            out["mask_points_labels"].append(labels_point)
            out["map_cos_fg"].append(map_fg[-1])
            map_cos_bg_ret.append(map_bg[-1])
            out["semantic_centers"].append(centers)
            out["semantic_centers_split"].append(centers_split)
            sim_fg_ret.append(sim_fg)
            out["semantic_centers_feat"].append(feat_centers)
            out["num_parts"].append(num_parts_obj)
            coords_sc_org.append(c_org)
            labels_sc_org.append(l_org)
            out["corres_gts"].append(corres)
            out["pseudo_gt_masks"].append(mask_np)
            out["inst_fg_feat"].append(feats_fg)
            out["inst_bg_feat"].append(feats_bg)
        out["semantic_centers_org"] = (coords_sc_org, labels_sc_org)
        if self.visualize:
            out.update(map_cos_bg=map_cos_bg_ret, sim_fg=sim_fg_ret, attns=attn_maps_dealed[-1])   # cams_up kept above
        return out


def _ns(cfg):
    if cfg is None or isinstance(cfg, types.SimpleNamespace):
        return cfg
    if isinstance(cfg, dict):
        return types.SimpleNamespace(**{k: v for k, v in cfg.items()})
    return cfg


def _get(obj, key, default=None):
    if isinstance(obj, dict):
        return obj.get(key, default)
    return getattr(obj, key, default)
