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
  B2', B3, B5, B6 (point sampling, erosion, part filtering/merging, masks): small data-dependent
      host logic kept in torch ops on the device, drawing random numbers from torch's global CPU
      generator in exactly the reference's order (so equal seeds give equal samples).
The trainable MIL / bbox / mask sub-heads are out of this path's scope: their configs are
accepted and kept, and the one value the path needs from the MIL head -- which roll-out depth to use per
object -- comes from `layer_selector` (a callable; default: the depth whose CAM box has the median area).
"""
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .registry import HEADS

STRIDE = 16


# --------------------------------------------------------------------------------------------------
# host-side matching (stdroi:2237-2257; HungarianPointAssigner mmdet/core/bbox/assigners/
# hungarian_point_assigner.py:54-113, FocalLossCost / PointL1Cost match_cost.py:52-104)
# --------------------------------------------------------------------------------------------------
def hungarian_point_match(point_pred, cls_pred, gt_points, gt_labels, img_shape, cls_weight=1.0, reg_weight=10.0,
                          alpha=0.25, gamma=2.0, eps=1e-12):
    """Returns (pos_inds ascending [G'], matched gt index per pos_ind [G'])."""
    from scipy.optimize import linear_sum_assignment
    if gt_points.shape[0] == 0 or point_pred.shape[0] == 0:
        z = torch.zeros(0, dtype=torch.long, device=point_pred.device)
        return z, z
    img_h, img_w = img_shape[:2]
    factor = gt_points.new_tensor([img_w, img_h]).unsqueeze(0)
    p = cls_pred.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    cost = (pos[:, gt_labels] - neg[:, gt_labels]) * cls_weight + torch.cdist(point_pred, gt_points / factor, p=1) * reg_weight
    rows, cols = linear_sum_assignment(cost.detach().cpu().numpy())
    order = np.argsort(rows)
    dev = point_pred.device
    return torch.as_tensor(rows[order], device=dev, dtype=torch.long), torch.as_tensor(cols[order], device=dev, dtype=torch.long)


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


def _minmax_maps(a):
    """stdroi:329-333 norm_attns."""
    flat = a.flatten(1)
    lo, hi = flat.min(1)[0][:, None, None], flat.max(1)[0][:, None, None]
    return (a - lo) / (hi - lo)


def sample_point_grid(maps, num_points, thr, is_pos, gt_points=None):
    """stdroi:343-371.  Random draws come from torch's global CPU generator exactly like the reference."""
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
        pick = (torch.randint(n, (n_draw,)) % n).to(coords.device)
        out.append(coords[pick][:num_points])
    return torch.stack(out).flip(-1)


def seed_features(point_xy, feat_chw):
    """stdroi:335-338: mean feature under the sampled pixels.  point_xy [G',K,2] (x,y)."""
    C, hp, wp = feat_chw.shape
    py = (point_xy[..., 1].long() // STRIDE).clamp(0, hp)
    px = (point_xy[..., 0].long() // STRIDE).clamp(0, wp)
    return feat_chw.permute(1, 2, 0)[py, px].mean(dim=1)


def mask_points_fg_bg(map_fg, map_bg, pos_thr, neg_thr, num_gt, corr_size):
    """stdroi:433-461 on one crop."""
    dev = map_fg.device
    pos = _erode((map_fg > map_fg.max() * pos_thr).float(), corr_size).nonzero()
    neg = (map_bg > map_bg.max() * neg_thr).nonzero()
    both = torch.cat((pos, neg), dim=0)
    lab = torch.cat((torch.ones(pos.shape[0], dtype=torch.bool, device=dev), torch.zeros(neg.shape[0], dtype=torch.bool, device=dev)))
    pick = torch.randperm(both.shape[0])[:num_gt].to(dev)
    if pick.shape[0] < num_gt:
        if pick.shape[0] == 0:
            return -torch.ones(num_gt, 2, dtype=torch.float, device=dev), torch.zeros(num_gt, dtype=torch.bool, device=dev)
        pick = _fill_in(pick, num_gt)       # NB: the reference's 1-D repeat bug (2-D index) is not reproduced
    return both[pick], lab[pick]


def grid_seed_coords(maps, rois, thr=0.35, n_points=20):
    """stdroi:1784-1810: (y,x) patch coords of n_points grid-strided positives per object."""
    out = []
    for g, m in enumerate(maps):
        pos = (m >= thr).nonzero()
        n = pos.shape[0]
        if n >= n_points:
            c = pos[torch.arange(0, n, step=n // n_points, device=pos.device)[:n_points]]
        elif n > 0:
            c = _fill_in(pos, n_points)
        else:
            c = ((rois[g][:2] + rois[g][2:]) // (2 * STRIDE)).long().view(1, 2).flip(1).repeat(n_points, 1)
        out.append(c)
    return torch.stack(out)


def _unit(x):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-8)


def filter_parts(sim, fg_inter, pos_thr=0.85):
    """stdroi:265-275 filter_maps."""
    support = (sim > 0.8).to(sim.dtype)
    score = (fg_inter[:, None] * support).sum(dim=[-2, -1]) / support.sum(dim=[-2, -1]).clamp(1e-6)
    return score >= pos_thr


def merge_parts(prot_list, thr):
    """stdroi:278-294 merge_maps."""
    out = []
    for prot in prot_list:
        if prot.shape[0] == 0:
            out.append([])
            continue
        u = _unit(prot)
        link = (torch.triu(u @ u.t(), diagonal=0) >= thr).to(prot.dtype)
        merged = []
        for i in range(link.shape[0]):
            wgt = link[i].clone()
            if wgt.sum() > 0:
                merged.append((wgt @ prot) / (wgt.sum() + 1e-8))
            link[wgt > 0] *= 0
        out.append(torch.stack(merged))
    return out


def part_similarity(prot, feat_chw):
    """stdroi:297-301 cal_similarity."""
    if isinstance(prot, list):
        return torch.zeros(0, 0)
    C, hp, wp = feat_chw.shape
    return (_unit(prot) @ _unit(feat_chw.flatten(1).t()).t()).reshape(-1, hp, wp)


def part_centers(maps, rois, obj_label, feat_chw, num_max_keep=50, num_max_obj=3):
    """stdroi:222-262 get_center_coord_with_feat (same eight outputs, same order)."""
    coords, labels, feats, owner = [], [], [], []
    split = [0 for _ in range(len(maps))]
    for g, m in enumerate(maps):
        if m.shape[0] == 0:
            continue
        peak = m.flatten(1).topk(dim=1, k=1)[0][:, -1, None, None]
        at_peak = (m >= peak).nonzero().float()
        x0, y0, x1, y1 = rois[g]
        order = (m > 0.9).sum(dim=[-2, -1]).argsort(descending=True, dim=0, stable=True)
        for i in range(m.shape[0]):
            if i > num_max_obj:
                break
            xy = at_peak[at_peak[:, 0] == order[i]].mean(dim=0)[1:].flip(0)
            c = (xy + 0.5) * STRIDE
            if (c[0] >= x0) & (c[0] <= x1) & (c[1] >= y0) & (c[1] <= y1):
                coords.append(c)
                labels.append(obj_label[g])
                owner.append(g)
                feats.append(feat_chw[:, xy[1].long(), xy[0].long()])
                split[g] += 1
    dev = rois[0].device
    if len(coords) == 0:
        z2 = torch.zeros(0, 2, dtype=rois[0].dtype, device=dev)
        zl = torch.zeros(0, dtype=obj_label[0].dtype, device=dev)
        return [z2, zl], [], [], [], split, z2.clone(), zl.clone(), torch.zeros(0, dtype=torch.long, device=dev)
    coords, labels, feats = torch.stack(coords), torch.stack(labels), torch.stack(feats)
    coords_org, labels_org = coords.clone(), labels.clone()
    coord_split, feats_split = list(coords.split(split, dim=0)), list(feats.split(split, dim=0))
    if coords.shape[0] > num_max_keep:
        pick = torch.randperm(coords.shape[0], device=coords.device)[:num_max_keep]
        coords, labels = coords[pick], labels[pick]
    return ([coords, labels], coord_split, feats_split, feats, split, coords_org, labels_org,
            torch.tensor(owner, device=dev, dtype=torch.long))


def median_area_selector(boxes_per_img, labels_per_img=None, roi_feature_map=None):
    """Stand-in for the trainable MIL head's choice (mae_bbox_head_mil.py:140-169): per object the roll-out
    depth whose CAM box has the median area.  boxes [G,Lc,4] -> index [G]."""
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
                 layer_selector=None):
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
        self.with_mil = mil_head is not None
        self.with_mask = mask_head is not None
        self.with_bbox = bbox_head is not None
        self.layer_selector = layer_selector or median_area_selector
        self.visualize = visualize
        self.epoch, self.epoch_semantic_centers = epoch, epoch_semantic_centers
        self.num_semantic_points = num_semantic_points
        self.semantic_to_token, self.pca_dim = semantic_to_token, pca_dim
        self.mean_shift_times_local = mean_shift_times_local
        self.num_reppoints_head = num_reppoints_head

    def forward_train(self, *a, **k):
        raise NotImplementedError("forward_train (trainable bbox/mask heads) is outside the hot path: SURVEY 8f")

    def simple_test(self, *a, **k):
        raise NotImplementedError("inference heads are outside the hot path: SURVEY 8f")

    # ---- stage helpers ----------------------------------------------------------------------------
    def rollout_cams(self, attns, num_proposals):
        """A3 (stdroi:2261): [B, Lc, T, N] rows of the roll-out for the T point tokens."""
        Lc = self.bbox_head.cam_layer
        states = attns[-Lc:]
        if not isinstance(states[0], ops.AttnLayerState):
            raise TypeError("attns must be the AttnLayerState handles returned by the MI355X VisionTransformerDet "
                            "(dense [B,N,N] attention maps are never materialised on this path)")
        return ops.rollout_rows(states, num_proposals)

    def refine_maps(self, attn_sel, feat_chw, rois, gt_points, refine_times, obj_tau):
        """B2 (stdroi:1000-1019).  attn_sel [G,H,W], feat [C,hp,wp].  Returns map_fg, map_bg [R+1,G,H,W],
        points_fg, points_bg, fg_feat, bg_feat."""
        G = attn_sel.shape[0]
        C, hp, wp = feat_chw.shape
        nm = _minmax_maps(attn_sel)
        pts_bg = sample_point_grid(nm, 20, 0.1, False)
        pts_fg = sample_point_grid(nm, 20, 0.2, True, gt_points)
        pts_supp = sample_point_grid(nm.mean(0, keepdim=True), 20, 0.1, False)
        pts_fg = torch.cat((pts_fg, pts_supp), dim=0)
        feat_tok = feat_chw.flatten(1).t().contiguous()
        box_patch = (rois // STRIDE).to(torch.int32).contiguous()
        sim_fg, fg_feat = ops.refine_similarity(feat_tok, seed_features(pts_fg, feat_chw).contiguous(), box_patch, G,
                                                refine_times, obj_tau, True, hp, wp)
        sim_bg, bg_feat = ops.refine_similarity(feat_tok, seed_features(pts_bg, feat_chw).contiguous(), box_patch, G,
                                                refine_times, obj_tau, False, hp, wp)
        map_fg, map_bg = ops.instance_maps(sim_fg, sim_bg, G, hp, wp, STRIDE)
        return map_fg, map_bg, pts_fg, pts_bg, fg_feat[:, :, None, None], bg_feat[:, :, None, None]

    def get_mask_sample_points_roi_best_attn_feat_refine(self, attn, rois, attn_idx, vit_feat, pos_thr=0.6, neg_thr=0.6,
                                                         num_gt=20, corr_size=21, refine_times=2, obj_tau=0.85,
                                                         gt_points=None):
        """stdroi:1966-1993 (same argument meaning and return order)."""
        G = attn.shape[1]
        attn_sel = attn[attn_idx, torch.arange(G, device=attn.device)].contiguous()
        map_fg, map_bg, pts_a, pts_b, f_fg, f_bg = self.refine_maps(attn_sel, vit_feat, rois, gt_points, refine_times, obj_tau)
        cs, ls = [], []
        for g in range(G):
            x0, y0, x1, y1 = rois[g].int().tolist()
            c, l = mask_points_fg_bg(map_fg[-1][g][y0:y1, x0:x1], map_bg[-1][g][y0:y1, x0:x1], pos_thr, neg_thr, num_gt, corr_size)
            c = c.clone()
            c[:, 0] += y0
            c[:, 1] += x0
            cs.append(c.flip(1))
            ls.append(l)
        return torch.stack(cs).float(), torch.stack(ls), map_fg, map_bg, pts_a, pts_b, f_fg, f_bg

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

    def get_semantic_centers(self, map_cos_fg, map_cos_bg, rois, vit_feat, pos_thr=0.35, refine_times=5, gt_labels=None,
                             merge_thr=0.85, num_semantic_points=3):
        """stdroi:1995-2031 (same nine outputs)."""
        hp, wp = vit_feat.shape[-2:]
        core = _erode((map_cos_fg > pos_thr).float()[None], 11)[0]
        fg_inter = F.interpolate(core.unsqueeze(0), (hp, wp), mode="bilinear")[0]
        bg_inter = F.interpolate(map_cos_bg.unsqueeze(0).max(dim=1, keepdim=True)[0], (hp, wp), mode="bilinear")[0]
        map_fg = (fg_inter > pos_thr).to(fg_inter.dtype)
        prot, sim = self.mean_shift_grid_prototype(map_fg, vit_feat, rois, tau=0.1, temp=0.1, n_shift=refine_times)
        G = map_cos_fg.shape[0]
        keep = filter_parts(sim.unflatten(0, (G, sim.shape[0] // G)), fg_inter)
        counts = keep.sum(dim=-1).tolist()
        merged = merge_parts(prot[keep.flatten()].split(counts, dim=0), thr=merge_thr)
        sim_parts = [part_similarity(p, vit_feat) for p in merged]
        (centers, split, feat_split, feats, num_parts, coords_org, labels_org, corres) = part_centers(
            sim_parts, rois, gt_labels, vit_feat, num_max_obj=num_semantic_points)
        return centers, split, sim_parts, feat_split, feats, num_parts, coords_org, labels_org, corres

    # ---- the hot-path entry point -------------------------------------------------------------------
    @torch.no_grad()
    def seed_pseudo_gt(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                       vit_feat=None, img=None, point_init=None, point_cls=None, point_reg=None, imgs_whwh=None,
                       attns=None, gt_points=None, gt_points_labels=None, roi_feature_map=None, return_mask=False,
                       pos_mask_thr=0.6, neg_mask_thr=0.1, num_mask_point_gt=10, corr_size=21, point_adjuster=None,
                       edges=None, obj_tau=0.85, pos_inds=None, matched_gt=None):
        """stdroi:2209-2415.  Extra optional inputs `pos_inds` / `matched_gt` (per-image lists) bypass the
        Hungarian matching when the caller already has it (fixtures, benchmarks)."""
        num_imgs = point_reg.size(0)
        num_proposals = point_cls.size(1)
        if pos_inds is None:
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
        rows = self.rollout_cams(attns, num_proposals)                       # [B, Lc, T, N]
        counts = [int(p.numel()) for p in pos_inds]
        # B1, batched over every (image, layer, object): one launch sequence for the whole batch
        cams_lr = torch.cat([rows[i][:, pos_inds[i], 1:-num_proposals].reshape(-1, patch_h, patch_w)
                             for i in range(num_imgs)]).contiguous()
        pts = torch.cat([point_targets[i].float().repeat(Lc, 1) for i in range(num_imgs)]).contiguous()
        if cams_lr.shape[0] == 0:
            raise RuntimeError("seed_pseudo_gt: no matched point tokens in the batch")
        boxes, status, cams_up = ops.cam_boxes(cams_lr, pts, self.bbox_head.seed_thr, self.bbox_head.seed_multiple,
                                               STRIDE, True)
        if bool((status == 0).any()):
            # the reference raises here too (torch.stack of an empty list, stdroi:80)
            raise RuntimeError("seed_pseudo_gt: a CAM has no foreground component (constant attention map)")
        gt_scale_bboxes, attn_maps_dealed, off = [], [], 0
        for i in range(num_imgs):
            n = Lc * counts[i]
            gt_scale_bboxes.append(boxes[off:off + n].reshape(Lc, counts[i], 4).permute(1, 0, 2).contiguous())
            attn_maps_dealed.append(cams_up[off:off + n].reshape(Lc, counts[i], H, W))
            off += n

        gt_box_index = self.layer_selector(gt_scale_bboxes, gt_labels, roi_feature_map)
        pseudo_boxes = [gt_scale_bboxes[i][torch.arange(counts[i], device=boxes.device), gt_box_index[i]]
                        for i in range(num_imgs)]
        mil_losses = {}

        out = dict(pseudo_gt_labels=gt_labels, pseudo_gt_bboxes=pseudo_boxes, mil_losses=mil_losses,
                   best_attn_idx=gt_box_index, map_cos_fg=[], mask_points_coords=[], mask_points_labels=[],
                   semantic_centers=[], semantic_centers_split=[], semantic_centers_feat_split=[],
                   semantic_centers_feat=[], num_parts=[], pseudo_gt_masks=[], corres_gts=[], inst_fg_feat=[],
                   inst_bg_feat=[])
        coords_sc_org, labels_sc_org, map_cos_bg_ret, sim_fg_ret = [], [], [], []
        for i in range(num_imgs):
            feat = vit_feat[i].float()
            if not feat.is_contiguous():
                feat = feat.contiguous()
            (coord_point, labels_point, map_fg, map_bg, _pb, _pf, feats_fg, feats_bg) = \
                self.get_mask_sample_points_roi_best_attn_feat_refine(
                    attn_maps_dealed[i], pseudo_boxes[i], gt_box_index[i], vit_feat=feat, pos_thr=pos_mask_thr,
                    neg_thr=neg_mask_thr, num_gt=num_mask_point_gt, corr_size=corr_size, obj_tau=obj_tau,
                    gt_points=gt_points[i])
            (centers, centers_split, sim_fg, feat_split, feat_centers, num_parts_obj, c_org, l_org, corres) = \
                self.get_semantic_centers(map_fg[-1].clone(), map_bg[-1].clone(), pseudo_boxes[i], feat,
                                          pos_thr=pos_mask_thr, refine_times=self.mean_shift_times_local,
                                          gt_labels=gt_labels[i], num_semantic_points=self.num_semantic_points)
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
            peak = map_fg[-1].flatten(1).max(1)[0][:, None, None]
            out["pseudo_gt_masks"].append((map_fg[-1] > peak * pos_mask_thr).to(torch.uint8).cpu().numpy())   # stdroi:2356
            out["inst_fg_feat"].append(feats_fg)
            out["inst_bg_feat"].append(feats_bg)
        out["semantic_centers_org"] = (coords_sc_org, labels_sc_org)
        if self.visualize:
            out.update(map_cos_bg=map_cos_bg_ret, sim_fg=sim_fg_ret, attns=attn_maps_dealed[-1])
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
