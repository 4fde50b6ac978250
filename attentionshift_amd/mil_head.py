"""The producer of `gt_box_index`: the MIL head that picks, per object, which roll-out depth's CAM box
becomes the pseudo box.  Host-side torch modules (a handful of small GEMMs per step), trainable, state-dict compatible.

    MAEBoxHeadMIL      mmdet/models/roi_heads/bbox_heads/mae_bbox_head_mil.py:19-170 (only the layers its forward
                       uses exist in the reference module as well: norm, decoder_embed, fc1, fc2, the two branches)
    roi_align          mmcv.ops.RoIAlign as configured at configs/mae/attnshift_voc12aug.py:64-68 (output 7x7,
                       sampling_ratio 0 = adaptive, aligned=True, average pooling).  mmcv-full 1.3.8 is not in the
                       reference tree: restated from its published algorithm, parity unpinned.
    MILLayerSelector   _mil_forward_train (stdroi:2953-2972) in the shape of AttnShiftRoIHead.layer_selector
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import HEADS


def _trunc_normal_(tensor, std=0.02):
    return nn.init.trunc_normal_(tensor, std=std, a=-2.0, b=2.0)


class _RoIAlignFn(torch.autograd.Function):
    """as_roi_align_fwd / as_roi_align_bwd under autograd.  feat [B,C,H,W] (any strides) -> [R, out*out, C]."""

    @staticmethod
    def forward(ctx, feat, rois, out, scale, sampling_ratio, aligned):
        from . import ops
        nhwc = feat.permute(0, 2, 3, 1).contiguous()             # a no-op copy when feat views token-major storage
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(nhwc.shape), out, scale, sampling_ratio, aligned)
        return ops.roi_align_fwd(nhwc, rois, out, scale, sampling_ratio, aligned)

    @staticmethod
    def backward(ctx, dout):
        from . import ops
        (rois,) = ctx.saved_tensors
        shape, out, scale, sampling_ratio, aligned = ctx.cfg
        dfeat = ops.roi_align_bwd(dout.contiguous().float(), rois, shape, out, scale, sampling_ratio, aligned)
        return dfeat.permute(0, 3, 1, 2), None, None, None, None, None


def roi_align(feat, rois, output_size=7, spatial_scale=1.0 / 16, sampling_ratio=0, aligned=True, max_bytes=1 << 28):
    """feat [B,C,H,W], rois [R,5] (batch index, x1, y1, x2, y2 in image coordinates) -> [R,C,out,out].
    Average of bilinear samples on a regular grid inside each bin; `sampling_ratio=0` uses ceil(roi size / out)
    samples per bin and axis; `aligned` shifts the box by half a pixel (no minimum size of 1); a RoI whose adaptive
    sample grid is empty (non-positive size with aligned=True) gives 0, as mmcv's kernel does.
    The gathered rows are [R, C, out*g, W]: RoIs are processed in chunks so that this stays under `max_bytes`
    (512 sampled RoIs x 2 images x 384 channels on an 84x50 map would otherwise be several GB)."""
    out = output_size
    B, C, H, W = feat.shape
    R = rois.shape[0]
    if R == 0:
        return feat.new_zeros(0, C, out, out)
    # the HIP backward keeps fixed pixels x 8 channels per thread and its 1-D tables in LDS (csrc/roi_align.hip): outside its
    # limits a trainable map takes the tensor-op path below instead of failing in backward
    bwd_ok = (not (feat.requires_grad and torch.is_grad_enabled())) or (
        C % 8 == 0 and H * W <= 8192 and 2 * out <= 1024 and
        # as_roi_align_bwd shrinks its RoI batch until the tables fit: the limit is ONE RoI's tables (padded to 16 bytes)
        (((out * (H + W) + 3) & ~3) + out * out * 8 + 4 + (H + W) + 4 + R) * 4 <= 150 * 1024)
    if feat.is_cuda and C % 4 == 0 and bwd_ok:
        # the HIP kernel (csrc/roi_align.hip): token-major in, [R, out*out, C] out; the [R, C, out, out] the heads expect
        # is a permuted VIEW of it (they flatten straight back to tokens)
        y = _RoIAlignFn.apply(feat.float(), rois.float().contiguous(), out, float(spatial_scale), int(sampling_ratio), bool(aligned))
        return y.view(R, out, out, C).permute(0, 3, 1, 2).to(feat.dtype)
    per_roi = C * max(H, out * 2) * W * feat.element_size() * 2        # feat[bidx] row + its gathered rows (g ~ 2)
    chunk = max(1, int(max_bytes // max(per_roi, 1)))
    if R > chunk:
        return torch.cat([_roi_align_chunk(feat, rois[i:i + chunk], out, spatial_scale, sampling_ratio, aligned)
                          for i in range(0, R, chunk)])
    return _roi_align_chunk(feat, rois, out, spatial_scale, sampling_ratio, aligned)


def _roi_align_chunk(feat, rois, out, spatial_scale, sampling_ratio, aligned):
    B, C, H, W = feat.shape
    R = rois.shape[0]
    off = 0.5 if aligned else 0.0
    bidx = rois[:, 0].long()
    x1, y1 = rois[:, 1] * spatial_scale - off, rois[:, 2] * spatial_scale - off
    x2, y2 = rois[:, 3] * spatial_scale - off, rois[:, 4] * spatial_scale - off
    rw, rh = x2 - x1, y2 - y1
    if not aligned:
        rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
    bw, bh = rw / out, rh / out
    empty = ((torch.ceil(rw / out) <= 0) | (torch.ceil(rh / out) <= 0)) if sampling_ratio <= 0 else torch.zeros_like(bidx, dtype=torch.bool)
    gw = torch.ceil(rw / out).clamp(min=1).long() if sampling_ratio <= 0 else torch.full_like(bidx, sampling_ratio)
    gh = torch.ceil(rh / out).clamp(min=1).long() if sampling_ratio <= 0 else torch.full_like(bidx, sampling_ratio)
    gmax_w, gmax_h = int(gw.max()), int(gh.max())
    dev, dt = feat.device, feat.dtype
    ph = torch.arange(out, device=dev, dtype=dt)
    iy = torch.arange(gmax_h, device=dev, dtype=dt)
    ix = torch.arange(gmax_w, device=dev, dtype=dt)
    # sample coordinates [R, out, g]
    ys = y1[:, None, None] + ph[None, :, None] * bh[:, None, None] + (iy[None, None, :] + 0.5) * bh[:, None, None] / gh[:, None, None]
    xs = x1[:, None, None] + ph[None, :, None] * bw[:, None, None] + (ix[None, None, :] + 0.5) * bw[:, None, None] / gw[:, None, None]
    vy = iy[None, None, :] < gh[:, None, None]                       # samples that exist for this RoI
    vx = ix[None, None, :] < gw[:, None, None]

    def axis(coord, n):
        inside = (coord >= -1.0) & (coord <= n)                      # outside: the sample contributes 0
        c = coord.clamp(min=0.0)
        lo = c.floor().long().clamp(max=n - 1)
        hi = (lo + 1).clamp(max=n - 1)
        c = torch.where(lo >= n - 1, lo.to(c.dtype), c)
        frac = c - lo.to(c.dtype)
        return lo, hi, 1.0 - frac, frac, inside

    ylo, yhi, wy0, wy1, iny = axis(ys, H)
    xlo, xhi, wx0, wx1, inx = axis(xs, W)
    wy0, wy1 = wy0 * (vy & iny), wy1 * (vy & iny)
    wx0, wx1 = wx0 * (vx & inx), wx1 * (vx & inx)
    fm = feat[bidx]                                                  # [R, C, H, W]
    # gather rows then columns: rows [R, C, out*g_h, W]
    def rows(idx):
        return fm.gather(2, idx.reshape(R, 1, -1, 1).expand(R, C, out * gmax_h, W))
    fy = rows(ylo) * wy0.reshape(R, 1, -1, 1) + rows(yhi) * wy1.reshape(R, 1, -1, 1)          # [R,C,out*gh,W]
    def cols(idx):
        return fy.gather(3, idx.reshape(R, 1, 1, -1).expand(R, C, out * gmax_h, out * gmax_w))
    fxy = cols(xlo) * wx0.reshape(R, 1, 1, -1) + cols(xhi) * wx1.reshape(R, 1, 1, -1)         # [R,C,out*gh,out*gw]
    fxy = fxy.reshape(R, C, out, gmax_h, out, gmax_w).sum(dim=(3, 5))
    res = fxy / (gh * gw).to(dt)[:, None, None, None]
    return torch.where(empty[:, None, None, None], torch.zeros_like(res), res)


@HEADS.register_module()
class MAEBoxHeadMIL(nn.Module):
    """Multiple-instance head over the Lc candidate boxes of every object (mae_bbox_head_mil.py).  Constructor kwargs
    of the reference are accepted; those that only configure unused MAE-decoder parts are ignored."""

    def __init__(self, in_channels=384, img_size=224, patch_size=16, embed_dim=256, depth=4, num_heads=8, mlp_ratio=4.,
                 num_classes=20, num_layers_query=12, loss_mil_factor=1.0, hidden_dim=1024, roi_size=7, pretrained=False,
                 **kwargs):
        super().__init__()
        self.pretrained, self.init_cfg = pretrained, kwargs.get("init_cfg")
        self.num_classes, self.num_layers_query = num_classes, num_layers_query
        self.loss_mil_factor, self.hidden_dim, self.roi_size = loss_mil_factor, hidden_dim, roi_size
        self.with_decoder_embed = in_channels != embed_dim
        if self.with_decoder_embed:
            self.norm = partial(nn.LayerNorm, eps=1e-6)(in_channels)
            self.decoder_embed = nn.Linear(in_channels, embed_dim, bias=True)
        self.fc1 = nn.Linear(embed_dim * roi_size ** 2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.proposal_branch = nn.Linear(hidden_dim, num_classes)
        self.classification_branch = nn.Linear(hidden_dim, num_classes)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def init_weights(self, pretrained=None):
        """mae_bbox_head_mil.py:73-103: `pretrained=True` + a checkpoint path loads the MAE decoder entries that exist in
        this head (decoder_embed, norm); otherwise the constructor's initialisation stands."""
        from .mae_heads import load_mae_decoder_weights
        path = pretrained if isinstance(pretrained, str) else (self.init_cfg or {}).get("checkpoint")
        if self.pretrained and isinstance(path, str):
            load_mae_decoder_weights(self, path)
        elif path is not None and not isinstance(path, str):
            raise TypeError("pretrained must be a str or None")

    def mil_losses(self, cls_score, labels):
        cls_score = cls_score.clamp(1e-6, 1 - 1e-6)
        labels = labels.clamp(0, 1)
        return (-labels * torch.log(cls_score) - (1 - labels) * torch.log(1 - cls_score)).mean()

    def forward(self, x, gt_labels=None):
        """x [(sum G) * Lc, C, r, r] RoI features, object-major; returns (gt_index [sum G], mil_loss)."""
        if isinstance(gt_labels, list):
            gt_labels = torch.cat(gt_labels)
        n = x.shape[0]
        x = x.flatten(2).transpose(1, 2)
        if self.with_decoder_embed:
            x = self.decoder_embed(self.norm(x))
        x = F.relu(self.fc1(x.reshape(n, -1)))
        x = F.relu(self.fc2(x))
        Lq, K = self.num_layers_query, self.num_classes
        cls = self.classification_branch(x).reshape(-1, Lq, K).softmax(-1)       # over classes
        prop = self.proposal_branch(x).reshape(-1, Lq, K).softmax(-2)             # over the Lc candidates
        bag = cls * prop
        picked = torch.gather(bag, -1, gt_labels.reshape(-1, 1, 1).repeat(1, Lq, 1))[..., 0]
        gt_index = picked.max(-1)[1]
        onehot = torch.zeros(len(gt_labels), K, dtype=bag.dtype, device=bag.device).scatter_(1, gt_labels.reshape(-1, 1), 1.0)
        return gt_index, self.loss_mil_factor * self.mil_losses(bag.sum(1), onehot)


class MILLayerSelector:
    """`layer_selector` for AttnShiftRoIHead: RoI-align the stride-16 feature map on every candidate box, run the MIL
    head, return the chosen depth per object (stdroi:2953-2972).  The loss of the last call is kept in `.last_loss`
    (the reference returns it as losses['mil_loss'])."""

    def __init__(self, mil_head, output_size=7, stride=16, sampling_ratio=0):
        self.head, self.output_size, self.stride, self.sampling_ratio = mil_head, output_size, stride, sampling_ratio
        self.last_loss = None

    def __call__(self, boxes_per_img, labels_per_img, roi_feature_map):
        fmap = roi_feature_map[0] if isinstance(roi_feature_map, (list, tuple)) else roi_feature_map
        rois = torch.cat([torch.cat((b.new_full((b.shape[0] * b.shape[1], 1), float(i)), b.reshape(-1, 4)), dim=1)
                          for i, b in enumerate(boxes_per_img)])                   # bbox2roi of [G_i, Lc, 4]
        feats = roi_align(fmap.float(), rois, self.output_size, 1.0 / self.stride, self.sampling_ratio, True)
        gt_index, self.last_loss = self.head(feats, gt_labels=list(labels_per_img))
        return list(gt_index.split([b.shape[0] for b in boxes_per_img], dim=0))
