"""MAE-decoder RoI heads on the small-N batched attention kernel.

    SmallAttnFn       autograd bridge of as_small_attn_fwd / as_small_attn_bwd
    DecoderBlock      models/vision_transformer.py:88-124 `Block` (pre-LN attention + MLP) with the reference's
                      parameter names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2); the attention core runs on
                      the HIP kernel, LayerNorm / Linear / GELU are library ops (four 256-wide GEMMs per block)
    MAEBoxHeadRec     mmdet/models/roi_heads/bbox_heads/mae_bbox_head_rec.py:24-168: det token + decoder over the 7x7
                      RoI tokens, `fc_cls` / `fc_reg` on the det token (the optional pixel reconstruction branch too).
                      Losses, target assignment and box coding belong to mmdet's BBoxHead and are not restated.
"""
import math
from functools import partial

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .registry import HEADS


_RESIZE = {}


def _resize_matrix(n, scale, mode, device, dtype):
    """[n * scale, n] matrix of F.interpolate(..., scale_factor=scale, mode=mode, align_corners=True) along one axis
    (obtained from ATen itself on the identity, so the coefficients are exactly the library's), cached."""
    key = (n, scale, mode, str(device), dtype)
    if key not in _RESIZE:
        eye = torch.eye(n, dtype=torch.float32).view(1, n, n, 1)           # channel = input index, H = the resized axis
        with torch.no_grad():
            a = F.interpolate(eye, scale_factor=(scale, 1), mode=mode, align_corners=True)[0, :, :, 0].t().contiguous()
        _RESIZE[key] = a.to(device=device, dtype=dtype)
    return _RESIZE[key]


def sincos_pos_embed_2d(embed_dim, grid_size, cls_token=False):
    """mmdet/models/utils/positional_encoding.py:175-225 get_2d_sincos_pos_embed (MAE / MoCo v3): [grid^2 (+1), D] float64
    numpy table; first half of the channels encodes the w coordinate (meshgrid puts w first), second half h."""
    import numpy as np

    def one_dim(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    assert embed_dim % 4 == 0
    gh = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gh, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([one_dim(embed_dim // 2, grid[0]), one_dim(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def load_mae_decoder_weights(module, pretrained, logger=None):
    """The `init_weights` loading branch shared by the three MAE-decoder heads (mae_bbox_head_rec.py:95-125,
    mae_bbox_head_mil.py:73-103, mae_mask_head_pointSup.py:108-138): take the MAE pre-training checkpoint, drop the
    encoder's `patch_embed*` / `blocks*` / `pos_embed` entries and load the rest (decoder_embed, decoder_blocks,
    decoder_pos_embed, ...) non-strictly."""
    import os
    from collections import OrderedDict
    from .checkpoint import _state_dict_of, load_state_dict
    if not os.path.isfile(pretrained):
        raise ValueError(f"checkpoint path {pretrained} is invalid")
    checkpoint = torch.load(pretrained, map_location="cpu", weights_only=False)
    sd = OrderedDict((k, v) for k, v in _state_dict_of(checkpoint).items()
                     if not (k.startswith("patch_embed") or k.startswith("blocks") or k == "pos_embed"))
    load_state_dict(module, sd, strict=False, logger=logger)


def _lin(mod, x):
    """nn.Linear `mod` on x: the HIP GEMM with the split-K weight gradient when training in a bf16 region on a GPU
    (autograd.LinearFn), the module itself otherwise (fp32 / CPU / no-grad: as the reference)."""
    from . import autograd as AG
    if torch.is_grad_enabled() and AG.linear_applies(x, mod.weight) and (mod.weight.requires_grad or x.requires_grad):
        return AG.linear(x, mod.weight, mod.bias)
    return mod(x)


def _run_blocks(blocks, x):
    """The decoder blocks on the token stream x [R, N, D].  Training in a bf16 region on a GPU: the residual stream stays
    fp32 and every residual add is fused into the LayerNorm that follows it, forward and backward
    (autograd.AddLayerNormFn, as in the backbone's blocks: models/vision_transformer.py:109-124 is the same Block);
    otherwise the modules as they are."""
    from . import autograd as AG
    if not (len(blocks) and torch.is_grad_enabled() and x.dtype == torch.float32
            and AG.linear_applies(x, blocks[0].attn.qkv.weight)):
        for blk in blocks:
            x = blk(x)
        return x
    bf, delta = torch.bfloat16, None
    fused = x.shape[0] > 0 and os.environ.get("AS_HEAD_BLOCKS_UNFUSED") is None and \
        all(AG.linear_shapes_ok(x, w) for w in (blocks[0].attn.proj.weight, blocks[0].mlp.fc1.weight, blocks[0].mlp.fc2.weight))
    for blk in blocks:
        if fused:                                           # ONE autograd node per block (autograd.DecoderBlockFn)
            x, delta = AG.decoder_block(x, delta, blk)
            continue
        x, y = AG.add_layernorm(x, delta, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, bf)     # x += previous MLP output
        x, z = AG.add_layernorm(x, blk.attn(y).contiguous(), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, bf)
        delta = blk.mlp(z)
    return torch.add(x, delta)


class SmallAttnFn(torch.autograd.Function):
    """qkv [Bp,N,3,h,32] -> out [Bp,N,h*32]."""

    @staticmethod
    def forward(ctx, qkv):
        out, lse = ops.small_attention_fwd(qkv)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, out, lse = ctx.saved_tensors
        return ops.small_attention_bwd(qkv, out, d_out, lse)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == 32, "the small-N attention kernel is built for head dim 32"
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        if B == 0:                                       # no RoIs (an image pair without positives): an empty result that
            return self.proj(self.qkv(x)[..., :C])       # still hangs on both layers' parameters, as the reference's does
        qkv = _lin(self.qkv, x).reshape(B, N, 3, self.num_heads, C // self.num_heads)   # the reference's packed layout
        return _lin(self.proj, SmallAttnFn.apply(qkv))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        from . import autograd as AG
        if (torch.is_grad_enabled() and AG.mlp_applies(x, self.fc1.weight, self.fc2.weight)
                and (self.fc1.weight.requires_grad or x.requires_grad)):
            return AG.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)     # GELU in the GEMM epilogues
        return _lin(self.fc2, F.gelu(_lin(self.fc1, x)))


class DecoderBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


@HEADS.register_module()
class MAEBoxHeadRec(nn.Module):
    def __init__(self, in_channels=384, img_size=224, patch_size=16, embed_dim=256, depth=4, num_heads=8, mlp_ratio=4.,
                 qkv_bias=True, num_classes=20, with_cls=True, with_reg=True, reg_class_agnostic=False,
                 with_reconstruct=True, seed_score_thr=0.2, seed_thr=0.2, seed_multiple=0.5, cam_layer=-1, pretrained=False,
                 init_cfg=None, **kwargs):
        super().__init__()
        self.pretrained, self.init_cfg = pretrained, init_cfg
        self.patch_size, self.num_classes = patch_size, num_classes
        self.with_cls, self.with_reg, self.with_reconstruct = with_cls, with_reg, with_reconstruct
        num_patches = (img_size // patch_size) ** 2
        self.det_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.with_decoder_embed = in_channels != embed_dim
        if self.with_decoder_embed:
            self.norm = partial(nn.LayerNorm, eps=1e-6)(in_channels)
            self.decoder_embed = nn.Linear(in_channels, embed_dim, bias=True)
        self.decoder_blocks = nn.ModuleList([DecoderBlock(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.decoder_box_norm = nn.LayerNorm(embed_dim, eps=1e-6)
        if with_cls:
            self.fc_cls = nn.Linear(embed_dim, num_classes + 1)
        if with_reg:
            self.fc_reg = nn.Linear(embed_dim, 4 if reg_class_agnostic else 4 * num_classes)
        if with_reconstruct:
            self.fc_rec = nn.Linear(embed_dim, 3 * patch_size * patch_size)
        self.seed_score_thr, self.seed_thr, self.seed_multiple, self.cam_layer = seed_score_thr, seed_thr, seed_multiple, cam_layer
        # the loss configuration the RoI head reads off the box head (mae_bbox_head_rec.py:36-70)
        self.reg_class_agnostic = reg_class_agnostic
        self.reg_decoded_bbox = kwargs.get("reg_decoded_bbox", False)
        coder = kwargs.get("bbox_coder") or {}
        self.target_means = tuple(coder.get("target_means", (0., 0., 0., 0.)))
        self.target_stds = tuple(coder.get("target_stds", (0.1, 0.1, 0.2, 0.2)))
        self.loss_cls_cfg = dict(kwargs.get("loss_cls") or dict(type="CrossEntropyLoss", loss_weight=1.0))
        self.loss_bbox_cfg = dict(kwargs.get("loss_bbox") or dict(type="L1Loss", loss_weight=1.0))
        self.loss_point_cfg = dict(kwargs.get("loss_point") or dict(type="L1Loss", loss_weight=10.0))
        self.loss_point_cls_cfg = dict(kwargs.get("loss_point_cls") or dict(type="FocalLoss", gamma=2.0, alpha=0.25,
                                                                            loss_weight=1.0))
        self.loss_weight_bbox_start = kwargs.get("loss_weight_bbox_start", 1.0)
        nn.init.trunc_normal_(self.det_token, std=.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def init_weights(self, pretrained=None):
        """mae_bbox_head_rec.py:95-125: with `pretrained=True` in the config and a checkpoint path here (or in
        init_cfg['checkpoint']) the MAE decoder weights are loaded; otherwise the constructor's truncated-normal
        initialisation stands (the reference re-applies it)."""
        path = pretrained if isinstance(pretrained, str) else (self.init_cfg or {}).get("checkpoint")
        if self.pretrained and isinstance(path, str):
            load_mae_decoder_weights(self, path)
        elif path is not None and not isinstance(path, str):
            raise TypeError("pretrained must be a str or None")

    def interpolate_pos_encoding(self, x, w, h):
        """mae_bbox_head_rec.py:126-146 (bicubic resize of the patch part, the +0.1 trick)."""
        npatch, n0 = x.shape[1] - 1, self.decoder_pos_embed.shape[1] - 1
        if npatch == n0 and w == h:
            return self.decoder_pos_embed
        cls_pe, patch_pe = self.decoder_pos_embed[:, 0], self.decoder_pos_embed[:, 1:]
        dim = x.shape[-1]
        w0, h0 = w // self.patch_size + 0.1, h // self.patch_size + 0.1
        s = int(math.sqrt(n0))
        patch_pe = F.interpolate(patch_pe.reshape(1, s, s, dim).permute(0, 3, 1, 2),
                                 scale_factor=(w0 / math.sqrt(n0), h0 / math.sqrt(n0)), mode="bicubic")
        assert int(w0) == patch_pe.shape[-2] and int(h0) == patch_pe.shape[-1]
        return torch.cat((cls_pe.unsqueeze(0), patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def forward(self, x, img=None):
        """x [R, C, 7, 7] RoI features -> (cls_score [R, K+1] | None, bbox_pred | None, img_rec | None)."""
        B, C, W, H = x.shape
        x = x.flatten(2).transpose(1, 2)
        if self.with_decoder_embed:
            x = _lin(self.decoder_embed, self.norm(x))
        x = torch.cat([self.det_token.expand(B, -1, -1), x], dim=1)
        x = x + self.interpolate_pos_encoding(x, W * self.patch_size, H * self.patch_size)
        x = _run_blocks(self.decoder_blocks, x)
        x = self.decoder_box_norm(x)
        cls_score = self.fc_cls(x[:, 0]) if self.with_cls else None
        bbox_pred = self.fc_reg(x[:, 0]) if self.with_reg else None
        img_rec = self.fc_rec(x[:, 1:]).transpose(1, 2).reshape(B, -1, W, H) if self.with_reconstruct else None
        return cls_score, bbox_pred, img_rec

    def get_targets(self, sampling_results, pos_weight=-1):
        """BBoxHead.get_targets as the RoI head calls it (stdroi:2980): (labels, label_weights, bbox_targets, bbox_weights)."""
        from .bbox_loss import bbox_targets
        return bbox_targets([r.pos_bboxes for r in sampling_results], [r.neg_bboxes for r in sampling_results],
                            [r.pos_gt_bboxes for r in sampling_results], [r.pos_gt_labels for r in sampling_results],
                            self.num_classes, self.target_means, self.target_stds, pos_weight, self.reg_decoded_bbox)

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights, pos_index=None,
             num_weighted=None):
        """mae_bbox_head_rec.py:170-221 (the reconstruction term needs with_reconstruct, off in the shipped config).
        `pos_index` / `num_weighted`: see bbox_loss.bbox_head_loss (known to a caller holding the sampling results)."""
        from .bbox_loss import bbox_head_loss
        out = bbox_head_loss(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, self.num_classes,
                             self.reg_class_agnostic, self.loss_cls_cfg.get("loss_weight", 1.0),
                             self.loss_bbox_cfg.get("loss_weight", 1.0),
                             rois=rois[:, 1:] if self.reg_decoded_bbox else None,
                             loss_bbox_type=self.loss_bbox_cfg.get("type", "L1Loss"), means=self.target_means,
                             stds=self.target_stds, pos_index=pos_index, num_weighted=num_weighted)
        return {k: (v * self.loss_weight_bbox_start if k != "acc" else v) for k, v in out.items()}


@HEADS.register_module()
class MAEMaskHeadPointSup(nn.Module):
    """mmdet/models/roi_heads/mask_heads/mae_mask_head_pointSup.py:30-273: the MAE decoder over the 14x14 RoI tokens (no
    det token), x`scale_factor` interpolation, 1x1 conv to per-class mask logits; `loss` is the point-supervised BCE on
    logits sampled at the mask points (targets built by attentionshift_amd.mask_targets)."""

    def __init__(self, roi_feat_size=14, num_classes=80, class_agnostic=False, in_channels=256, img_size=224, patch_size=16,
                 embed_dim=256, depth=4, num_heads=8, mlp_ratio=4., qkv_bias=True, scale_factor=2, scale_mode="bilinear",
                 loss_weight_mask_start=1.0, init_cfg=None, **kwargs):
        super().__init__()
        self.init_cfg = init_cfg
        self.patch_size, self.num_classes, self.class_agnostic = patch_size, num_classes, class_agnostic
        self.scale_factor, self.scale_mode, self.loss_weight_mask_start = scale_factor, scale_mode, loss_weight_mask_start
        self.roi_feat_size = (roi_feat_size, roi_feat_size) if isinstance(roi_feat_size, int) else tuple(roi_feat_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.with_decoder_embed = in_channels != embed_dim
        if self.with_decoder_embed:
            self.norm = nn.LayerNorm(in_channels, eps=1e-6)
            self.decoder_embed = nn.Linear(in_channels, embed_dim, bias=True)
        self.decoder_blocks = nn.ModuleList([DecoderBlock(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim), requires_grad=False)
        self.decoder_box_norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.conv_logits = nn.Conv2d(embed_dim, 1 if class_agnostic else num_classes, 1)
        self.apply(MAEBoxHeadRec._init_weights)
        self._fill_sincos()

    def _fill_sincos(self):
        """mae_mask_head_pointSup.py:105-106: the frozen decoder position table starts as the 2-D sin-cos encoding."""
        table = sincos_pos_embed_2d(self.decoder_pos_embed.shape[-1], int(self.num_patches ** .5), cls_token=True)
        with torch.no_grad():
            self.decoder_pos_embed.copy_(torch.from_numpy(table).float().unsqueeze(0))

    def init_weights(self):
        """mae_mask_head_pointSup.py:108-148: init_cfg=dict(type='Pretrained', checkpoint=...) loads the MAE decoder
        weights (encoder entries skipped); without it the sin-cos table + truncated-normal initialisation stand.  The
        mask-logit convolution is always re-initialised (kaiming normal, fan_out, relu; zero bias)."""
        if self.init_cfg is not None:
            assert "checkpoint" in self.init_cfg, "only init_cfg=dict(type='Pretrained', checkpoint=...) is supported"
            load_mae_decoder_weights(self, self.init_cfg["checkpoint"])
        else:
            self.apply(MAEBoxHeadRec._init_weights)
            self._fill_sincos()
        nn.init.kaiming_normal_(self.conv_logits.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.conv_logits.bias, 0)

    interpolate_pos_encoding = MAEBoxHeadRec.interpolate_pos_encoding

    def forward(self, x):
        """x [R, C, 14, 14] RoI features -> mask logits [R, K, 14 * scale, 14 * scale]."""
        B, _, W, H = x.shape
        x = x.flatten(2).transpose(1, 2)
        if self.with_decoder_embed:
            x = _lin(self.decoder_embed, self.norm(x))
        C = x.shape[-1]
        # as the reference: x carries no class token here, so `npatch = x.shape[1] - 1` never equals the table size and
        # the patch part always goes through the bicubic resize (scale (W + 0.1) / sqrt(N)), even at the native size
        x = x + self.interpolate_pos_encoding(x, W * self.patch_size, H * self.patch_size)[:, 1:]
        x = _run_blocks(self.decoder_blocks, x)
        x = self.decoder_box_norm(x).view(B, W, H, C)                      # [R, h, w, C] tokens on the RoI grid
        # F.interpolate(scale_factor, mode, align_corners=True) is a fixed separable linear map: applied as two small
        # matrix products (ATen's bicubic BACKWARD kernel needs 0.74 s for 256 RoIs x 256 channels on this GPU), and the
        # 1x1 mask-logit convolution as a matmul over the channel axis (MIOpen falls back to a naive weight-gradient
        # kernel for it).  Same operations in the same order as mae_mask_head_pointSup.py:186-189.
        # F.interpolate is not an autocast op: the reference resizes in the dtype the (fp32) LayerNorm produced, also under
        # apex O1 / autocast -- keep the resize out of the low-precision region
        with torch.autocast(x.device.type, enabled=False):
            xf = x.float()
            Ah = _resize_matrix(xf.shape[1], self.scale_factor, self.scale_mode, xf.device, xf.dtype)
            Aw = _resize_matrix(xf.shape[2], self.scale_factor, self.scale_mode, xf.device, xf.dtype)
            x = torch.einsum("oh,bhwc,pw->bopc", Ah, xf, Aw)
        wl = self.conv_logits.weight.view(self.conv_logits.out_channels, -1)
        return F.linear(x, wl, self.conv_logits.bias).permute(0, 3, 1, 2)

    def loss(self, mask_pred, mask_targets, labels):
        """mask_pred [R, K, P] logits already sampled at the points (stdroi:3154), mask_targets [R, P] with 2 = ignore."""
        if mask_pred.size(0) == 0:
            return dict(loss_mask=mask_pred.sum() * self.loss_weight_mask_start)
        cls = torch.zeros_like(labels) if self.class_agnostic else labels
        logits = mask_pred[torch.arange(mask_pred.size(0), device=mask_pred.device), cls]
        loss = F.binary_cross_entropy_with_logits(logits, mask_targets.to(torch.float32), reduction="mean",
                                                  weight=~(mask_targets == 2))
        return dict(loss_mask=loss * self.loss_weight_mask_start)
