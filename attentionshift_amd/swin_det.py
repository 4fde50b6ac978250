"""The detection-style Swin backbone of BASELINE config 5: registry name `SwinTransformer` in BACKBONES, the mmdet
signature (out_indices, per-stage output norms, tuple of NCHW feature maps, frozen_stages, init_weights(pretrained)).

Host-side mirror of mmdet/models/backbones/swin_transformer.py:448-630 (class), :300-400 (BasicLayer), :250-298
(PatchMerging), :403-445 (PatchEmbed): same constructor arguments, module names and state-dict keys
(`patch_embed.proj/norm`, `layers.i.blocks.j.{norm1,attn.{qkv,proj,relative_position_bias_table,
relative_position_index},norm2,mlp.{fc1,fc2}}`, `layers.i.downsample.{norm,reduction}`, `norm{i}`,
optional `absolute_pos_embed`), so the reference's checkpoints load.  Unlike the classifier variant (swin.py
SwinTransformer = models/swin_transformer.py) the token grid may be non-square and is padded to a window multiple per
block -- which the fused kernel `as_window_attn_fwd` does internally (pad, shift, partition, bias, mask, softmax, PV,
reverse) -- and the window is never shrunk.  LayerNorm eps is torch's default 1e-5, as the reference's norm_layer.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import BACKBONES
from . import ops
from .swin import SwinTransformerBlock, fused_path, merge_tokens, patch_embed_tokens, run_blocks


class PatchMergingDet(nn.Module):
    """mmdet/models/backbones/swin_transformer.py:250-298: pad to even, 2x2 neighbourhood -> 4C, LayerNorm, Linear 4C -> 2C."""

    def __init__(self, dim, compute_dtype=torch.bfloat16):
        super().__init__()
        self.dim, self.compute_dtype = dim, compute_dtype
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        if fused_path(self) and x.is_cuda:
            return merge_tokens(x, H, W, self.norm, self.reduction, self.compute_dtype)
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
        return self.reduction(self.norm(x))


class BasicLayerDet(nn.Module):
    """:300-400: `depth` blocks alternating shift 0 / window_size // 2 on an (H, W) grid, then the merge.
    Returns (x_out, H, W, x_down, Wh, Ww) like the reference."""

    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, downsample=False,
                 use_checkpoint=False, compute_dtype=torch.bfloat16, drop_path=0.):
        super().__init__()
        self.window_size, self.shift_size, self.depth, self.use_checkpoint = window_size, window_size // 2, depth, use_checkpoint
        dpr = list(drop_path) if isinstance(drop_path, (list, tuple)) else [drop_path] * depth        # :352
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, None, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 qkv_bias, drop_path=dpr[i], compute_dtype=compute_dtype, return_attention=False)
            for i in range(depth)])
        self.downsample = PatchMergingDet(dim, compute_dtype) if downsample else None

    def forward(self, x, H, W):
        if fused_path(self) and x.is_cuda:
            x = run_blocks(self.blocks, x, (H, W))
            if self.downsample is not None:
                return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
            return x, H, W, x, H, W
        for blk in self.blocks:
            if self.use_checkpoint and self.training and torch.is_grad_enabled():
                import torch.utils.checkpoint as cp
                x = cp.checkpoint(lambda t, b=blk: b(t, (H, W))[0], x, use_reentrant=False)
            else:
                x, _ = blk(x, (H, W))
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbedDet(nn.Module):
    """:403-445: pad the image to a patch multiple, conv patch x patch / patch, optional LayerNorm, NCHW out."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.LayerNorm(embed_dim) if norm else None

    def _pad(self, x):
        _, _, H, W = x.shape
        if W % self.patch_size[1]:
            x = F.pad(x, (0, self.patch_size[1] - W % self.patch_size[1]))
        if H % self.patch_size[0]:
            x = F.pad(x, (0, 0, 0, self.patch_size[0] - H % self.patch_size[0]))
        return x

    def tokens(self, x):
        """No-grad path: (tokens [B, Wh*Ww, C] fp32, Wh, Ww) without the NCHW round trip."""
        return patch_embed_tokens(self.proj, self.norm, self._pad(x))

    def forward(self, x):
        if fused_path(self) and x.is_cuda:
            t, Wh, Ww = self.tokens(x)
            return t.transpose(1, 2).reshape(x.shape[0], self.embed_dim, Wh, Ww)
        x = self.proj(self._pad(x))
        if self.norm is not None:
            Wh, Ww = x.shape[2:]
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(-1, self.embed_dim, Wh, Ww)
        return x


@BACKBONES.register_module(name="SwinTransformer")
class SwinTransformerDet(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=None, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False, compute_dtype=torch.bfloat16):
        super().__init__()
        if qk_scale is not None or drop_rate or attn_drop_rate:
            raise NotImplementedError("qk_scale / dropout are unused by every reference config")
        if norm_layer is not None and norm_layer is not nn.LayerNorm:
            raise NotImplementedError("norm_layer other than nn.LayerNorm")
        self.pretrain_img_size, self.num_layers, self.embed_dim = pretrain_img_size, len(depths), embed_dim
        self.ape, self.patch_norm, self.out_indices, self.frozen_stages = ape, patch_norm, tuple(out_indices), frozen_stages
        self.drop_path_rate = drop_path_rate
        # stochastic depth: the rate rises linearly over ALL blocks (:520); active on the training path only
        dpr = [drop_path_rate * i / max(sum(depths) - 1, 1) for i in range(sum(depths))]
        self.patch_embed = PatchEmbedDet(patch_size, in_chans, embed_dim, patch_norm)
        if ape:
            ps = self.patch_embed.patch_size
            pis = (pretrain_img_size, pretrain_img_size) if isinstance(pretrain_img_size, int) else tuple(pretrain_img_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, pis[0] // ps[0], pis[1] // ps[1]))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.layers = nn.ModuleList([
            BasicLayerDet(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias,
                          downsample=i < self.num_layers - 1, use_checkpoint=use_checkpoint, compute_dtype=compute_dtype,
                          drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])])
            for i in range(self.num_layers)])
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in self.out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        """:549-566."""
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.ape:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            for i in range(0, self.frozen_stages - 1):
                m = self.layers[i]
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=None):
        """:568-593: truncated-normal Linear / unit LayerNorm init, then the checkpoint (non-strict) if a path is given."""
        def _init(m):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

        if isinstance(pretrained, str):
            self.apply(_init)
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            self.apply(_init)
        else:
            raise TypeError("pretrained must be a str or None")

    def forward(self, x):
        """img [B,3,H,W] -> tuple of NCHW maps of the stages in out_indices (:595-622)."""
        if fused_path(self) and x.is_cuda:
            x, Wh, Ww = self.patch_embed.tokens(x)
            if self.ape:
                x = x + F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode="bicubic").flatten(2).transpose(1, 2)
        else:
            x = self.patch_embed(x)
            Wh, Ww = x.shape[2:]
            if self.ape:
                x = x + F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode="bicubic")
            x = x.flatten(2).transpose(1, 2)
        outs = []
        for i, layer in enumerate(self.layers):
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww)
            if i in self.out_indices:
                nrm = getattr(self, f"norm{i}")
                if fused_path(self) and x_out.is_cuda:
                    _, y = ops.add_layernorm(x_out.float().contiguous(), None, nrm.weight.detach().float(),
                                             nrm.bias.detach().float(), nrm.eps, torch.float32, want_x=False)
                else:
                    y = nrm(x_out)
                outs.append(y.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self
