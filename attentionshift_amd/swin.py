"""Swin window-attention block on the MI355X hot path (BASELINE config 5).

Host-side mirror of the reference's `WindowAttention` / `SwinTransformerBlock` (models/swin_transformer.py:77-314):
same constructor arguments, same parameter / buffer names (so reference state dicts load), same return values
`(x, attn)`.  The compute is three GEMMs (qkv WITHOUT bias on the original token grid, proj, MLP) through
`as_linear_fwd` and ONE fused kernel, `as_window_attn_fwd`, for everything between them: pad, cyclic shift, window
partition, scaled q k^T + relative-position bias + shift mask, softmax, attn @ v, window reverse, reverse shift, un-pad.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def relative_position_index(ws):
    """models/swin_transformer.py:120-130."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


class WindowAttnFn(torch.autograd.Function):
    """Fused (shifted-)window attention core under autograd: as_window_attn_fwd / as_window_attn_bwd.
    forward(qkv [B,H,W,3C] (bias-free), b_qkv fp32 [3C] | None, table fp32 [(2ws-1)^2,h], num_heads, ws, shift) -> [B,H,W,C]"""

    @staticmethod
    def forward(ctx, qkv, b_qkv, table, num_heads, ws, shift):
        qkv = qkv.contiguous()
        bq = torch.zeros(qkv.shape[-1], device=qkv.device, dtype=torch.float32) if b_qkv is None else b_qkv.contiguous()
        table = table.contiguous()
        out, _ = ops.window_attention_fwd(qkv, bq, table, num_heads, ws, shift, return_attn=False)
        ctx.save_for_backward(qkv, bq, table)
        ctx.cfg = (num_heads, ws, shift, b_qkv is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, bq, table = ctx.saved_tensors
        num_heads, ws, shift, has_bias = ctx.cfg
        dqkv, dtable, dpad = ops.window_attention_bwd(qkv, bq, table, d_out.contiguous(), num_heads, ws, shift)
        # the bias is added inside the kernel: its gradient = column sum over real tokens + the padded tokens' share
        dbias = (dqkv.float().sum(dim=(0, 1, 2)) + dpad) if has_bias else None
        return dqkv, dbias, dtable, None, None, None


class WindowAttention(nn.Module):
    """Parameter container with the reference's names (models/swin_transformer.py:92-123)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if qk_scale is not None or attn_drop or proj_drop:
            raise NotImplementedError("qk_scale / dropout are unused by every reference config")
        ws = window_size[0] if isinstance(window_size, (tuple, list)) else window_size
        self.dim, self.window_size, self.num_heads = dim, (ws, ws), num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02, a=-.04, b=.04)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class SwinTransformerBlock(nn.Module):
    """models/swin_transformer.py:182-314 (inference / no-grad semantics: drop-path inactive)."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., compute_dtype=torch.bfloat16, return_attention=True):
        super().__init__()
        # input_resolution=None: the detection variant (mmdet/models/backbones/swin_transformer.py:170-200) -- the grid is
        # given per call and padded to a window multiple inside the kernel, the window is never shrunk
        self.dim, self.num_heads = dim, num_heads
        self.input_resolution = None if input_resolution is None else tuple(input_resolution)
        self.window_size, self.shift_size = window_size, shift_size
        if self.input_resolution is not None and min(self.input_resolution) <= self.window_size:          # :211-214
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.compute_dtype = compute_dtype
        self.return_attention = return_attention
        self.drop_path = float(drop_path)                 # stochastic depth rate of this block (training path only)
        self._wcache = {}

    def _drop_path(self, t):
        """timm DropPath (models/swin_transformer.py:196, :302-304): per-sample, scaled by 1 / keep; train() + grad only."""
        if self.drop_path == 0.0 or not self.training:
            return t
        keep = 1.0 - self.drop_path
        mask = torch.rand(t.shape[0], *([1] * (t.dim() - 1)), device=t.device, dtype=torch.float32).add_(keep).floor_().div_(keep)
        return t * mask

    def _forward_train(self, x, hw=None):
        """Autograd path (train() + grad enabled): the window attention core runs on the HIP forward / backward kernels
        (WindowAttnFn), the Linear layers on the HIP GEMMs forward and backward (autograd.LinearFn, bf16 compute dtype;
        library GEMMs otherwise); LayerNorm and GELU are torch ops."""
        B, L, C = x.shape
        H, W = hw if hw is not None else (int(math.sqrt(L)),) * 2
        cd = self.compute_dtype
        from .autograd import linear_in
        y = F.layer_norm(x, (C,), self.norm1.weight, self.norm1.bias, self.norm1.eps).to(cd)
        qkv = linear_in(y, self.attn.qkv.weight, None, cd).reshape(B, H, W, 3 * C)
        bq = None if self.attn.qkv.bias is None else self.attn.qkv.bias.float()
        o = WindowAttnFn.apply(qkv, bq, self.attn.relative_position_bias_table.float(), self.num_heads, self.window_size,
                               self.shift_size)
        y = linear_in(o.reshape(B, L, C), self.attn.proj.weight, self.attn.proj.bias, cd)
        x = x + self._drop_path(y.float())
        z = F.layer_norm(x, (C,), self.norm2.weight, self.norm2.bias, self.norm2.eps).to(cd)
        from . import autograd as AG
        if cd == torch.bfloat16 and AG.linear_shapes_ok(z, self.mlp.fc1.weight) and self.mlp.fc2.weight.shape[0] % 32 == 0 \
                and not AG._UNFUSED_MLP:
            z = AG.mlp(z, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias)
        else:
            z = linear_in(F.gelu(linear_in(z, self.mlp.fc1.weight, self.mlp.fc1.bias, cd)), self.mlp.fc2.weight,
                          self.mlp.fc2.bias, cd)
        return x + self._drop_path(z.float()), None

    def _w(self, prm, dtype=None):
        """`dtype` (default: compute dtype) copy of a parameter, cached while it does not change (no-grad path only)."""
        dtype = self.compute_dtype if dtype is None else dtype
        if prm.dtype == dtype and prm.is_contiguous():
            return prm.detach()
        key = (id(prm), dtype)
        hit = self._wcache.get(key)
        if hit is None or hit[0] != prm._version or hit[1].device != prm.device:
            hit = (prm._version, prm.detach().to(dtype).contiguous())
            self._wcache[key] = hit
        return hit[1]

    def forward_pending(self, x, delta, hw):
        """No-grad path with the residual adds fused into the LayerNorms (ops.add_layernorm, as the ViT blocks):
        x fp32 [B,L,C] residual stream, delta = the previous block's MLP output not added yet (compute dtype | None).
        Returns (x after the attention residual, this block's MLP output still pending, attn | None)."""
        B, L, C = x.shape
        H, W = hw
        assert H * W == L, "token count does not match the grid"
        cd, f32 = self.compute_dtype, torch.float32
        x, y = ops.add_layernorm(x, delta, self._w(self.norm1.weight, f32), self._w(self.norm1.bias, f32), self.norm1.eps, cd)
        qkv = ops.linear(y.reshape(B * L, C), self._w(self.attn.qkv.weight), None).reshape(B, H, W, 3 * C)
        bq = None if self.attn.qkv.bias is None else self._w(self.attn.qkv.bias, f32)
        o, attn = ops.window_attention_fwd(qkv, bq, self._w(self.attn.relative_position_bias_table, f32), self.num_heads,
                                           self.window_size, self.shift_size, return_attn=self.return_attention)
        a = ops.linear(o.reshape(B * L, C), self._w(self.attn.proj.weight), self._w(self.attn.proj.bias, f32))
        x, z = ops.add_layernorm(x, a.reshape(B, L, C), self._w(self.norm2.weight, f32), self._w(self.norm2.bias, f32),
                                 self.norm2.eps, cd)
        z = ops.linear(z.reshape(B * L, C), self._w(self.mlp.fc1.weight), self._w(self.mlp.fc1.bias, f32), act="gelu")
        z = ops.linear(z, self._w(self.mlp.fc2.weight), self._w(self.mlp.fc2.bias, f32))
        return x, z.reshape(B, L, C), attn

    def forward(self, x, hw=None):
        """x [B, H*W, C]; hw = (H, W) of the token grid (default: square)."""
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x, hw)
        L = x.shape[1]
        hw = hw if hw is not None else (int(math.sqrt(L)),) * 2
        x, z, attn = self.forward_pending(x.float().contiguous(), None, hw)
        x, _ = ops.add_layernorm(x, z, None, None, 0.0, self.compute_dtype, want_y=False)
        return x, attn


def run_blocks(blocks, x, hw):
    """A stage's blocks on the no-grad path: every residual add rides in the next LayerNorm, the last one is applied here."""
    x, delta = x.float().contiguous(), None
    for blk in blocks:
        x, delta, _ = blk.forward_pending(x, delta, hw)
    x, _ = ops.add_layernorm(x, delta, None, None, 0.0, blocks[-1].compute_dtype, want_y=False)
    return x


def fused_path(module):
    return not (module.training and torch.is_grad_enabled())


def patch_embed_tokens(proj, norm, x, compute_dtype=torch.float32):
    """Non-overlapping patch convolution (+ LayerNorm) as an unfold + `as_linear_fwd` GEMM in fp32 (K = 3 p^2, zero-padded
    to a multiple of 32) + `as_add_layernorm`: img [B,3,H,W] (H, W multiples of the patch) -> tokens [B, Hp*Wp, C] fp32."""
    B, Cin, H, W = x.shape
    ph, pw = proj.kernel_size
    Hp, Wp = H // ph, W // pw
    K = Cin * ph * pw
    Kp = -(-K // 32) * 32
    cols = x[:, :, :Hp * ph, :Wp * pw].float().reshape(B, Cin, Hp, ph, Wp, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * Hp * Wp, K)
    wgt = proj.weight.detach().float().reshape(proj.out_channels, K)
    if Kp != K:
        cols = F.pad(cols, (0, Kp - K))
        wgt = F.pad(wgt, (0, Kp - K))
    t = ops.linear(cols.contiguous(), wgt.contiguous(), None if proj.bias is None else proj.bias.detach().float())
    t = t.view(B, Hp * Wp, proj.out_channels)
    if norm is not None:
        _, t = ops.add_layernorm(t, None, norm.weight.detach().float(), norm.bias.detach().float(), norm.eps,
                                 torch.float32, want_x=False)
    return t, Hp, Wp


def merge_tokens(x, H, W, norm, reduction, compute_dtype):
    """Patch merging on the no-grad path: 2x2 gather (one copy), LayerNorm -> compute dtype, bias-free GEMM, fp32 out."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).reshape(B, -1, 4 * C)
    _, y = ops.add_layernorm(x.float().contiguous(), None, norm.weight.detach().float(), norm.bias.detach().float(), norm.eps,
                             compute_dtype, want_x=False)
    wgt = reduction.weight.detach().to(compute_dtype).contiguous()
    return ops.linear(y.reshape(-1, 4 * C), wgt, None).reshape(B, -1, 2 * C).float()


# ---- the whole backbone around the block (BASELINE config 5: Swin-B = embed 128, depths 2/2/18/2, heads 4/8/16/32) ------
class PatchMerging(nn.Module):
    """models/swin_transformer.py:337-377: 2x2 neighbourhood -> 4C, LayerNorm, Linear(4C -> 2C, no bias)."""

    def __init__(self, input_resolution, dim, compute_dtype=torch.bfloat16):
        super().__init__()
        self.input_resolution, self.dim, self.compute_dtype = tuple(input_resolution), dim, compute_dtype
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim, eps=1e-6)

    def forward(self, x):
        B, L, C = x.shape
        H = W = int(math.sqrt(L))
        if fused_path(self) and x.is_cuda:
            return merge_tokens(x, H, W, self.norm, self.reduction, self.compute_dtype)
        x = x.view(B, H, W, C)
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).reshape(B, -1, 4 * C)
        return self.reduction(self.norm(x))


class BasicLayer(nn.Module):
    """models/swin_transformer.py:393-460: `depth` blocks alternating shift 0 / window_size // 2, then the merge."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True,
                 downsample=False, compute_dtype=torch.bfloat16, return_attention=False):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 mlp_ratio, qkv_bias, compute_dtype=compute_dtype, return_attention=return_attention)
            for i in range(depth)])
        for blk in self.blocks:                               # the reference's norm_layer = LayerNorm(eps=1e-6)
            blk.norm1.eps = blk.norm2.eps = 1e-6
        self.downsample = PatchMerging(input_resolution, dim, compute_dtype) if downsample else None

    def forward(self, x):
        if fused_path(self) and x.is_cuda:
            L = x.shape[1]
            x = run_blocks(self.blocks, x, (int(math.sqrt(L)),) * 2)
        else:
            for blk in self.blocks:
                x, _ = blk(x)
        return self.downsample(x) if self.downsample is not None else x

    def forward_with_features(self, x):
        feats = []
        for blk in self.blocks:
            x, _ = blk(x)
            feats.append(x)
        return (self.downsample(x) if self.downsample is not None else x), feats


class SwinPatchEmbed(nn.Module):
    """models/swin_transformer.py:474-509: conv patch x patch / patch, optional LayerNorm over channels, NCHW out."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, patch_norm=True):
        super().__init__()
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6) if patch_norm else None

    def tokens(self, x):
        """No-grad path: tokens [B, Hp*Wp, C] fp32 without the NCHW round trip."""
        return patch_embed_tokens(self.proj, self.norm, x)[0]

    def forward(self, x):
        if fused_path(self) and x.is_cuda:
            t, H, W = patch_embed_tokens(self.proj, self.norm, x)
            return t.transpose(1, 2).reshape(x.shape[0], -1, H, W)
        x = self.proj(x)
        B, C, H, W = x.shape
        t = x.flatten(2).transpose(1, 2)
        if self.norm is not None:
            t = self.norm(t)
        return t.transpose(1, 2).reshape(B, C, H, W)


class SwinTransformer(nn.Module):
    """models/swin_transformer.py:519-645 with the same constructor arguments, module names and state-dict keys
    (`patch_embed.*`, `layers.i.blocks.j.*`, `layers.i.downsample.*`, `norm.*`, `head.*`, optional
    `absolute_pos_embed`).  Every window attention runs on `as_window_attn_fwd` / `_bwd` through the block above; the
    patch embedding / merging / norms are tensor-op glue.  `forward` returns what the reference returns (pooled feature,
    or [pooled | tokens] with return_all_tokens); `forward_stages` additionally returns the tokens after every stage."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., ape=False, patch_norm=True, return_all_tokens=False,
                 compute_dtype=torch.bfloat16, **kwargs):
        super().__init__()
        self.num_classes, self.depths, self.num_layers = num_classes, list(depths), len(depths)
        self.embed_dim, self.ape, self.return_all_tokens = embed_dim, ape, return_all_tokens
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = SwinPatchEmbed(img_size, patch_size, in_chans, embed_dim, patch_norm)
        res = self.patch_embed.patches_resolution
        self.patches_resolution = res
        if ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), (res[0] // 2 ** i, res[1] // 2 ** i), depths[i], num_heads[i], window_size,
                       mlp_ratio, qkv_bias, downsample=i < self.num_layers - 1, compute_dtype=compute_dtype)
            for i in range(self.num_layers)])
        self.norm = nn.LayerNorm(self.num_features, eps=1e-6)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
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

    def forward_stages(self, x):
        if fused_path(self) and x.is_cuda:
            x = self.patch_embed.tokens(x)
        else:
            x = self.patch_embed(x).flatten(2).transpose(1, 2)
        if self.ape:
            x = x + self.absolute_pos_embed
        stages = []
        for layer in self.layers:
            x = layer(x)
            stages.append(x)
        return self.norm(x), stages

    def forward(self, x, return_all_tokens=None):
        x_region, _ = self.forward_stages(x)
        pooled = x_region.mean(dim=1)                          # AdaptiveAvgPool1d(1) over the tokens
        rat = self.return_all_tokens if return_all_tokens is None else return_all_tokens
        return torch.cat([pooled.unsqueeze(1), x_region], dim=1) if rat else pooled
