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
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(self.input_resolution) <= self.window_size:          # :211-214
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.compute_dtype = compute_dtype
        self.return_attention = return_attention

    def _forward_train(self, x):
        """Autograd path (train() + grad enabled): the window attention core runs on the HIP forward / backward kernels
        (WindowAttnFn); LayerNorm, the Linear layers and GELU are torch ops (library GEMMs)."""
        B, L, C = x.shape
        H = W = int(math.sqrt(L))
        cd = self.compute_dtype
        y = F.layer_norm(x, (C,), self.norm1.weight, self.norm1.bias, self.norm1.eps).to(cd)
        qkv = F.linear(y, self.attn.qkv.weight.to(cd)).reshape(B, H, W, 3 * C)
        bq = None if self.attn.qkv.bias is None else self.attn.qkv.bias.float()
        o = WindowAttnFn.apply(qkv, bq, self.attn.relative_position_bias_table.float(), self.num_heads, self.window_size,
                               self.shift_size)
        y = F.linear(o.reshape(B, L, C), self.attn.proj.weight.to(cd), self.attn.proj.bias.to(cd))
        x = x + y.float()
        z = F.layer_norm(x, (C,), self.norm2.weight, self.norm2.bias, self.norm2.eps).to(cd)
        z = F.linear(F.gelu(F.linear(z, self.mlp.fc1.weight.to(cd), self.mlp.fc1.bias.to(cd))),
                     self.mlp.fc2.weight.to(cd), self.mlp.fc2.bias.to(cd))
        return x + z.float(), None

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x)
        B, L, C = x.shape
        H = W = int(math.sqrt(L))
        cd = self.compute_dtype
        w = lambda p: p.detach().to(cd).contiguous()
        y = F.layer_norm(x, (C,), self.norm1.weight, self.norm1.bias, self.norm1.eps).to(cd)
        qkv = ops.linear(y.reshape(B * L, C).contiguous(), w(self.attn.qkv.weight), None).reshape(B, H, W, 3 * C)
        bq = None if self.attn.qkv.bias is None else self.attn.qkv.bias.detach().float().contiguous()
        o, attn = ops.window_attention_fwd(qkv, bq, self.attn.relative_position_bias_table.detach().float().contiguous(),
                                           self.num_heads, self.window_size, self.shift_size,
                                           return_attn=self.return_attention)
        y = ops.linear(o.reshape(B * L, C), w(self.attn.proj.weight), self.attn.proj.bias.detach().float())
        x = x + y.reshape(B, L, C).float()
        z = F.layer_norm(x, (C,), self.norm2.weight, self.norm2.bias, self.norm2.eps).to(cd)
        z = ops.linear(z.reshape(B * L, C).contiguous(), w(self.mlp.fc1.weight), self.mlp.fc1.bias.detach().float(), act="gelu")
        z = ops.linear(z, w(self.mlp.fc2.weight), self.mlp.fc2.bias.detach().float())
        return x + z.reshape(B, L, C).float(), attn
