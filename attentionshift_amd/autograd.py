"""torch.autograd bridge for the trainable part of the hot path: Attention.forward with the HIP forward
(as_attn_fwd) and the HIP backward (as_attn_bwd).

Reference: models/vision_transformer.py:74-86 is an ordinary nn.Module differentiated by autograd, with the
[B,h,N,N] softmax saved for backward (or recomputed per block under `use_checkpoint`,
visual_transformer_det.py:232-236).  Here the saved tensors are q, k, v^T, o and the row log-sum-exp; the backward
recomputes softmax tiles on chip (csrc/sdpa_bwd.hip)."""
import torch

from . import ops


class AttentionFn(torch.autograd.Function):
    """out = proj(softmax(q k^T / sqrt(d)) v),  (q,k,v) = split(x Wqkv^T + bqkv).

    forward(x [B,N,D], w_qkv [3D,D], b_qkv fp32 [3D] | None, w_proj [D,D], b_proj fp32 [D] | None, num_heads, sink)
    `sink` (a list or None) receives the layer's AttnLayerState so the roll-out can recompute attention rows."""

    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, w_proj, b_proj, num_heads, sink):
        x = x.contiguous()
        w_qkv, w_proj = w_qkv.contiguous(), w_proj.contiguous()
        out, st = ops.attention_fwd(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, keep_state=True, keep_o=True)
        ctx.save_for_backward(x, w_qkv, w_proj)
        ctx.state = st
        ctx.has_bias = (b_qkv is not None, b_proj is not None)
        if sink is not None:
            sink.append(st)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_qkv, w_proj = ctx.saved_tensors
        dx, dwqkv, dbqkv, dwproj, dbproj = ops.attention_bwd(x, w_qkv, w_proj, dout.contiguous(), ctx.state,
                                                             want_bias=ctx.has_bias)
        return dx, dwqkv, dbqkv, dwproj, dbproj, None, None


def attention(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, sink=None):
    return AttentionFn.apply(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, sink)
