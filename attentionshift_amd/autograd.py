"""torch.autograd bridge for the trainable part of the hot path: Attention.forward with the HIP forward
(as_attn_fwd) and the HIP backward (as_attn_bwd).

Reference: models/vision_transformer.py:74-86 is an ordinary nn.Module differentiated by autograd, with the
[B,h,N,N] softmax saved for backward (or recomputed per block under `use_checkpoint`,
visual_transformer_det.py:232-236).  Here the saved tensors are q, k, v^T, o and the row log-sum-exp; the backward
recomputes softmax tiles on chip (csrc/sdpa_bwd.hip)."""
import os

import torch

from . import ops

_UNFUSED_MLP = bool(os.environ.get("AS_MLP_UNFUSED"))           # A/B switch: fc1 -> F.gelu -> fc2 as three autograd nodes
_LIBRARY_LINEAR = bool(os.environ.get("AS_LINEAR_LIBRARY"))     # A/B switch: leave every nn.Linear to the library GEMMs


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


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the HIP GEMM (as_linear_fwd) with the backward as_linear_bwd computes: dx on the same kernel, dW
    as a split-K product over the rows (the library runs these few-tile, deep-contraction shapes on a handful of
    workgroups), db by fixed-order column sums.  x bf16 [..., K]; W bf16 or fp32 master [Nout, K] (cast here, dW comes
    back in W's dtype); b fp32 / bf16 / None.  models/vision_transformer.py:47-59 (Mlp) and the decoder blocks of the
    MAE heads (mae_bbox_head_rec.py:148-168) under autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        wb = weight if weight.dtype == torch.bfloat16 else weight.to(torch.bfloat16)
        wb = wb.contiguous()
        out = ops.linear(x2, wb, None if bias is None else bias.float())
        ctx.save_for_backward(x2, wb)
        ctx.meta = (x.shape, weight.dtype, None if bias is None else bias.dtype)
        return out.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wb = ctx.saved_tensors
        shape, w_dtype, b_dtype = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        dy2 = (dy2 if dy2.dtype == torch.bfloat16 else dy2.to(torch.bfloat16)).contiguous()
        need_db = b_dtype is not None and ctx.needs_input_grad[2]
        dx, dw, db = ops.linear_bwd(x2, wb, dy2, ctx.needs_input_grad[0], ctx.needs_input_grad[1], need_db, dw_dtype=w_dtype)
        return (None if dx is None else dx.view(shape), dw, None if db is None else db.to(b_dtype))


class MlpFn(torch.autograd.Function):
    """fc2(GELU(fc1(x))) as ONE autograd node (models/vision_transformer.py:47-59): the GELU runs in fc1's epilogue
    (as_linear_gelu_fwd writes the bf16 pre-activation and its GELU), its derivative in the epilogue of fc2's input-gradient
    GEMM (as_linear_bwd_dgelu) -- no activation pass in either direction (ATen's GeluBackward alone moved 155 MB per
    ViT-B layer).  Same rounding points as LinearFn -> F.gelu -> LinearFn: h, GELU(h) and every gradient are bf16 tensors."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        cast = lambda w: (w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)).contiguous()    # noqa: E731
        w1b, w2b = cast(w1), cast(w2)
        a, pre = ops.linear_gelu(x2, w1b, None if b1 is None else b1.float())
        y = ops.linear(a, w2b, None if b2 is None else b2.float())
        ctx.save_for_backward(x2, w1b, w2b, pre, a)
        ctx.meta = (x.shape, w1.dtype, None if b1 is None else b1.dtype, w2.dtype, None if b2 is None else b2.dtype)
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w1b, w2b, pre, a = ctx.saved_tensors
        shape, w1_dtype, b1_dtype, w2_dtype, b2_dtype = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1])
        dy2 = (dy2 if dy2.dtype == torch.bfloat16 else dy2.to(torch.bfloat16)).contiguous()
        need = ctx.needs_input_grad
        need_b1, need_b2 = b1_dtype is not None and need[2], b2_dtype is not None and need[4]
        first = need[0] or need[1] or need_b1               # does anything upstream of fc2's input want a gradient?
        if not (first or need[3] or need_b2):
            return None, None, None, None, None
        # (a frozen fc1 under a frozen input -- head-only fine-tuning -- needs neither d(pre) nor the second call)
        dpre, dw2, db2 = ops.linear_bwd(a, w2b, dy2, first, need[3], need_b2, dw_dtype=w2_dtype, gelu_pre=pre if first else None)
        dx = dw1 = db1 = None
        if first:
            dx, dw1, db1 = ops.linear_bwd(x2, w1b, dpre, need[0], need[1], need_b1, dw_dtype=w1_dtype)
        return (None if dx is None else dx.view(shape), dw1, None if db1 is None else db1.to(b1_dtype), dw2,
                None if db2 is None else db2.to(b2_dtype))


def mlp(x, w1, b1, w2, b2):
    """The transformer MLP under autograd on the HIP kernels in bf16 (MlpFn)."""
    return MlpFn.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), w1, b1, w2, b2)


def mlp_applies(x, w1, w2):
    return linear_applies(x, w1) and w2.shape[0] % 32 == 0 and not _UNFUSED_MLP


def linear(x, weight, bias=None):
    """nn.Linear under autograd on the HIP kernels, computing in bf16 (what autocast / apex O1 does to F.linear)."""
    return LinearFn.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), weight, bias)


def linear_shapes_ok(x, weight):
    """The HIP linear takes these operands (device tensors, non-empty, feature sizes in rows of 32) -- for callers that
    have already decided to compute in bf16 (the backbones' compute_dtype)."""
    return not (_LIBRARY_LINEAR or not x.is_cuda or x.numel() == 0 or weight.shape[0] % 32 or weight.shape[1] % 32)


def linear_in(x, weight, bias, compute_dtype):
    """nn.Linear of a trainable path whose compute dtype is `compute_dtype`: the HIP forward / backward kernels (LinearFn:
    dW as a split-K product, db as fixed-order column sums, fp32 master weights cast inside) when the dtype is bf16 and the
    sizes fit, the library GEMM otherwise."""
    if compute_dtype == torch.bfloat16 and linear_shapes_ok(x, weight):
        return linear(x, weight, bias)
    return torch.nn.functional.linear(x.to(compute_dtype), weight.to(compute_dtype),
                                      None if bias is None else bias.to(compute_dtype))


def linear_applies(x, weight):
    """Whether `linear` can stand in for F.linear here: device tensors, a bf16 autocast region (or bf16 operands
    already) and sizes the kernels take (rows of 32)."""
    if _LIBRARY_LINEAR or not x.is_cuda or x.numel() == 0 or weight.shape[0] % 32 or weight.shape[1] % 32:
        return False                                 # (an empty batch -- no positive RoIs -- stays on the tensor ops)
    if x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16:
        return True
    return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


class ParamCastFn(torch.autograd.Function):
    """Compute-dtype copies of MANY parameters in one multi-tensor pass, differentiable: what `p.to(bf16)` per parameter
    and per forward does under apex O1 / autocast (the reference trains that way, run_train.py), without its one cast
    launch per parameter in the forward and one per gradient in the backward (ViT-B: 72 + 72 launches per step).
    forward(dtype, *params) -> tuple of `dtype` copies (views of one flat buffer); backward: the copies' gradients are
    converted back to the parameters' dtype by one multi-tensor copy."""

    @staticmethod
    def forward(ctx, dtype, *params):
        ctx.meta = [(p.dtype, p.shape) for p in params]
        ctx.set_materialize_grads(False)             # an unused copy costs nothing in backward (None is skipped below)
        flat = torch.empty(sum(p.numel() for p in params), device=params[0].device, dtype=dtype)
        outs = [v.view(p.shape) for v, p in zip(flat.split([p.numel() for p in params]), params)]
        torch._foreach_copy_(outs, [p.detach() for p in params])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        idx = [i for i, g in enumerate(grads) if g is not None]
        res = [None] * len(grads)
        if idx:
            outs = [torch.empty(ctx.meta[i][1], device=grads[i].device, dtype=ctx.meta[i][0]) for i in idx]
            torch._foreach_copy_(outs, [grads[i] for i in idx])
            for o, i in zip(outs, idx):
                res[i] = o
        return (None, *res)


def cast_params(params, dtype):
    """{id(p): copy of p in `dtype`} for a list of parameters (one fused cast; differentiable when grad is enabled)."""
    params = list(params)
    if not params:
        return {}
    outs = ParamCastFn.apply(dtype, *params)
    return {id(p): o for p, o in zip(params, outs)}


class AddLayerNormFn(torch.autograd.Function):
    """(x_out, y) = (x + delta, LayerNorm(x + delta) * gamma + beta) in one pass forward (as_add_layernorm) and one pass
    backward (as_add_layernorm_bwd): the residual glue of Block.forward (models/vision_transformer.py:109-124).
    x fp32 [B,N,D] residual stream, delta `out_dtype` | None (the previous sub-layer's output), y in `out_dtype`.
    delta_scale fp32 [B] | None: x_out = x + delta_scale[b] * delta -- the DropPath around the sub-layer
    (models/vision_transformer.py:114-118) folded into the add, in fp32, at no extra pass; not differentiated."""

    @staticmethod
    def forward(ctx, x, delta, gamma, beta, eps, out_dtype, delta_scale=None):
        x = x.contiguous()
        delta = None if delta is None else delta.contiguous()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        sc = None if (delta_scale is None or delta is None) else delta_scale.detach().float().contiguous()
        x_out, y = ops.add_layernorm(x, delta, g, b, eps, out_dtype, delta_scale=sc)
        ctx.set_materialize_grads(False)                 # an unused output arrives as None, not as a zero tensor
        ctx.save_for_backward(x_out, g, sc)
        ctx.cfg = (float(eps), out_dtype, delta is not None, gamma.dtype)
        return x_out, y

    @staticmethod
    def backward(ctx, dx_out, dy):
        x_out, g, sc = ctx.saved_tensors
        eps, out_dtype, has_delta, pdt = ctx.cfg
        if dx_out is None and dy is None:
            return None, None, None, None, None, None, None
        dx_out = None if dx_out is None else dx_out.contiguous()
        dy = None if dy is None else dy.to(out_dtype).contiguous()
        dx, dd, dg, db = ops.add_layernorm_bwd(x_out, dy, dx_out, g, eps, out_dtype, want_dx=ctx.needs_input_grad[0],
                                               want_ddelta=has_delta and ctx.needs_input_grad[1],
                                               want_affine=ctx.needs_input_grad[2] or ctx.needs_input_grad[3], delta_scale=sc)
        dg = None if dg is None or not ctx.needs_input_grad[2] else dg.to(pdt)
        db = None if db is None or not ctx.needs_input_grad[3] else db.to(pdt)
        return dx, dd, dg, db, None, None, None


def add_layernorm(x, delta, gamma, beta, eps, out_dtype, delta_scale=None):
    return AddLayerNormFn.apply(x, delta, gamma, beta, eps, out_dtype, delta_scale)


class DecoderBlockFn(torch.autograd.Function):
    """One pre-LN transformer block of the MAE-decoder heads (mae_bbox_head_rec.py:148-168 = the Block of
    models/vision_transformer.py:109-124 at embed 256 / 8 heads of 32) as ONE autograd node:

        x1 = x + delta;  y = LN1(x1);  a = proj(attn(qkv(y)));  x2 = x1 + a;  z = LN2(x2);  out_delta = fc2(gelu(fc1(z)))

    forward(x fp32 [R,N,C], delta bf16 | None, 12 parameters, eps1, eps2, num_heads) -> (x2 fp32, out_delta bf16).  The same
    kernels as the per-op bridges (AddLayerNormFn, LinearFn, MlpFn, SmallAttnFn) in the same order, so values and gradients are
    theirs bit for bit; what goes away is seven `Function.apply` calls and autograd nodes per block and direction -- the RoI
    head's loss phase was host-bound on them (6.2 ms of host for ~3 ms of device work per training step)."""

    @staticmethod
    def forward(ctx, x, delta, g1, b1, wq, bq, wp, bp, g2, b2, w1, bf1, w2, bf2, eps1, eps2, heads):
        bf = torch.bfloat16
        R, N, C = x.shape
        f = lambda t: None if t is None else t.detach().float().contiguous()        # noqa: E731
        c = lambda t: t.detach().to(bf).contiguous()                              # noqa: E731
        g1f, g2f = f(g1), f(g2)
        wqb, wpb, w1b, w2b = c(wq), c(wp), c(w1), c(w2)
        x1, y = ops.add_layernorm(x.contiguous(), None if delta is None else delta.contiguous(), g1f, f(b1), eps1, bf)
        qkv = ops.linear(y.reshape(R * N, C), wqb, f(bq)).reshape(R, N, 3, heads, C // heads)
        o, lse = ops.small_attention_fwd(qkv)
        a = ops.linear(o.reshape(R * N, C), wpb, f(bp)).reshape(R, N, C)
        x2, z = ops.add_layernorm(x1, a, g2f, f(b2), eps2, bf)
        if _UNFUSED_MLP:
            h = ops.linear(z.reshape(R * N, C), w1b, f(bf1))
            hg = torch.nn.functional.gelu(h)
        else:
            hg, h = ops.linear_gelu(z.reshape(R * N, C), w1b, f(bf1))        # GELU in fc1's epilogue (MlpFn's kernels)
        d = ops.linear(hg, w2b, f(bf2)).reshape(R, N, C)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x1, y, qkv, o, lse, x2, z, h, hg, wqb, wpb, w1b, w2b, g1f, g2f)
        ctx.cfg = (float(eps1), float(eps2), delta is not None, [None if t is None else t.dtype for t in
                                                                  (g1, b1, wq, bq, wp, bp, g2, b2, w1, bf1, w2, bf2)])
        return x2, d

    @staticmethod
    def backward(ctx, dx2, dd):
        x1, y, qkv, o, lse, x2, z, h, hg, wqb, wpb, w1b, w2b, g1f, g2f = ctx.saved_tensors
        eps1, eps2, has_delta, dts = ctx.cfg
        bf = torch.bfloat16
        R, N, C = x1.shape
        M = R * N
        need = ctx.needs_input_grad
        dz = dw2 = db2 = dw1 = db1 = None
        if dd is not None:
            dd2 = dd.to(bf).reshape(M, C).contiguous()
            if _UNFUSED_MLP:
                dhg, dw2, db2 = ops.linear_bwd(hg, w2b, dd2, True, need[12], need[13] and dts[11] is not None, dw_dtype=dts[10])
                dh = torch.ops.aten.gelu_backward(dhg, h)
            else:                                       # GELU' in the epilogue of fc2's input-gradient GEMM
                dh, dw2, db2 = ops.linear_bwd(hg, w2b, dd2, True, need[12], need[13] and dts[11] is not None, dw_dtype=dts[10],
                                              gelu_pre=h)
            dz, dw1, db1 = ops.linear_bwd(z.reshape(M, C), w1b, dh, True, need[10], need[11] and dts[9] is not None, dw_dtype=dts[8])
            dz = dz.reshape(R, N, C)
        if dz is None and dx2 is None:
            return (None,) * 17
        dx1, da, dg2, dbt2 = ops.add_layernorm_bwd(x2, dz, None if dx2 is None else dx2.contiguous(), g2f, eps2, bf,
                                                   want_dx=True, want_ddelta=True, want_affine=need[8] or need[9])
        do, dwp, dbp = ops.linear_bwd(o.reshape(M, C), wpb, da.reshape(M, C), True, need[6], need[7] and dts[5] is not None,
                                      dw_dtype=dts[4])
        dqkv = ops.small_attention_bwd(qkv, o, do.reshape(R, N, C), lse)
        dy, dwq, dbq = ops.linear_bwd(y.reshape(M, C), wqb, dqkv.reshape(M, 3 * C), True, need[4], need[5] and dts[3] is not None,
                                      dw_dtype=dts[2])
        dx, ddelta, dg1, dbt1 = ops.add_layernorm_bwd(x1, dy.reshape(R, N, C), dx1, g1f, eps1, bf, want_dx=need[0],
                                                      want_ddelta=has_delta and need[1], want_affine=need[2] or need[3])
        cast = lambda t, i: None if (t is None or not need[i + 2]) else t.to(dts[i])     # noqa: E731
        return (dx, ddelta, cast(dg1, 0), cast(dbt1, 1), dwq if need[4] else None, cast(dbq, 3), dwp if need[6] else None,
                cast(dbp, 5), cast(dg2, 6), cast(dbt2, 7), dw1 if need[10] else None, cast(db1, 9), dw2 if need[12] else None,
                cast(db2, 11), None, None, None)


def decoder_block(x, delta, blk):
    """DecoderBlockFn on a block module with .norm1 / .attn.qkv / .attn.proj / .norm2 / .mlp.fc1 / .mlp.fc2 (mae_heads.DecoderBlock)."""
    a, m = blk.attn, blk.mlp
    return DecoderBlockFn.apply(x, delta, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                                blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                blk.norm1.eps, blk.norm2.eps, a.num_heads)
