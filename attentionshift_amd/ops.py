"""Torch-tensor wrappers over the C ABI (include/attnshift.h).

PyTorch is plumbing here: device memory, the current HIP stream, autograd bookkeeping.  Every
function validates device/dtype/contiguity, passes raw pointers and raises on any non-zero return.
"""
import ctypes
import os

import threading

import torch

from . import _lib
from ._lib import AS_BF16, AS_F32, AttnShiftError

HEAD_DIM = 64


def _dt(t):
    if t.dtype == torch.float32:
        return AS_F32
    if t.dtype == torch.bfloat16:
        return AS_BF16
    raise AttnShiftError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


def _chk(*tensors, dtype=None):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AttnShiftError("hot-path ops need device (HBM) tensors; there is no CPU fallback")
        if not t.is_contiguous():
            raise AttnShiftError("hot-path ops need contiguous row-major tensors")
        if dtype is not None and t.dtype != dtype:
            raise AttnShiftError(f"expected {dtype}, got {t.dtype}")


def _p(t):
    return 0 if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_DEV = [None]


def _stream():
    """raw hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream
    object through three Python layers (~10 us, ~70 times per step in a host-bound phase); the raw accessor is ~0.3 us."""
    if _RAW_STREAM is None:
        return torch.cuda.current_stream().cuda_stream
    d = _DEV[0]
    if d is None:
        d = _DEV[0] = torch.cuda.current_device()      # one process drives one GPU (one rank per GPU)
    return _RAW_STREAM(d)


# ------------------------------------------------------------------------------------------------
# optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
# ------------------------------------------------------------------------------------------------
_TIMED = None      # None = off; else {name: [(start_event, stop_event), ...]}


def enable_timing(names):
    global _TIMED
    _TIMED = {n: [] for n in names}


def disable_timing():
    global _TIMED
    _TIMED = None


def collect_timing():
    """-> {name: (launches, mean_ms)}; call after a device synchronize."""
    out = {}
    for n, evs in (_TIMED or {}).items():
        if n.endswith(":bytes"):
            out[n] = (len(evs), float(sum(evs)) / max(len(evs), 1))
        elif evs:
            ms = [a.elapsed_time(b) for a, b in evs]
            out[n] = (len(ms), sum(ms) / len(ms))
    return out


class _timed:
    def __init__(self, name):
        self.on = _TIMED is not None and name in _TIMED
        self.name = name

    def __enter__(self):
        if self.on:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())

    def __exit__(self, *exc):
        if self.on:
            self.b.record(torch.cuda.current_stream())
            _TIMED[self.name].append((self.a, self.b))


def npad(n):
    return _lib.load().as_npad(int(n))


# ------------------------------------------------------------------------------------------------
# Part A
# ------------------------------------------------------------------------------------------------
_SK_BYTES = {}
_SK_WS = {}


def _sk_bytes(lib, M, Nout, K):
    # (the library re-reads AS_GEMM_SK on every call -- tests switch it -- so the switch is part of the key: ADVICE r05)
    key = (M, Nout, K, os.environ.get("AS_GEMM_SK"))
    v = _SK_BYTES.get(key)
    if v is None:
        v = _SK_BYTES[key] = int(lib.as_linear_sk_workspace_bytes(M, Nout, K))
    return v


def _sk_workspace(device, nbytes):
    """One stream-K workspace per (device, stream): calls on a stream are ordered, so consecutive GEMMs share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SK_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _SK_WS[key] = torch.empty(nbytes, device=device, dtype=torch.uint8)
    return ws


def linear(x, weight, bias=None, act="none"):
    """act(x @ weight.T + bias); x [..., K] in fp32/bf16, bias fp32."""
    lib = _lib.load()
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    _chk(x2, weight)
    if bias is not None:
        _chk(bias, dtype=torch.float32)
    if weight.dtype != x.dtype or weight.shape[1] != K:
        raise AttnShiftError("linear: weight must be [Nout, K] in the dtype of x")
    out = torch.empty(x2.shape[0], weight.shape[0], device=x.device, dtype=x.dtype)
    code = {"none": 0, "gelu": 1, "relu": 4}[act]
    M, Nout = x2.shape[0], weight.shape[0]
    need = _sk_bytes(lib, M, Nout, K) if x.dtype == torch.bfloat16 else 0
    if need:                                               # stream-K schedule (as_linear_sk_fwd) with its fp32 partial tiles
        ws = _sk_workspace(x.device, need)
        _lib.check(lib.as_linear_sk_fwd(_p(x2), _p(weight), _p(bias), _p(out), M, Nout, K, _dt(x), code, _p(ws), ws.numel(),
                                        _stream()), "as_linear_sk_fwd")
    else:
        _lib.check(lib.as_linear_fwd(_p(x2), _p(weight), _p(bias), _p(out), M, Nout, K, _dt(x), code, _stream()), "as_linear_fwd")
    return out.reshape(*x.shape[:-1], Nout)


def linear_small(x, weight, bias=None, act="none", out=None):
    """fp32 nn.Linear for a few hundred rows (the point head, as_linear_small_fwd): x [M, K] fp32 whose rows may be a column
    slice of a wider tensor (stride(0) >= K, stride(1) == 1), weight [Nout, K] fp32 contiguous, bias fp32 | None;
    act none | gelu | relu | sigmoid.  `out` [M, Nout] (row stride >= Nout) is written in place when given."""
    lib = _lib.load()
    if x.dim() != 2 or x.dtype != torch.float32 or weight.dtype != torch.float32 or x.stride(1) != 1 or not weight.is_contiguous():
        raise AttnShiftError("linear_small: x [M, K] fp32 with unit column stride, weight [Nout, K] fp32 contiguous")
    M, K = x.shape
    Nout = weight.shape[0]
    if weight.shape[1] != K:
        raise AttnShiftError("linear_small: weight must be [Nout, K]")
    if bias is not None:
        _chk(bias, dtype=torch.float32)
    if out is None:
        out = torch.empty(M, Nout, device=x.device, dtype=torch.float32)
    elif out.shape != (M, Nout) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise AttnShiftError("linear_small: out must be [M, Nout] fp32 with unit column stride")
    code = {"none": 0, "gelu": 1, "relu": 4, "sigmoid": 5}[act]
    _lib.check(lib.as_linear_small_fwd(_p(x), x.stride(0), _p(weight), _p(bias), _p(out), out.stride(0), M, Nout, K, code, _stream()),
               "as_linear_small_fwd")
    return out


def linear_small_applies(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and 0 < x.shape[0] <= 1024
            and x.shape[1] >= 64 and x.shape[1] % 16 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and weight.is_contiguous()
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)


def linear_gelu(x, weight, bias=None):
    """The Mlp's fc1 under autograd: (GELU(h), h) with h = x @ weight.T + bias rounded to bf16, both written by the GEMM's
    epilogue (as_linear_gelu_fwd).  x [M,K] bf16, weight [Nout,K] bf16, bias fp32 | None."""
    lib = _lib.load()
    _chk(x, weight, dtype=torch.bfloat16)
    if bias is not None:
        _chk(bias, dtype=torch.float32)
    M, K = x.shape
    Nout = weight.shape[0]
    if weight.shape[1] != K:
        raise AttnShiftError("linear_gelu: weight must be [Nout, K]")
    out = torch.empty(M, Nout, device=x.device, dtype=torch.bfloat16)
    pre = torch.empty_like(out)
    _lib.check(lib.as_linear_gelu_fwd(_p(x), _p(weight), _p(bias), _p(out), _p(pre), M, Nout, K, AS_BF16, _stream()),
               "as_linear_gelu_fwd")
    return out, pre


def linear_gelu_applies(x, weight):
    """Sizes as_linear_gelu_fwd / as_linear_bwd_dgelu take (rows of 32 on both feature axes)."""
    return x.is_cuda and x.numel() > 0 and weight.shape[0] % 32 == 0 and weight.shape[1] % 32 == 0


def linear_splitk(x, weight, out_dtype=None):
    """x @ weight.T for bf16 x [M,K], weight [Nout,K] with the contraction split over workgroups (fixed-order fp32
    partials): the shape of a weight gradient, dW = dy^T x with the tokens as K.  out_dtype: bf16 (default) or fp32."""
    lib = _lib.load()
    _chk(x, weight, dtype=torch.bfloat16)
    M, K = x.shape
    Nout = weight.shape[0]
    if weight.shape[1] != K:
        raise AttnShiftError("linear_splitk: weight must be [Nout, K]")
    out_dtype = out_dtype or torch.bfloat16
    out = torch.empty(M, Nout, device=x.device, dtype=out_dtype)
    nbytes = int(lib.as_linear_splitk_workspace_bytes(M, Nout, K))
    ws = torch.empty(max(nbytes, 4) // 4, device=x.device, dtype=torch.float32)
    _lib.check(lib.as_linear_splitk_fwd(_p(x), _p(weight), _p(out), M, Nout, K, AS_BF16, 1 if out_dtype == torch.float32 else 0,
                                        _p(ws), nbytes, _stream()), "as_linear_splitk_fwd")
    return out


_LINEAR_BWD_WS = {}


def linear_bwd(x, weight, dy, need_dx=True, need_dw=True, need_db=True, dw_dtype=None, gelu_pre=None):
    """Backward of y = x @ weight.T + b for bf16 x [M,K], weight [Nout,K], dy [M,Nout] -> (dx bf16 | None, dW | None,
    db fp32 | None); dW in `dw_dtype` (bf16 default, fp32 for fp32 master weights).  as_linear_bwd.
    gelu_pre [M,K] bf16: x was GELU(gelu_pre); dx then comes back multiplied by GELU'(gelu_pre) -- the gradient of the
    pre-activation (as_linear_bwd_dgelu)."""
    lib = _lib.load()
    _chk(x, weight, dy, dtype=torch.bfloat16)
    M, K = x.shape
    Nout = weight.shape[0]
    if weight.shape[1] != K or tuple(dy.shape) != (M, Nout):
        raise AttnShiftError("linear_bwd: shapes must be x [M,K], weight [Nout,K], dy [M,Nout]")
    dw_dtype = dw_dtype or torch.bfloat16
    dx = torch.empty(M, K, device=x.device, dtype=torch.bfloat16) if need_dx else None
    dW = torch.empty(Nout, K, device=x.device, dtype=dw_dtype) if need_dw else None
    db = torch.empty(Nout, device=x.device, dtype=torch.float32) if need_db else None
    nbytes = int(lib.as_linear_bwd_workspace_bytes(M, Nout, K))
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _LINEAR_BWD_WS.get(key)                    # one growing scratch buffer per (device, stream): the calls of a
    if ws is None or ws.numel() * 4 < nbytes:       # backward pass are serial on their stream
        ws = _LINEAR_BWD_WS[key] = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    if gelu_pre is not None:
        _chk(gelu_pre, dtype=torch.bfloat16)
        if tuple(gelu_pre.shape) != (M, K) or not need_dx:
            raise AttnShiftError("linear_bwd: gelu_pre must be [M,K] and needs dx")
        _lib.check(lib.as_linear_bwd_dgelu(_p(x), _p(weight), _p(dy), _p(gelu_pre), _p(dx), _p(dW), _p(db), M, Nout, K, AS_BF16,
                                           1 if dw_dtype == torch.float32 else 0, _p(ws), nbytes, _stream()), "as_linear_bwd_dgelu")
        return dx, dW, db
    _lib.check(lib.as_linear_bwd(_p(x), _p(weight), _p(dy), _p(dx), _p(dW), _p(db), M, Nout, K, AS_BF16,
                                 1 if dw_dtype == torch.float32 else 0, _p(ws), nbytes, _stream()), "as_linear_bwd")
    return dx, dW, db


def deconv2x2(x_nhwc, w4, bias4=None, act="none"):
    """ConvTranspose2d(k=2, s=2) on a channels-last bf16 map [B,h,w,cin] -> [B,2h,2w,cout] (one GEMM, scattered epilogue).
    w4 [4*cout, cin] bf16 (row (di*2+dj)*cout + co), bias4 [4*cout] fp32 | None."""
    lib = _lib.load()
    B, h, w, cin = x_nhwc.shape
    cout = w4.shape[0] // 4
    _chk(x_nhwc, w4)
    if bias4 is not None:
        _chk(bias4, dtype=torch.float32)
    out = torch.empty(B, 2 * h, 2 * w, cout, device=x_nhwc.device, dtype=x_nhwc.dtype)
    _lib.check(lib.as_deconv2x2_fwd(_p(x_nhwc), _p(w4), _p(bias4), _p(out), B * h * w, w, cin, cout, _dt(x_nhwc),
                                    1 if act == "gelu" else 0, _stream()), "as_deconv2x2_fwd")
    return out


def _scale_arg(delta_scale, x, delta):
    """(pointer holder, rows per scale) of a per-sample delta scale [B] for x [B, N, D] (DropPath); None -> (None, 1)."""
    if delta_scale is None:
        return None, 1
    if delta is None or x.dim() != 3 or delta_scale.numel() != x.shape[0]:
        raise AttnShiftError("add_layernorm: delta_scale needs delta and x [B, N, D] with one scale per image")
    _chk(delta_scale, dtype=torch.float32)
    return delta_scale, x.shape[1]


def add_layernorm(x, delta, gamma, beta, eps, out_dtype, want_x=True, want_y=True, delta_scale=None, x_out=None):
    """x_new = x + delta (fp32; delta may be None), y = LayerNorm(x_new) in `out_dtype` -- one pass (csrc/layernorm.hip).
    delta_scale fp32 [B] | None: x_new = x + delta_scale[b] * delta (per-sample stochastic depth).
    x_out: the caller's buffer for x_new (fp32, x's shape, contiguous; e.g. a feature tap's slot) instead of a new one.
    Returns (x_new | None, y | None)."""
    lib = _lib.load()
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    _chk(x2, dtype=torch.float32)
    _chk(delta)
    _chk(gamma, beta, dtype=torch.float32)       # (None is allowed when only the add is wanted)
    if delta is not None and delta.dtype != out_dtype:
        raise AttnShiftError("add_layernorm: delta must have the output dtype")
    if x_out is not None:
        if not (want_x and delta is not None) or x_out.shape != x.shape:
            raise AttnShiftError("add_layernorm: x_out needs want_x, a delta and x's shape")
        _chk(x_out, dtype=torch.float32)
        x_out = x_out.view(-1, D)
    else:
        x_out = torch.empty_like(x2) if (want_x and delta is not None) else None
    y = torch.empty(x2.shape, device=x.device, dtype=out_dtype) if want_y else None
    dt = AS_BF16 if out_dtype == torch.bfloat16 else AS_F32
    sc, rps = _scale_arg(delta_scale, x, delta)
    _lib.check(lib.as_add_layernorm_scaled(_p(x2), _p(delta), _p(gamma), _p(beta), float(eps), _p(x_out), _p(y), x2.shape[0], D,
                                           dt, _p(sc), rps, _stream()), "as_add_layernorm")
    xo = x if (delta is None or not want_x) else x_out.reshape(x.shape)
    return (xo if want_x else None), (None if y is None else y.reshape(x.shape))


def assemble_tokens(emb, table):
    """emb [B,Np,D] (fp32 / bf16), table [N,D] fp32 with N > Np -> x [B,N,D] fp32: x[:, 1:1+Np] = emb + table[1:1+Np], every
    other row = table's (class token in front, point tokens behind; csrc/layernorm.hip as_assemble_tokens)."""
    lib = _lib.load()
    emb = emb.contiguous()
    _chk(emb)
    _chk(table, dtype=torch.float32)
    B, Np, D = emb.shape
    N = table.shape[0]
    x = torch.empty(B, N, D, device=emb.device, dtype=torch.float32)
    _lib.check(lib.as_assemble_tokens(_p(emb), _p(table), _p(x), B, Np, N, D, _dt(emb), _stream()), "as_assemble_tokens")
    return x


def maxpool_nhwc(x_nhwc, k):
    """nn.MaxPool2d(k, k) of a token-major fp32 map [B,H,W,C] -> [B,H/k,W/k,C] (csrc/layernorm.hip as_maxpool_nhwc).
    Every image must be dense ([H,W,C] contiguous); the images may be strided (the patch-token slice of [B,N,C])."""
    lib = _lib.load()
    B, H, W, C = x_nhwc.shape
    if not (x_nhwc.is_cuda and x_nhwc.dtype == torch.float32):
        raise AttnShiftError("maxpool_nhwc: fp32 device tensor expected")
    if x_nhwc.stride()[1:] != (W * C, C, 1) or (B > 1 and x_nhwc.stride(0) % 4):
        x_nhwc = x_nhwc.contiguous()
    bs = x_nhwc.stride(0) if B > 1 else H * W * C
    out = torch.empty(B, H // k, W // k, C, device=x_nhwc.device, dtype=torch.float32)
    _lib.check(lib.as_maxpool_nhwc(_p(x_nhwc), _p(out), B, H, W, C, int(k), int(bs), _stream()), "as_maxpool_nhwc")
    return out


def add_layernorm_bwd(x_out, dy, dx_res, gamma, eps, dtype, want_dx=True, want_ddelta=True, want_affine=True, delta_scale=None):
    """Backward of add_layernorm (csrc/layernorm.hip): x_out fp32 [.., D] saved by the forward, dy (`dtype`) | None,
    dx_res fp32 | None, gamma fp32 [D] | None -> (dx fp32 | None, ddelta `dtype` | None, dgamma, dbeta fp32 [D] | None)."""
    lib = _lib.load()
    D = x_out.shape[-1]
    x2 = x_out.reshape(-1, D)
    _chk(x2, dx_res, gamma, dtype=torch.float32)
    _chk(dy, dtype=dtype)
    M = x2.shape[0]
    dev = x_out.device
    scratch = not want_dx and not want_ddelta          # only the affine gradients are wanted: the kernel still writes ONE of them
    dx = torch.empty_like(x2) if (want_dx or scratch) else None
    dd = torch.empty(x2.shape, device=dev, dtype=dtype) if want_ddelta else None
    affine = want_affine and dy is not None
    dg = torch.empty(D, device=dev, dtype=torch.float32) if affine else None
    db = torch.empty(D, device=dev, dtype=torch.float32) if affine else None
    nbytes = lib.as_add_layernorm_bwd_workspace_bytes(M, D)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    dt = AS_BF16 if dtype == torch.bfloat16 else AS_F32
    sc, rps = (None, 1) if delta_scale is None else (delta_scale, x_out.shape[1] if x_out.dim() == 3 else 1)
    if sc is not None and (x_out.dim() != 3 or sc.numel() != x_out.shape[0]):
        raise AttnShiftError("add_layernorm_bwd: delta_scale needs x_out [B, N, D] with one scale per image")
    _lib.check(lib.as_add_layernorm_bwd_scaled(_p(x2), _p(dy), _p(dx_res), _p(gamma), float(eps), _p(dx), _p(dd), _p(dg), _p(db),
                                               _p(ws), nbytes, M, D, dt, _p(sc), rps, _stream()), "as_add_layernorm_bwd")
    return (None if dx is None or scratch else dx.reshape(x_out.shape), None if dd is None else dd.reshape(x_out.shape), dg, db)


class AttnLayerState:
    """What a layer must keep so its attention rows can be recomputed (q, k, lse)."""

    __slots__ = ("q", "k", "vt", "lse", "B", "N", "h", "dtype", "o")

    def __init__(self, q, k, vt, lse, B, N, h, dtype, o=None):
        self.q, self.k, self.vt, self.lse, self.B, self.N, self.h, self.dtype = q, k, vt, lse, B, N, h, dtype
        self.o = o                                  # pre-projection attention output, kept only for the backward


def _sdpa_ws(lib, B, N, h, like):
    """optional key-split workspace of as_sdpa_fwd (non-zero only for shapes whose grid leaves a short last round)"""
    n = lib.as_sdpa_fwd_workspace_bytes(B, N, h, _dt(like))
    return (torch.empty(n, device=like.device, dtype=torch.uint8), n) if n else (None, 0)


def attention_fwd(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, keep_state=True, keep_o=False):
    """Attention.forward (reference models/vision_transformer.py:74-86) -> (out [B,N,D], AttnLayerState)."""
    lib = _lib.load()
    B, N, D = x.shape
    _chk(x, w_qkv, w_proj)
    if D != num_heads * HEAD_DIM:
        raise AttnShiftError(f"head dim must be {HEAD_DIM} (D={D}, heads={num_heads})")
    Np_ = npad(N)
    dev, dt = x.device, x.dtype
    q = torch.empty(B, num_heads, Np_, HEAD_DIM, device=dev, dtype=dt)
    k = torch.empty_like(q)
    vt = torch.empty(B, num_heads, HEAD_DIM, Np_, device=dev, dtype=dt)
    o = torch.empty(B, N, D, device=dev, dtype=dt)
    out = torch.empty(B, N, D, device=dev, dtype=dt)
    lse = torch.empty(B, num_heads, N, device=dev, dtype=torch.float32)
    ws, wbytes = _sdpa_ws(lib, B, N, num_heads, x)
    if _TIMED is None:
        _lib.check(lib.as_attn_fwd(_p(x), _p(w_qkv), _p(b_qkv), _p(w_proj), _p(b_proj), _p(out), _p(lse), _p(q), _p(k),
                                   _p(vt), _p(o), _p(ws), wbytes, B, N, D, num_heads, _dt(x), _stream()), "as_attn_fwd")
    else:       # same three launches as as_attn_fwd, with event pairs around each
        with _timed("qkv_gemm"):
            _lib.check(lib.as_qkv_fwd(_p(x), _p(w_qkv), _p(b_qkv), _p(q), _p(k), _p(vt), B, N, D, num_heads, _dt(x),
                                      _stream()), "as_qkv_fwd")
        with _timed("sdpa_fwd"):
            _lib.check(lib.as_sdpa_fwd(_p(q), _p(k), _p(vt), _p(o), _p(lse), _p(ws), wbytes, B, N, num_heads, _dt(x),
                                       _stream()), "as_sdpa_fwd")
        with _timed("proj_gemm"):
            _lib.check(lib.as_linear_fwd(_p(o), _p(w_proj), _p(b_proj), _p(out), B * N, D, D, _dt(x), 0, _stream()),
                       "as_linear_fwd")
    return out, (AttnLayerState(q, k, vt, lse, B, N, num_heads, dt, o if keep_o else None) if keep_state else None)


def attention_bwd(x, w_qkv, w_proj, dout, state, want_bias=(True, True)):
    """Backward of attention_fwd: (dx, dWqkv, dbqkv, dWproj, dbproj); bias gradients fp32 (None if not wanted)."""
    lib = _lib.load()
    B, N, D = x.shape
    if state.o is None:
        raise AttnShiftError("attention_bwd needs the forward's pre-projection output: call attention_fwd(keep_o=True)")
    _chk(x, w_qkv, w_proj, dout, state.q, state.k, state.vt, state.o, state.lse)
    dev, dt = x.device, x.dtype
    dx = torch.empty_like(x)
    dwqkv = torch.empty_like(w_qkv)
    dwproj = torch.empty_like(w_proj)
    dbqkv = torch.empty(3 * D, device=dev, dtype=torch.float32) if want_bias[0] else None
    dbproj = torch.empty(D, device=dev, dtype=torch.float32) if want_bias[1] else None
    nbytes = lib.as_attn_bwd_workspace_bytes(B, N, D, state.h, _dt(x))
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with _timed("attn_bwd"):
        _lib.check(lib.as_attn_bwd(_p(x), _p(w_qkv), _p(w_proj), _p(dout), _p(state.q), _p(state.k), _p(state.vt),
                                   _p(state.o), _p(state.lse), _p(dx), _p(dwqkv), _p(dbqkv), _p(dwproj), _p(dbproj),
                                   _p(ws), nbytes, B, N, D, state.h, _dt(x), _stream()), "as_attn_bwd")
    return dx, dwqkv, dbqkv, dwproj, dbproj


# as_qkv_fwd stores q PRE-SCALED by log2(e) / sqrt(64) (rounded once from the fp32 accumulator, csrc/common.h): the
# scores q' . k of every consumer are base-2 logits.  Hand-built q workspaces must carry the same factor.
QSCALE = 0.125 * 1.4426950408889634
LN2 = 0.6931471805599453


def q_to_fragment_major(q_rows):
    """[B,h,Npad,64] row-major -> the fragment-major tile layout as_qkv_fwd writes (include/attnshift.h).  Layout only:
    the VALUES must already be q * QSCALE."""
    B, h, Np_, d = q_rows.shape
    t = q_rows.reshape(B, h, Np_ // 32, 32, 4, 2, 8)            # [.., tile, r, ks, half, e]
    return t.permute(0, 1, 2, 4, 5, 3, 6).contiguous().reshape(B, h, Np_, d)


def q_from_fragment_major(q_frag):
    B, h, Np_, d = q_frag.shape
    t = q_frag.reshape(B, h, Np_ // 32, 4, 2, 32, 8)            # [.., tile, ks, half, r, e]
    return t.permute(0, 1, 2, 5, 3, 4, 6).contiguous().reshape(B, h, Np_, d)


def qkv_fwd(x, w_qkv, b_qkv, num_heads):
    lib = _lib.load()
    B, N, D = x.shape
    _chk(x, w_qkv)
    Np_ = npad(N)
    q = torch.empty(B, num_heads, Np_, HEAD_DIM, device=x.device, dtype=x.dtype)
    k = torch.empty_like(q)
    vt = torch.empty(B, num_heads, HEAD_DIM, Np_, device=x.device, dtype=x.dtype)
    _lib.check(lib.as_qkv_fwd(_p(x), _p(w_qkv), _p(b_qkv), _p(q), _p(k), _p(vt), B, N, D, num_heads, _dt(x), _stream()),
               "as_qkv_fwd")
    return q, k, vt


def sdpa_fwd(q, k, vt, N):
    lib = _lib.load()
    B, h = q.shape[0], q.shape[1]
    _chk(q, k, vt)
    o = torch.empty(B, N, h * HEAD_DIM, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, h, N, device=q.device, dtype=torch.float32)
    ws, wbytes = _sdpa_ws(lib, B, N, h, q)
    _lib.check(lib.as_sdpa_fwd(_p(q), _p(k), _p(vt), _p(o), _p(lse), _p(ws), wbytes, B, N, h, _dt(q), _stream()),
               "as_sdpa_fwd")
    return o, lse


def sdpa_bwd(q, k, vt, o, d_o, lse, N):
    """Gradient of sdpa_fwd w.r.t. the QKV projection output: returns dqkv [B,N,3*h*64] in the reference's
    reshape(B,N,3,h,d) order (models/vision_transformer.py:76)."""
    lib = _lib.load()
    B, h = q.shape[0], q.shape[1]
    _chk(q, k, vt, o, d_o, lse)
    dqkv = torch.empty(B, N, 3 * h * HEAD_DIM, device=q.device, dtype=q.dtype)
    nbytes = lib.as_sdpa_bwd_workspace_bytes(B, N, h, _dt(q))
    ws = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    _lib.check(lib.as_sdpa_bwd(_p(q), _p(k), _p(vt), _p(o), _p(d_o), _p(lse), _p(dqkv), _p(ws), nbytes, B, N, h,
                               _dt(q), _stream()), "as_sdpa_bwd")
    return dqkv


def window_attention_fwd(qkv, b_qkv, table, num_heads, ws, shift, return_attn=False):
    """Fused Swin (shifted-)window attention on the un-partitioned grid: qkv [B,H,W,3C] (no bias) -> out [B,H,W,C]
    (+ softmax [B*nW,h,ws^2,ws^2] fp32 if asked).  See include/attnshift.h / csrc/window_attn.hip."""
    lib = _lib.load()
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    _chk(qkv)
    _chk(b_qkv, table, dtype=torch.float32)
    out = torch.empty(B, H, W, C, device=qkv.device, dtype=qkv.dtype)
    nW = -(-H // ws) * -(-W // ws)
    attn = torch.empty(B * nW, num_heads, ws * ws, ws * ws, device=qkv.device, dtype=torch.float32) if return_attn else None
    with _timed("window_attn_fwd"):
        _lib.check(lib.as_window_attn_fwd(_p(qkv), _p(b_qkv), _p(table), _p(out), _p(attn), B, H, W, C, num_heads, int(ws),
                                          int(shift), _dt(qkv), _stream()), "as_window_attn_fwd")
    if _TIMED is not None and "window_attn_fwd" in _TIMED:
        _TIMED.setdefault("window_attn_fwd:bytes", []).append(qkv.numel() * qkv.element_size() + out.numel() * out.element_size())
    return out, attn


def window_attention_bwd(qkv, b_qkv, table, d_out, num_heads, ws, shift):
    """Backward of window_attention_fwd -> (dqkv [B,H,W,3C], dtable [(2ws-1)^2,h] fp32, dbqkv_pad [3C] fp32)."""
    lib = _lib.load()
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    _chk(qkv, d_out)
    _chk(b_qkv, table, dtype=torch.float32)
    dqkv = torch.zeros_like(qkv)
    dtable = torch.empty_like(table)
    dpad = torch.empty(C3, device=qkv.device, dtype=torch.float32)
    nbytes = lib.as_window_attn_bwd_workspace_bytes(B, H, W, num_heads, int(ws))
    wsbuf = torch.empty(nbytes, device=qkv.device, dtype=torch.uint8)
    _lib.check(lib.as_window_attn_bwd(_p(qkv), _p(b_qkv), _p(table), _p(d_out), _p(dqkv), _p(dtable), _p(dpad), _p(wsbuf),
                                      nbytes, B, H, W, C, num_heads, int(ws), int(shift), _dt(qkv), _stream()),
               "as_window_attn_bwd")
    return dqkv, dtable, dpad


def attn_mean_rows(state, row0, nrows):
    """Head-mean softmax rows [B,nrows,N] fp32 recomputed from (q,k,lse)."""
    lib = _lib.load()
    out = torch.empty(state.B, nrows, state.N, device=state.q.device, dtype=torch.float32)
    _lib.check(lib.as_attn_mean_rows(_p(state.q), _p(state.k), _p(state.lse), _p(out), state.B, state.N, state.h,
                                     int(row0), int(nrows), _dt(state.q), _stream()), "as_attn_mean_rows")
    return out


def rollout_rows(states, num_point_tokens, rows=None):
    """Row-sliced attention roll-out over `states` (ordered bottom -> top, as the reference's
    attns[-cam_layer:]).  Returns [B, Lc, T, N] fp32; index k = product of the top k+1 layers
    (reference stdroi:1257-1272 + the row slice of :2272).  `rows` (long [B, G], indices into the T point-token rows):
    only those rows are pushed through the layers below the top one -> [B, Lc, G, N] (a row of the product depends on
    that row of R alone, so the values are the ones the full roll-out holds)."""
    lib = _lib.load()
    top = states[-1]
    T = int(num_point_tokens)
    dev, dt = top.q.device, _dt(top.q)
    nbytes = lib.as_rollout_rfrag_bytes(top.B, top.N, dt)
    outs = []
    R = torch.empty(top.B, T, top.N, device=dev, dtype=torch.float32)
    lower = list(reversed(states[:-1]))
    subset = rows is not None and len(lower) > 0
    rf = torch.empty(nbytes, device=dev, dtype=torch.uint8) if not subset else None
    _lib.check(lib.as_rollout_top(_p(top.q), _p(top.k), _p(top.lse), _p(R), _p(rf), top.B, top.N, top.h, T, dt, _stream()),
               "as_rollout_top")
    if rows is not None:
        R = torch.gather(R, 1, rows[:, :, None].expand(-1, -1, top.N)).contiguous()
        T = R.shape[1]
        if subset:
            rf = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            _lib.check(lib.as_rollout_pack(_p(R), _p(rf), top.B, top.N, T, dt, _stream()), "as_rollout_pack")
    outs.append(R)
    wbytes = lib.as_rollout_step_workspace_bytes(top.B, top.N, T) if lower else 0
    ws = torch.empty(wbytes, device=dev, dtype=torch.uint8) if wbytes else None     # contraction-split partials
    for n, st in enumerate(lower):
        Rn = torch.empty_like(R)
        rfn = torch.empty_like(rf) if n + 1 < len(lower) else None
        _lib.check(lib.as_rollout_step(_p(st.q), _p(st.k), _p(st.lse), _p(R), _p(rf), _p(Rn), _p(rfn), _p(ws), wbytes,
                                       st.B, st.N, st.h, T, dt, _stream()), "as_rollout_step")
        outs.append(Rn)
        R, rf = Rn, rfn
    return torch.stack(outs, dim=1)


def rollout_rows_dense(states, num_point_tokens, rows=None):
    """VALIDATION path for rollout_rows (bench.py's warm-up check and the full-size tests): the same rows through an
    independent route -- the dense head-mean matrix of every layer from as_attn_mean_rows ([B,N,N] fp32, O(N^2) memory)
    augmented and multiplied with fp32 torch matmuls exactly as stdroi:1257-1272 writes it.  Not used by the step."""
    T = int(num_point_tokens)
    run, outs = None, []
    for st in reversed(states):
        a = attn_mean_rows(st, 0, st.N)
        a.diagonal(dim1=1, dim2=2).add_(1.0)
        a /= a.sum(-1, keepdim=True)
        if run is None:
            run = a[:, st.N - T:, :]
            if rows is not None:
                run = torch.gather(run, 1, rows[:, :, None].expand(-1, -1, st.N))
        else:
            run = torch.bmm(run, a)
        outs.append(run.contiguous())
        del a
    return torch.stack(outs, dim=1)


# ------------------------------------------------------------------------------------------------
# Part B
# ------------------------------------------------------------------------------------------------
def ccl_2d(binary):
    """binary uint8 [M,H,W] (or [H,W]) -> int32 labels, 8-connectivity, label = 1 + min raster index."""
    lib = _lib.load()
    squeeze = binary.dim() == 2
    img = binary[None] if squeeze else binary
    _chk(img, dtype=torch.uint8)
    M, H, W = img.shape
    labels = torch.empty(M, H, W, device=img.device, dtype=torch.int32)
    _lib.check(lib.as_ccl_2d(_p(img), _p(labels), M, H, W, _stream()), "as_ccl_2d")
    return labels[0] if squeeze else labels


def cam_boxes(cams, points, cam_thr, area_ratio, up=16, return_upsampled=False, return_minmax=False):
    """cams [M,Hp,Wp] fp32, points [M,2] -> boxes [M,4], kept-pixel counts [M] (int32); with
    return_upsampled also the upsampled maps [M,H,W] and their per-map (min, max) [M,2]; with return_minmax
    (boxes, status, minmax) -- the maps are then never materialised."""
    lib = _lib.load()
    _chk(cams, points, dtype=torch.float32)
    M, Hp, Wp = cams.shape
    boxes = torch.empty(M, 4, device=cams.device, dtype=torch.float32)
    status = torch.empty(M, device=cams.device, dtype=torch.int32)
    cams_up = torch.empty(M, Hp * up, Wp * up, device=cams.device, dtype=torch.float32) if return_upsampled else None
    minmax = torch.empty(M, 2, device=cams.device, dtype=torch.float32) if (return_upsampled or return_minmax) else None
    nbytes = lib.as_cam_boxes_workspace_bytes(M, Hp, Wp, up)
    ws = torch.empty(nbytes, device=cams.device, dtype=torch.uint8)
    _lib.check(lib.as_cam_boxes(_p(cams), _p(points), float(cam_thr), float(area_ratio), M, Hp, Wp, up, _p(boxes),
                                _p(status), _p(cams_up), _p(minmax), _p(ws), nbytes, _stream()), "as_cam_boxes")
    if return_upsampled:
        return boxes, status, cams_up, minmax
    return (boxes, status, minmax) if return_minmax else (boxes, status)


def cam_sample_masks(cams, map_idx, minmax, thr_bg, thr_fg, up=16, want_counts=True):
    """cams [M,Hp,Wp] fp32, map_idx [G] int32, minmax [M,2] -> (masks [2G+1, H, W] uint8, counts [2G+1] int32):
    the background / foreground / shared-background candidate masks of the seed sampling (stdroi:1003-1007).
    want_counts=False: counts is None (the selection kernel behind it counts the rows anyway: no fill, no atomics)."""
    lib = _lib.load()
    _chk(cams, minmax, dtype=torch.float32)
    _chk(map_idx, dtype=torch.int32)
    M, Hp, Wp = cams.shape
    G = map_idx.shape[0]
    masks = torch.empty(2 * G + 1, Hp * up, Wp * up, device=cams.device, dtype=torch.uint8)
    counts = torch.empty(2 * G + 1, device=cams.device, dtype=torch.int32) if want_counts else None
    nbytes = lib.as_cam_sample_masks_workspace_bytes(G, Hp, Wp, up)
    ws = torch.empty(nbytes, device=cams.device, dtype=torch.uint8) if nbytes else None
    _lib.check(lib.as_cam_sample_masks(_p(cams), _p(map_idx), _p(minmax), G, Hp, Wp, up, float(thr_bg), float(thr_fg),
                                       _p(masks), _p(counts), _p(ws), nbytes, _stream()), "as_cam_sample_masks")
    return masks, counts


def semantic_prestage(map_fg, thr, k=11, up=16, want_counts=True):
    """map_fg [G,H,W] fp32 -> (fg_inter [G,H/up,W/up] fp32, mask uint8 same shape, counts [G] int32 | None):
    bilinear down-sampling of erode_k(map_fg > thr), its > thr binarisation and the per-object counts."""
    lib = _lib.load()
    _chk(map_fg, dtype=torch.float32)
    G, H, W = map_fg.shape
    hp, wp = H // up, W // up
    fg_inter = torch.empty(G, hp, wp, device=map_fg.device, dtype=torch.float32)
    mask = torch.empty(G, hp, wp, device=map_fg.device, dtype=torch.uint8)
    counts = torch.empty(G, device=map_fg.device, dtype=torch.int32) if want_counts else None
    _lib.check(lib.as_semantic_prestage(_p(map_fg), float(thr), int(k), G, hp, wp, up, _p(fg_inter), _p(mask), _p(counts),
                                        _stream()), "as_semantic_prestage")
    return fg_inter, mask, counts


# cosine-shift workspaces, reused across calls: (device, stream, shape) -> workspace.  Stream-ordered reuse is safe because the
# key holds the stream; bounded by BYTES (a full-size workspace is tens of MB) and guarded by a lock (the RoI head's optional
# per-image worker threads call in concurrently).
_shift_ws = {}
_shift_ws_lock = threading.Lock()
_SHIFT_WS_MAX_BYTES = 256 << 20


def cosine_shift(feat, box_patch, obj_img, prot, n_shift, hp, wp, tau0=0.1, temp=0.1, return_trace=False):
    """feat [B,Np,C] fp32 token-major; box_patch [G,4] int32; obj_img [G] int32; prot [G,P,C] (seeds, not modified).
    Returns (prot_out [G,P,C], sim [G,P,Np]) and, with return_trace, (assign [S,G,Np], tau [S,G,P])."""
    lib = _lib.load()
    _chk(prot, dtype=torch.float32)
    _chk(box_patch, obj_img, dtype=torch.int32)
    B, Np_, C = feat.shape
    G, P, _ = prot.shape
    if Np_ != hp * wp:
        raise AttnShiftError("cosine_shift: Np != hp*wp")
    # images may be apart by more than Np*C floats (a view of last_feat [B, 1 + Np, C] without its cls row): no copy
    if not (feat.is_cuda and feat.dtype == torch.float32 and feat.stride(2) == 1 and feat.stride(1) == C
            and (B == 1 or (feat.stride(0) >= Np_ * C and feat.stride(0) % 4 == 0)) and feat.data_ptr() % 16 == 0):
        raise AttnShiftError("cosine_shift: feat must be fp32 [B,Np,C] on the device with contiguous, 16-byte aligned image blocks")
    fbs = feat.stride(0) if B > 1 else Np_ * C
    prot_out = torch.empty_like(prot)
    sim = torch.empty(G, P, Np_, device=feat.device, dtype=torch.float32)
    assign = torch.empty(max(n_shift, 1), G, Np_, device=feat.device, dtype=torch.int32) if return_trace else None
    tau = torch.empty(max(n_shift, 1), G, P, device=feat.device, dtype=torch.float32) if return_trace else None
    key = (feat.device, torch.cuda.current_stream(feat.device).cuda_stream, B, C, hp, wp, G, P)
    with _shift_ws_lock:
        ws = _shift_ws.get(key)
        if ws is None:
            nbytes = lib.as_cosine_shift_workspace_bytes(B, C, hp, wp, G, P)
            if sum(t.numel() for t in _shift_ws.values()) + nbytes > _SHIFT_WS_MAX_BYTES:
                _shift_ws.clear()                       # (tensors still in flight stay alive through the caching allocator's stream rules)
            ws = torch.empty(nbytes, device=feat.device, dtype=torch.uint8)
            _shift_ws[key] = ws
    with _timed("cosine_shift"):
        _lib.check(lib.as_cosine_shift_strided(_p(feat), int(fbs), _p(box_patch), _p(obj_img), _p(prot), _p(prot_out), float(tau0),
                                               float(temp), int(n_shift), _p(sim), _p(assign), _p(tau), _p(ws), ws.numel(), B, C,
                                               hp, wp, G, P, _stream()), "as_cosine_shift")
    if return_trace:
        return prot_out, sim, assign[:n_shift], tau[:n_shift]
    return prot_out, sim


def refine_similarity(feat, seeds, boxes_patch, num_obj, refine_times, tau, is_select, hp, wp):
    """feat [Np,C], seeds [Gp,C], boxes_patch [G,4] int32 -> (maps [R+1,Gp,Np], seeds_out [Gp,C]).
    is_select: False / True as the reference's flag, or an int = size of the leading selection group (see the header)."""
    lib = _lib.load()
    _chk(feat, seeds, dtype=torch.float32)
    if boxes_patch is not None:
        _chk(boxes_patch, dtype=torch.int32)
    Np_, C = feat.shape
    Gp = seeds.shape[0]
    maps = torch.empty(refine_times + 1, Gp, Np_, device=feat.device, dtype=torch.float32)
    # (no refinement level: the seeds come back unchanged -- passed as their own output, which the library does not copy)
    seeds_out = seeds if refine_times == 0 else torch.empty(Gp, C, device=feat.device, dtype=torch.float32)
    nbytes = lib.as_refine_similarity_workspace_bytes(C, Np_, Gp)
    ws = torch.empty(nbytes, device=feat.device, dtype=torch.uint8)
    n_select = (Gp if is_select else 0) if isinstance(is_select, bool) else int(is_select)
    _lib.check(lib.as_refine_similarity(_p(feat), _p(seeds), _p(boxes_patch), int(num_obj), Gp, int(refine_times),
                                        float(tau), n_select, _p(maps), _p(seeds_out), _p(ws), nbytes, C,
                                        hp, wp, _stream()), "as_refine_similarity")
    return maps, seeds_out


def instance_maps(sim_fg, sim_bg, num_obj, hp, wp, up=16):
    """sim_fg [L,Gp,Np], sim_bg [L,G,Np] -> map_fg, map_bg [L,G,H,W] (stdroi:1010-1019)."""
    lib = _lib.load()
    _chk(sim_fg, sim_bg, dtype=torch.float32)
    L, Gp, _ = sim_fg.shape
    G = int(num_obj)
    H, W = hp * up, wp * up
    map_fg = torch.empty(L, G, H, W, device=sim_fg.device, dtype=torch.float32)
    map_bg = torch.empty(L, G, H, W, device=sim_fg.device, dtype=torch.float32)
    nbytes = lib.as_instance_maps_workspace_bytes(L, G)
    ws = torch.empty(nbytes, device=sim_fg.device, dtype=torch.uint8)
    _lib.check(lib.as_instance_maps(_p(sim_fg), _p(sim_bg), L, G, Gp, hp, wp, up, _p(map_fg), _p(map_bg), _p(ws), nbytes,
                                    _stream()), "as_instance_maps")
    return map_fg, map_bg


def crop_threshold_erode(maps, crops, thr, relative, k):
    """maps [M,H,W] fp32, crops [M,4] int32 (x0,y0,x1,y1 half-open) or None -> (mask uint8 [M,H,W], counts int32 [M]).
    mask = erode_k(map > (thr * cropmax if relative else thr)) restricted to the crop (stdroi:442-443, :2011)."""
    lib = _lib.load()
    _chk(maps, dtype=torch.float32)
    if crops is not None:
        _chk(crops, dtype=torch.int32)
    M, H, W = maps.shape
    mask = torch.empty(M, H, W, device=maps.device, dtype=torch.uint8)
    counts = torch.empty(M, device=maps.device, dtype=torch.int32)
    nbytes = lib.as_crop_threshold_erode_workspace_bytes(M, H, W)
    ws = torch.empty(nbytes, device=maps.device, dtype=torch.uint8)
    _lib.check(lib.as_crop_threshold_erode(_p(maps), _p(crops), float(thr), 1 if relative else 0, int(k), _p(mask),
                                           _p(counts), _p(ws), nbytes, M, H, W, _stream()), "as_crop_threshold_erode")
    return mask, counts


def mask_candidates(map_fg, map_bg, crops, pos_thr, neg_thr, mask_thr, k):
    """map_fg, map_bg [G,H,W] fp32, crops [G,4] int32 -> (pos, neg, pseudo uint8 [G,H,W], counts int32 [3,G]):
    the foreground / background point candidates of stdroi:442-443 and the pseudo mask of :2357 in one call."""
    lib = _lib.load()
    _chk(map_fg, map_bg, dtype=torch.float32)
    _chk(crops, dtype=torch.int32)
    G, H, W = map_fg.shape
    pos, neg, pseudo = (torch.empty(G, H, W, device=map_fg.device, dtype=torch.uint8) for _ in range(3))
    counts = torch.empty(3, G, device=map_fg.device, dtype=torch.int32)
    nbytes = lib.as_mask_candidates_workspace_bytes(G, H, W)
    ws = torch.empty(nbytes, device=map_fg.device, dtype=torch.uint8)
    _lib.check(lib.as_mask_candidates(_p(map_fg), _p(map_bg), _p(crops), float(pos_thr), float(neg_thr), float(mask_thr),
                                      int(k), _p(pos), _p(neg), _p(pseudo), _p(counts), _p(ws), nbytes, G, H, W,
                                      _stream()), "as_mask_candidates")
    return pos, neg, pseudo, counts


def part_stats_raw(maps, rois, owner, stride=16):
    """as_part_stats with the kernel's own output types: (c [M,2] fp32, yx [M,2] int32, area [M] int32, inside [M] uint8)."""
    lib = _lib.load()
    maps = maps.contiguous()
    rois = rois.contiguous().float()
    own = owner.to(torch.int32).contiguous()
    _chk(maps, rois, dtype=torch.float32)
    M, hp, wp = maps.shape
    c = torch.empty(M, 2, device=maps.device, dtype=torch.float32)
    yx = torch.empty(M, 2, device=maps.device, dtype=torch.int32)
    area = torch.empty(M, device=maps.device, dtype=torch.int32)
    inside = torch.empty(M, device=maps.device, dtype=torch.uint8)
    _lib.check(lib.as_part_stats(_p(maps), _p(rois), _p(own), float(stride), _p(c), _p(yx), _p(area), _p(inside), M, hp, wp,
                                 _stream()), "as_part_stats")
    return c, yx, area, inside


def part_stats(maps, rois, owner, stride=16):
    """maps [M,hp,wp] fp32, rois [G,4] fp32, owner [M] int -> (c [M,2] (x,y) image coords of the peak centroid,
    yx [M,2] long (integer centroid), area [M] long (pixels > 0.9), inside [M] bool) -- stdroi:222-262 per part."""
    c, yx, area, inside = part_stats_raw(maps, rois, owner, stride)
    return c, yx.long(), area.long(), inside.bool()


def part_select(area, inside, ngroups, c, yx, labels, feat_tok, G, P, wp, num_points):
    """Visiting order / cap logic of stdroi:222-262 + the gathers that follow it, on the device (as_part_select).
    area / inside / c / yx: part_stats_raw outputs for G*P slots; ngroups [G] int32; labels [G] int64; feat_tok [Np,C]
    fp32.  Returns padded (coords, coords_org [G*P,2], labels, labels_org, corres [G*P] int64, feats [G*P,C], split [G+1]
    int32: per-object counts + total); rows >= total are zero."""
    lib = _lib.load()
    _chk(c, feat_tok, dtype=torch.float32)
    _chk(area, yx, ngroups, dtype=torch.int32)
    _chk(inside, dtype=torch.uint8)
    _chk(labels, dtype=torch.int64)
    dev, M, C = c.device, G * P, feat_tok.shape[-1]
    coords = torch.empty(2, M, 2, device=dev, dtype=torch.float32)
    lab = torch.empty(3, M, device=dev, dtype=torch.int64)
    feats = torch.empty(M, C, device=dev, dtype=torch.float32)
    ints = torch.empty(M + G + 1, device=dev, dtype=torch.int32)          # sel_slot | split
    _lib.check(lib.as_part_select(_p(area), _p(inside), _p(ngroups), _p(c), _p(yx), _p(labels), _p(feat_tok), G, P, C, int(wp),
                                  int(num_points), _p(coords[0]), _p(coords[1]), _p(lab[0]), _p(lab[1]), _p(lab[2]), _p(feats),
                                  _p(ints[:M]), _p(ints[M:]), _stream()), "as_part_select")
    return coords[0], coords[1], lab[0], lab[1], lab[2], feats, ints[M:]


def filter_parts(sim, fg_inter, sim_thr=0.8, pos_thr=0.85):
    """sim [G,P,hp,wp] fp32, fg_inter [G,hp,wp] fp32 -> keep [G,P] bool (stdroi:263-271 filter_maps)."""
    lib = _lib.load()
    sim = sim.contiguous()
    fg = fg_inter.contiguous()
    _chk(sim, fg, dtype=torch.float32)
    G, P = sim.shape[:2]
    keep = torch.empty(G, P, device=sim.device, dtype=torch.uint8)
    _lib.check(lib.as_filter_parts(_p(sim), _p(fg), float(sim_thr), float(pos_thr), _p(keep), G, P, sim[0, 0].numel(),
                                   _stream()), "as_filter_parts")
    return keep.view(torch.bool)                      # 0/1 bytes: a view, not a conversion launch


def draw_distinct(counts, u, k, flag=None):
    """counts [G,2] int32 (n_pos, n_neg) in any strided layout (e.g. the transposed rows of mask_candidates), u [G,M] fp32
    uniform -> (rank_pos, rank_neg int32 [G,k], is_pos bool [G,k], flag int32 [1]): the first k distinct floor(u * n) per
    object, split by candidate kind.  `flag`: a zeroed int32 [1] slot of the caller's (OR-ed); made here when absent."""
    lib = _lib.load()
    if counts.dtype != torch.int32:
        counts = counts.to(torch.int32)
    if tuple(counts.shape) != (u.shape[0], 2):
        raise AttnShiftError(f"draw_distinct: counts {tuple(counts.shape)} for {u.shape[0]} objects")
    sg, sk = counts.stride()                               # any layout of the [G,2] table is read in place
    if sg <= 0 or sk <= 0:
        counts = counts.contiguous()
        sg, sk = 2, 1
    if not counts.is_cuda:
        raise AttnShiftError("hot-path ops need device (HBM) tensors; there is no CPU fallback")
    u = u.contiguous()
    _chk(u, dtype=torch.float32)
    G, M = u.shape
    rp = torch.empty(G, k, device=u.device, dtype=torch.int32)
    rn = torch.empty(G, k, device=u.device, dtype=torch.int32)
    ip = torch.empty(G, k, device=u.device, dtype=torch.uint8)
    if flag is None:
        flag = torch.zeros(1, device=u.device, dtype=torch.int32)
    _lib.check(lib.as_draw_distinct(_p(counts), sg, sk, _p(u), _p(rp), _p(rn), _p(ip), _p(flag), G, M, int(k), _stream()),
               "as_draw_distinct")
    return rp, rn, ip.view(torch.bool), flag


def small_attention_fwd(qkv):
    """qkv [Bp,N,3,h,32] fp32/bf16 -> (out [Bp,N,h*32] same dtype, lse [Bp,h,N] fp32): batched small-N self-attention."""
    lib = _lib.load()
    qkv = qkv.contiguous()
    _chk(qkv)
    Bp, N, _, h, d = qkv.shape
    out = torch.empty(Bp, N, h * d, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(Bp, h, N, device=qkv.device, dtype=torch.float32)
    _lib.check(lib.as_small_attn_fwd(_p(qkv), _p(out), _p(lse), Bp, N, h, d, _dt(qkv), _stream()), "as_small_attn_fwd")
    return out, lse


def small_attention_bwd(qkv, out, d_out, lse):
    lib = _lib.load()
    qkv, out, d_out = qkv.contiguous(), out.contiguous(), d_out.contiguous().to(qkv.dtype)
    _chk(qkv, out, d_out)
    Bp, N, _, h, d = qkv.shape
    dqkv = torch.empty_like(qkv)
    nbytes = lib.as_small_attn_bwd_workspace_bytes(Bp, N, h)
    ws = torch.empty(nbytes, device=qkv.device, dtype=torch.uint8)
    _lib.check(lib.as_small_attn_bwd(_p(qkv), _p(out), _p(d_out), _p(lse), _p(dqkv), _p(ws), nbytes, Bp, N, h, d, _dt(qkv),
                                     _stream()), "as_small_attn_bwd")
    return dqkv


def chamfer_2d_fwd(xyz1, xyz2):
    """xyz1 [B,n,2], xyz2 [B,m,2] fp32 -> (dist1 [B,n], dist2 [B,m] squared nearest distances, idx1, idx2 int32)."""
    lib = _lib.load()
    xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
    _chk(xyz1, xyz2, dtype=torch.float32)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = torch.empty(B, n, device=xyz1.device, dtype=torch.float32)
    d2 = torch.empty(B, m, device=xyz1.device, dtype=torch.float32)
    i1 = torch.empty(B, n, device=xyz1.device, dtype=torch.int32)
    i2 = torch.empty(B, m, device=xyz1.device, dtype=torch.int32)
    _lib.check(lib.as_chamfer_2d_fwd(_p(xyz1), _p(xyz2), _p(d1), _p(d2), _p(i1), _p(i2), B, n, m, _stream()), "as_chamfer_2d_fwd")
    return d1, d2, i1, i2


def chamfer_2d_bwd(xyz1, xyz2, g1, g2, idx1, idx2):
    lib = _lib.load()
    xyz1, xyz2, g1, g2 = xyz1.contiguous(), xyz2.contiguous(), g1.contiguous().float(), g2.contiguous().float()
    _chk(xyz1, xyz2, g1, g2, dtype=torch.float32)
    _chk(idx1, idx2, dtype=torch.int32)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    gx1, gx2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
    _lib.check(lib.as_chamfer_2d_bwd(_p(xyz1), _p(xyz2), _p(g1), _p(g2), _p(idx1), _p(idx2), _p(gx1), _p(gx2), B, n, m,
                                     _stream()), "as_chamfer_2d_bwd")
    return gx1, gx2


def merge_plan(keep, link):
    """keep [G,P] bool/uint8, link [G,P,P] bool/uint8 -> (groups [G,P] int32 bit sets, ngroups [G] int32): the greedy
    grouping of stdroi:278-294 for every object, on the device."""
    lib = _lib.load()
    k8 = keep.to(torch.uint8).contiguous()
    l8 = link.to(torch.uint8).contiguous()
    _chk(k8, l8)
    G, P = k8.shape
    groups = torch.empty(G, P, device=k8.device, dtype=torch.int32)
    ngroups = torch.empty(G, device=k8.device, dtype=torch.int32)
    _lib.check(lib.as_merge_plan(_p(k8), _p(l8), _p(groups), _p(ngroups), G, P, _stream()), "as_merge_plan")
    return groups, ngroups


def merge_parts(prot, keep, thr, slots, flag=None):
    """prot [G,P,C] fp32, keep [G,P] uint8/bool -> (merged [G,slots,C] fp32, ngroups [G] int32 clamped to slots): the
    cosine links, the greedy grouping (merge_plan) and the merged prototypes of stdroi:278-294 in one launch
    (as_merge_parts); `flag` (one int32 on the device) is OR-ed with 1 if an object has more than `slots` groups."""
    lib = _lib.load()
    prot = prot.contiguous()
    _chk(prot, dtype=torch.float32)
    k8 = keep.view(torch.uint8) if keep.dtype == torch.bool else keep
    k8 = k8.contiguous()
    _chk(k8, dtype=torch.uint8)
    G, P, C = prot.shape
    merged = torch.empty(G, slots, C, device=prot.device, dtype=torch.float32)
    ngroups = torch.empty(G, device=prot.device, dtype=torch.int32)
    _lib.check(lib.as_merge_parts(_p(prot), _p(k8), float(thr), _p(merged), _p(ngroups), _p(flag), G, P, C, int(slots),
                                  _stream()), "as_merge_parts")
    return merged, ngroups


def select_median_boxes(boxes, meta, Lc, stride, status=None, bad=None, pick_in=None):
    """boxes [rows,4] fp32 (every image's CAM boxes layer-major), meta [n,3] int32 = (image's first row, objects in the image,
    index in the image) -> (pick [n] int64, chosen [n,4] fp32, map_idx [n] int32, box_patch [n,4] int32, box_int [n,4]
    int32): the median-area layer per object, its box, that box's row in `boxes`, floor(box / stride) and the box truncated
    to integers (as_select_median_boxes).  With `status` ([rows] int32 of cam_boxes) the zeroed int32 slot `bad` is OR-ed
    with 1 if any box of a listed object has status <= 0.  `pick_in` [n] int64: another selector's layer choice (the median
    rule is then skipped; `pick` returns it clamped to [0, Lc))."""
    lib = _lib.load()
    _chk(boxes, dtype=torch.float32)
    _chk(meta, status, bad, dtype=torch.int32)
    _chk(pick_in, dtype=torch.int64)
    n = meta.shape[0]
    pick = torch.empty(n, device=boxes.device, dtype=torch.int64)
    chosen = torch.empty(n, 4, device=boxes.device, dtype=torch.float32)
    map_idx = torch.empty(n, device=boxes.device, dtype=torch.int32)
    ints = torch.empty(2, n, 4, device=boxes.device, dtype=torch.int32)
    if pick_in is not None and pick_in.numel() != n:
        raise AttnShiftError(f"select_median_boxes: {pick_in.numel()} choices for {n} objects")
    _lib.check(lib.as_select_median_boxes(_p(boxes), _p(meta), int(Lc), int(stride), _p(pick_in), _p(pick), _p(chosen), _p(map_idx),
                                          _p(ints[0]), _p(ints[1]), _p(status), _p(bad), n, _stream()),
               "as_select_median_boxes")
    return pick, chosen, map_idx, ints[0], ints[1]


def mask_count(mask):
    """mask bool/uint8 [M,HW] (HW % 16 == 0) -> int32 [M] number of set elements per row."""
    lib = _lib.load()
    m = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    _chk(m, dtype=torch.uint8)
    M, HW = m.shape
    out = torch.empty(M, device=m.device, dtype=torch.int32)
    _lib.check(lib.as_mask_count(_p(m), _p(out), M, HW, _stream()), "as_mask_count")
    return out


def mt_sample_ranks(state, counts, k):
    """Device draws of sample_point_grid in the reference's stream (as_mt_sample_ranks): state int32[626] on the device
    (advanced in place), counts int32 [S] -> (ranks int32 [S,k], flag int32 [1])."""
    lib = _lib.load()
    _chk(state, counts, dtype=torch.int32)
    S = counts.numel()
    ranks = torch.zeros(S, k, device=counts.device, dtype=torch.int32)
    flag = torch.empty(1, device=counts.device, dtype=torch.int32)
    _lib.check(lib.as_mt_sample_ranks(_p(state), _p(counts), _p(ranks), _p(flag), S, int(k), _stream()), "as_mt_sample_ranks")
    return ranks, flag


def mt_perm_ranks(state, counts2, k):
    """Device draws of torch.randperm(n)[:k] per object in the reference's stream (as_mt_perm_ranks): counts2 int32 [G,2]
    -> (ranks int32 [G,k], flag int32 [1])."""
    lib = _lib.load()
    _chk(state, counts2, dtype=torch.int32)
    G = counts2.shape[0]
    ranks = torch.zeros(G, k, device=counts2.device, dtype=torch.int32)
    flag = torch.empty(1, device=counts2.device, dtype=torch.int32)
    _lib.check(lib.as_mt_perm_ranks(_p(state), _p(counts2), _p(ranks), _p(flag), G, int(k), _stream()), "as_mt_perm_ranks")
    return ranks, flag


def rank_select(mask, ranks):
    """mask uint8 [M,HW] (0/1), ranks int [M,K] -> int64 [M,K] flat index of the ranks[m,k]-th set byte of row m in
    raster order (= mask[m].nonzero()[rank]), -1 if out of range."""
    lib = _lib.load()
    _chk(mask, dtype=torch.uint8)
    r32 = ranks.to(torch.int32).contiguous()
    _chk(r32)
    M, HW = mask.shape
    K = r32.shape[1]
    out = torch.empty(M, K, device=mask.device, dtype=torch.int32)
    nbytes = lib.as_rank_select_workspace_bytes(M, HW)
    ws = torch.empty(nbytes, device=mask.device, dtype=torch.uint8)
    _lib.check(lib.as_rank_select(_p(mask), _p(r32), _p(out), _p(ws), nbytes, M, HW, K, _stream()), "as_rank_select")
    return out.long()


def rank_select_xy(mask, ranks, W, yx=False):
    """rank_select returning coordinates: int64 [M,K,2] = (x, y) (or (y, x) with yx) of the ranks[m,k]-th set byte of row m
    on a W-wide grid; out-of-range ranks give pixel 0 (one launch pair instead of rank_select + clamp / % / // / stack)."""
    lib = _lib.load()
    _chk(mask, dtype=torch.uint8)
    r32 = ranks if ranks.dtype == torch.int32 and ranks.is_contiguous() else ranks.to(torch.int32).contiguous()
    _chk(r32)
    M, HW = mask.shape
    K = r32.shape[1]
    out = torch.empty(M, K, 2, device=mask.device, dtype=torch.int64)
    nbytes = lib.as_rank_select_workspace_bytes(M, HW)
    ws = torch.empty(nbytes, device=mask.device, dtype=torch.uint8)
    _lib.check(lib.as_rank_select_xy(_p(mask), _p(r32), _p(out), _p(ws), nbytes, M, HW, K, int(W), 1 if yx else 0, _stream()),
               "as_rank_select_xy")
    return out


def rank_draw_xy(mask, K, W, u=None, flag=None, yx=False, patch=None, want_xy=True):
    """rank_select_xy whose ranks come from each row's population n on the device (as_rank_draw_xy): with `u` [M,K] uniform
    numbers rank = min(int(u * n), max(n - 1, 0)) (the fast-RNG seed draws); without, the K grid-strided positives
    k * max(n // K, 1).  `flag` (int32 [1] or 0-dim view, OR-ed with 1 when some row has fewer than K set bytes).
    patch = (out [M,K] int64 or None, div, width, base, row_rot): also the token index base + (y // div) * width + x // div
    of every selected pixel, row m stored at row (m - row_rot) mod M (the gather index of the seed features).
    Returns xy [M,K,2] int64, or (xy | None, patch_out) with `patch`."""
    lib = _lib.load()
    _chk(mask, dtype=torch.uint8)
    M, HW = mask.shape
    if u is not None:
        _chk(u, dtype=torch.float32)
        if tuple(u.shape) != (M, K):
            raise AttnShiftError("rank_draw_xy: u must be [M, K]")
    if flag is not None and (flag.dtype != torch.int32 or flag.numel() != 1 or not flag.is_cuda):
        raise AttnShiftError("rank_draw_xy: flag must be one int32 on the device")
    out = torch.empty(M, K, 2, device=mask.device, dtype=torch.int64) if want_xy or patch is None else None
    pout, pdiv, pw, pbase, prot = None, 1, 1, 0, 0
    if patch is not None:
        pout, pdiv, pw, pbase, prot = patch
        if pout is None:
            pout = torch.empty(M, K, device=mask.device, dtype=torch.int64)
        if pout.dtype != torch.int64 or pout.numel() != M * K or not pout.is_contiguous():
            raise AttnShiftError("rank_draw_xy: patch output must be a contiguous int64 [M, K]")
    nbytes = lib.as_rank_select_workspace_bytes(M, HW)
    ws = torch.empty(nbytes, device=mask.device, dtype=torch.uint8)
    _lib.check(lib.as_rank_draw_xy(_p(mask), 1 if u is not None else 2, _p(u), _p(flag), _p(out), _p(pout), int(pdiv), int(pw),
                                   int(pbase), int(prot), _p(ws), nbytes, M, HW, int(K), int(W), 1 if yx else 0, _stream()),
               "as_rank_draw_xy")
    return out if patch is None else (out, pout)


def roi_align_fwd(feat_nhwc, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """feat [B,H,W,C] fp32 token-major, rois [R,5] fp32 -> [R, out*out, C] (csrc/roi_align.hip)."""
    lib = _lib.load()
    _chk(feat_nhwc, rois, dtype=torch.float32)
    B, H, W, C = feat_nhwc.shape
    R = rois.shape[0]
    out = torch.empty(R, out_size * out_size, C, device=feat_nhwc.device, dtype=torch.float32)
    _lib.check(lib.as_roi_align_fwd(_p(feat_nhwc), _p(rois), _p(out), B, H, W, C, R, int(out_size), float(spatial_scale),
                                    int(sampling_ratio), int(bool(aligned)), _stream()), "as_roi_align_fwd")
    return out


def roi_align_bwd(dout, rois, shape, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """dout [R, out*out, C] -> dfeat [B,H,W,C] (float atomics)."""
    lib = _lib.load()
    _chk(dout, rois, dtype=torch.float32)
    B, H, W, C = shape
    dfeat = torch.empty(B, H, W, C, device=dout.device, dtype=torch.float32)
    _lib.check(lib.as_roi_align_bwd(_p(dout), _p(rois), _p(dfeat), B, H, W, C, rois.shape[0], int(out_size),
                                    float(spatial_scale), int(sampling_ratio), int(bool(aligned)), _stream()), "as_roi_align_bwd")
    return dfeat
