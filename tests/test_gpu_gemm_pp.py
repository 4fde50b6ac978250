"""The persistent ping-pong GEMM (csrc/gemm_pp.hip, round 6) against an fp32 reference of the same product AND against the
one-tile-per-workgroup kernels of csrc/gemm.hip, bit for bit: both accumulate every output as one k-ordered fp32 chain on
the MFMA and share their epilogue arithmetic (csrc/gemm_epi.h), so whichever kernel the round model picks for a shape, the
result is the same -- the fixture-backed parity tests of the backbone hold for both.  Replaces nn.Linear / the fused QKV
projection / nn.GELU of Attention and Mlp (reference models/vision_transformer.py:47-59, 75-77, 84) and the FPN's
ConvTranspose2d (mmdet/models/backbones/visual_transformer_det.py:107-117).

Covers: both tile shapes (a = 256 x 256, b = 256 x 128), ragged M (clamped rows, masked stores), several tiles per
workgroup (the cross-tile operand stream), the shortest legal K (two K steps), bias / GELU / ReLU, the training epilogues
(pre-activation + GELU; GELU' times the accumulator), the QKV scatter (fragment-major pre-scaled q, k, V^T) incl. a tile
that straddles two images, the deconvolution scatter, and run-to-run bitwise reproducibility."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def rnd(*shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return ((torch.rand(*shape, device="cuda", generator=g) * 2 - 1) * scale).to(dtype)


class force:
    """AS_GEMM_PP for the calls inside the block: "0" = gemm.hip, "a" / "b" = the persistent kernel's tile shapes."""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = os.environ.get("AS_GEMM_PP")
        os.environ["AS_GEMM_PP"] = self.v

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("AS_GEMM_PP", None)
        else:
            os.environ["AS_GEMM_PP"] = self.old


SHAPES = [  # M, N, K, act
    (777, 512, 192, "relu"),        # ragged last panel, 3 K steps
    (256, 256, 128, "none"),        # one tile, the shortest stream (2 K steps)
    (1000, 768, 768, "gelu"),
    (4197, 1536, 320, "none"),      # 17 panels, 5 K steps
    (8394, 3072, 768, "gelu"),      # fc1 of BASELINE config 2: 396 / 792 tiles, several per workgroup
    (8394, 768, 3072, "none"),      # fc2
]


@pytest.mark.parametrize("M,N,K,act", SHAPES)
@pytest.mark.parametrize("cfg", ["a", "b"])
def test_linear_equals_reference_and_gemm_hip_bitwise(M, N, K, act, cfg):
    from attentionshift_amd import ops
    x, w = rnd(M, K, seed=M + K), rnd(N, K, seed=N + K + 1)
    b = rnd(N, seed=7, dtype=torch.float32)
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    ref = torch.nn.functional.gelu(ref) if act == "gelu" else torch.relu(ref) if act == "relu" else ref
    with force("0"):
        old = ops.linear(x, w, b, act=act)
    with force(cfg):
        new = ops.linear(x, w, b, act=act)
        again = ops.linear(x, w, b, act=act)
        nobias = ops.linear(x, w, None, act=act)
    with force("0"):
        old_nobias = ops.linear(x, w, None, act=act)
    assert torch.isfinite(new.float()).all()
    err = float((new.float() - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err                                  # bf16 rounding of the result: 2^-8 of the output range
    assert torch.equal(new, old), float((new.float() - old.float()).abs().max())
    assert torch.equal(new, again)
    assert torch.equal(nobias, old_nobias)


def test_transposes_are_detected():
    """Asymmetric operands with distinct rows AND columns: out[m, n] = (m + 1) * 2^-(n % 7) from one-hot weights."""
    from attentionshift_amd import ops
    M, N, K = 512, 256, 256
    x = torch.zeros(M, K, device="cuda")
    x[:, 0] = torch.arange(1, M + 1, device="cuda", dtype=torch.float32) / 64
    w = torch.zeros(N, K, device="cuda")
    w[:, 0] = 2.0 ** -(torch.arange(N, device="cuda") % 7).float()
    ref = x @ w.t()
    for cfg in ("a", "b"):
        with force(cfg):
            out = ops.linear(x.bfloat16(), w.bfloat16(), None)
        assert torch.equal(out.float(), ref.bfloat16().float()), cfg


@pytest.mark.parametrize("M,N,K", [(1000, 768, 256), (8394, 3072, 768), (8394, 768, 3072)])
@pytest.mark.parametrize("cfg", ["a", "b"])
def test_training_epilogues_equal_gemm_hip_bitwise(M, N, K, cfg):
    """as_linear_gelu_fwd (h and GELU(h) in one pass) and as_linear_dgelu_fwd (acc * GELU'(h)) -- the Mlp under autograd."""
    from attentionshift_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
    b = rnd(N, seed=3, dtype=torch.float32)
    res = {}
    for v in ("0", cfg):
        with force(v):
            out, pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            _lib.check(lib.as_linear_gelu_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), pre.data_ptr(), M, N, K, 1, st), "gelu_fwd")
            dg = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            _lib.check(lib.as_linear_dgelu_fwd(x.data_ptr(), w.data_ptr(), pre.data_ptr(), dg.data_ptr(), M, N, K, 1, st), "dgelu_fwd")
            res[v] = (out, pre, dg)
    for got, want, name in zip(res[cfg], res["0"], ("GELU(h)", "h", "acc * GELU'(h)")):
        assert torch.equal(got, want), name
    h = torch.nn.functional.linear(x.float(), w.float(), b)
    assert float((res[cfg][1].float() - h).abs().max() / h.abs().max()) < 6e-3
    assert float((res[cfg][0].float() - torch.nn.functional.gelu(res[cfg][1].float())).abs().max()) < 2e-2


@pytest.mark.parametrize("B,N,D,h", [(3, 333, 256, 4), (2, 4197, 768, 12), (1, 6501, 1024, 16)])
def test_qkv_scatter_equals_gemm_hip_bitwise(B, N, D, h):
    """q (fragment-major, pre-scaled), k and V^T of as_qkv_fwd: the 256 x 128 tiles of the persistent kernel against gemm.hip,
    and k / V^T against the fp32 product.  (3, 333): 999 rows = tiles that straddle two images, a ragged last panel."""
    from attentionshift_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    Npad = lib.as_npad(N)
    x, w = rnd(B, N, D, seed=B * N), rnd(3 * D, D, seed=D, scale=0.05)
    b = rnd(3 * D, seed=5, dtype=torch.float32)
    res = {}
    for v in ("0", "b"):
        with force(v):
            q = torch.zeros(B, h, Npad, 64, device="cuda", dtype=torch.bfloat16)
            k, vt = torch.zeros_like(q), torch.zeros(B, h, 64, Npad, device="cuda", dtype=torch.bfloat16)
            _lib.check(lib.as_qkv_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, N, D, h, 1, st), "qkv")
            res[v] = (q, k, vt)
    for got, want, name in zip(res["b"], res["0"], "q k vt".split()):
        assert torch.equal(got, want), name
    y = torch.nn.functional.linear(x.float(), w.float(), b).reshape(B, N, 3, h, 64)
    kref, vref = y[:, :, 1].permute(0, 2, 1, 3), y[:, :, 2].permute(0, 2, 3, 1)
    assert float((res["b"][1][:, :, :N].float() - kref).abs().max() / kref.abs().max()) < 6e-3
    assert float((res["b"][2][:, :, :, :N].float() - vref).abs().max() / vref.abs().max()) < 6e-3


@pytest.mark.parametrize("cfg", ["a", "b"])
@pytest.mark.parametrize("act", ["none", "gelu"])
def test_deconv_scatter_equals_gemm_hip_bitwise(cfg, act):
    from attentionshift_amd import ops
    B, hh, ww, cin, cout = 2, 24, 20, 256, 192
    x = rnd(B, hh, ww, cin, seed=11)
    w4 = rnd(4 * cout, cin, seed=12, scale=0.1)
    bias = rnd(cout, seed=13, dtype=torch.float32)
    b4 = bias.repeat(4).contiguous()
    with force("0"):
        old = ops.deconv2x2(x, w4, b4, act=act)
    with force(cfg):
        new = ops.deconv2x2(x, w4, b4, act=act)
    assert torch.equal(new, old)
    wt = w4.float().reshape(2, 2, cout, cin).permute(3, 2, 0, 1)                       # [cin, cout, di, dj]
    ref = torch.nn.functional.conv_transpose2d(x.float().permute(0, 3, 1, 2), wt, bias, stride=2)
    ref = torch.nn.functional.gelu(ref) if act == "gelu" else ref
    assert float((new.float().permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()) < 6e-3


@pytest.mark.parametrize("M,N,K,act", [(200, 1536, 768, "relu"), (200, 21, 768, "none"), (200, 2, 768, "sigmoid"),
                                       (333, 100, 256, "gelu"), (1, 64, 64, "none"), (97, 70, 1024, "relu")])
def test_linear_small_fp32_matches_torch(M, N, K, act):
    """as_linear_small_fwd (the point head's FFNs, visual_transformer_det.py:26-38): exact-fp32 MFMA chains with the K range
    split over a workgroup's four waves, against torch's fp32 linear; ragged rows and columns; deterministic."""
    from attentionshift_amd import ops
    x, w = rnd(M, K, seed=M, dtype=torch.float32), rnd(N, K, seed=N, scale=0.1, dtype=torch.float32)
    b = rnd(N, seed=3, dtype=torch.float32)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = {"relu": torch.relu, "gelu": torch.nn.functional.gelu, "sigmoid": torch.sigmoid, "none": lambda t_: t_}[act](ref)
    out = ops.linear_small(x, w, b, act=act)
    assert float((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) < 2e-6
    assert torch.equal(out, ops.linear_small(x, w, b, act=act))
    nb = ops.linear_small(x, w, None, act="none")
    assert float((nb.double() - torch.nn.functional.linear(x.double(), w.double())).abs().max()) < 1e-4


def test_linear_small_reads_and_writes_column_slices_in_place():
    """The two heads share a packed first layer: each second layer reads ITS half of the [M, 2 hc] activation through the row
    stride, and an output slice of a wider buffer is written without touching its neighbours."""
    from attentionshift_amd import ops
    M, hc, K = 200, 768, 768
    h1 = rnd(M, 2 * hc, seed=1, dtype=torch.float32)
    w = rnd(hc, K, seed=2, scale=0.1, dtype=torch.float32)
    for sl in (slice(0, hc), slice(hc, 2 * hc)):
        got = ops.linear_small(h1[:, sl], w, None)
        assert torch.equal(got, ops.linear_small(h1[:, sl].contiguous(), w, None))
    wide = torch.full((M, 3 * hc), -7.0, device="cuda")
    ops.linear_small(h1[:, :hc], w, None, out=wide[:, hc:2 * hc])
    assert torch.equal(wide[:, hc:2 * hc], ops.linear_small(h1[:, :hc], w, None))
    assert (wide[:, :hc] == -7.0).all() and (wide[:, 2 * hc:] == -7.0).all()


# ---- stream-K tail (round 6): launches with more than one round of tiles and a ragged last one share the tail out by K steps; an
# open tile is handed to the next workgroup through a slab of fp32 accumulators (one k-ordered chain per output still: bit-identical)
SK_SHAPES = [  # M, N, K, act, cfg -- tiles / 256 workgroups
    (8394, 3072, 768, "gelu", "a"),     # 396 tiles: D = 0, 396 tail tiles (every workgroup: open head, whole tile or not, open tail)
    (8394, 3072, 768, "none", "b"),     # 792: two whole rounds + 280 tail tiles
    (8394, 2304, 768, "none", "b"),     # 594 (the QKV shape as a plain linear): one whole round + 338
    (8394, 2304, 128, "relu", "b"),     # the shortest stream: 2 K steps per tile, open pieces of ONE K step
    (10402, 4096, 1024, "gelu", "a"),   # ViT-L fc1: 41 x 16 = 656 tiles: D = 1, 400 tail tiles
    (6501, 3072, 192, "none", "b"),     # 26 x 24 = 624: ragged last panel inside the tail
    (16500, 2048, 256, "none", "a"),    # 65 x 8 = 520: D = 1, 264 tail tiles (a tail barely longer than one round)
    (16500, 2048, 256, "none", "b"),    # 65 x 16 = 1040: D = 3, 272
]


@pytest.mark.parametrize("M,N,K,act,cfg", SK_SHAPES)
def test_stream_k_tail_is_bitwise_the_whole_tile_kernel(M, N, K, act, cfg):
    from attentionshift_amd import ops
    x, w = rnd(M, K, seed=M + K + 3), rnd(N, K, seed=N + K + 5)
    b = rnd(N, seed=11, dtype=torch.float32)
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    ref = torch.nn.functional.gelu(ref) if act == "gelu" else torch.relu(ref) if act == "relu" else ref
    os.environ["AS_GEMM_PP_SK"] = "0"
    try:
        with force(cfg):
            whole = ops.linear(x, w, b, act=act)
        os.environ["AS_GEMM_PP_SK"] = "1"
        with force(cfg):
            outs = [ops.linear(x, w, b, act=act) for _ in range(4)]       # the slabs / flags of one stream, launch after launch
    finally:
        os.environ.pop("AS_GEMM_PP_SK", None)
    err = float((outs[0].float() - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err
    for o in outs:
        assert torch.equal(o, whole), float((o.float() - whole.float()).abs().max())


def test_stream_k_many_launches_two_streams():
    """200 launches alternating two shapes on each of two streams that run concurrently: every launch reuses its stream's slabs and
    flags (the flags are never reset, only compared with the launch's epoch) -- every result bit-identical to the first."""
    from attentionshift_amd import ops
    shapes = [(8394, 3072, 768, "gelu"), (8394, 2304, 768, "none")]
    data = []
    for (M, N, K, act) in shapes:
        x, w = rnd(M, K, seed=M + N), rnd(N, K, seed=N + K)
        b = rnd(N, seed=2, dtype=torch.float32)
        os.environ["AS_GEMM_PP_SK"] = "0"
        want = ops.linear(x, w, b, act=act)
        data.append((x, w, b, act, want))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = 0
    os.environ["AS_GEMM_PP_SK"] = "1"                       # (forced: the round model leaves these shapes with whole tiles)
    for it in range(50):
        outs = []
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                for j in range(2):
                    x, w, b, act, want = data[(it + si + j) % 2]
                    outs.append((ops.linear(x, w, b, act=act), want))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, want) else 1 for o, want in outs)
    os.environ.pop("AS_GEMM_PP_SK", None)
    assert bad == 0, bad
