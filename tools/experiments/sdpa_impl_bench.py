"""A/B of the attention-forward kernels behind as_sdpa_fwd (AS_SDPA_IMPL: 0 = sdpa_fwd_glds_kernel, 1 / 2 =
sdpa_fwd_pipe_kernel<1 / 2>) on the GPU box: max error against an fp32 torch reference on the same bf16 operands,
agreement of lse, and interleaved timing rounds (median / min).

    python tools/experiments/sdpa_impl_bench.py [--impls 0,1,2] [--rounds 7] [--shapes 2x12x4197,1x16x6501]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attentionshift_amd import ops  # noqa: E402


def reference(q_rows, k, vt, N):
    q, kk, v = q_rows[:, :, :N].float(), k[:, :, :N].float(), vt[:, :, :, :N].float().transpose(-1, -2)
    s = (q @ kk.transpose(-1, -2)) * ops.LN2             # q is stored pre-scaled by log2(e) / 8
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ v                       # [B,h,N,64]
    return o.permute(0, 2, 1, 3).reshape(q.shape[0], N, -1), lse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impls", default="0,1,2")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", default="2x12x4197")
    ap.add_argument("--scale", type=float, default=0.06)
    a = ap.parse_args()
    impls = [int(v) for v in a.impls.split(",")]
    for shp in a.shapes.split(","):
        B, h, N = (int(v) for v in shp.split("x"))
        D = 64 * h
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, N, D, generator=g).cuda().bfloat16()
        w = (torch.randn(3 * D, D, generator=g) * a.scale).cuda().bfloat16()
        bias = (torch.randn(3 * D, generator=g) * 0.1).cuda()
        q, k, vt = ops.qkv_fwd(x, w, bias, h)
        ref_o, ref_lse = reference(ops.q_from_fragment_major(q), k, vt, N)
        flops = 4.0 * B * h * N * N * 64
        outs = {}
        for im in impls:
            os.environ["AS_SDPA_IMPL"] = str(im)
            o, lse = ops.sdpa_fwd(q, k, vt, N)
            torch.cuda.synchronize()
            err = (o.float() - ref_o).abs().max().item() / ref_o.abs().max().item()
            lerr = (lse - ref_lse).abs().max().item()
            outs[im] = o
            print(f"[{shp}] impl {im}: max err / range {err:.3e}   lse max abs err {lerr:.3e}   finite {bool(torch.isfinite(o.float()).all())}", flush=True)
        times = {im: [] for im in impls}
        for r in range(a.rounds):
            for im in impls:
                os.environ["AS_SDPA_IMPL"] = str(im)
                for _ in range(3):
                    ops.sdpa_fwd(q, k, vt, N)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    ops.sdpa_fwd(q, k, vt, N)
                e1.record()
                torch.cuda.synchronize()
                times[im].append(e0.elapsed_time(e1) / a.reps)
        for im in impls:
            t = sorted(times[im])
            med, mn = t[len(t) // 2], t[0]
            print(f"[{shp}] impl {im}: median {med * 1e3:7.1f} us ({flops / med / 1e9:6.0f} TFLOP/s)   min {mn * 1e3:7.1f} us "
                  f"({flops / mn / 1e9:6.0f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
