"""A/B + refcheck of the persistent ping-pong GEMM (csrc/gemm_pp.hip) against the one-tile-per-workgroup kernels (csrc/gemm.hip)
and the vendor library, in ONE process with interleaved rounds (cdna_hip_programming.md 5.4 rule 24).

    python tools/experiments/gemm_pp_bench.py [--rounds 5] [--reps 30] [--lib] [--qkv]            (GPU box)

AS_GEMM_PP is switched per call (AS_GEMM_PP_DYN=1): "0" = gemm.hip, "a" = 256 x 256 tiles, "b" = 256 x 128 tiles; the digit behind
the letter is AS_GEMM_PP_SK (0 = whole tiles only, 1 = stream-K tail where the launch has a ragged last round).
Every variant is checked against an fp32 reference of the same product (max error relative to the output range) on
uniform [-1, 1) operands; the timing is on the same operands.  One JSON line per (shape, act, variant).
"""
import argparse
import json
import os
import statistics
import sys

os.environ["AS_GEMM_PP_DYN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPES = [  # (name, M, N, K, act)
    ("qkv_plain", 8394, 2304, 768, 0), ("proj", 8394, 768, 768, 0), ("fc1", 8394, 3072, 768, 0), ("fc1_gelu", 8394, 3072, 768, 1),
    ("fc2", 8394, 768, 3072, 0), ("sq4096", 4096, 4096, 4096, 0), ("vitl_fc1", 10402, 4096, 1024, 1), ("vitl_fc2", 10402, 1024, 4096, 0),
    ("one_tile", 256, 256, 128, 0), ("ragged", 777, 512, 192, 4),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--lib", action="store_true", help="also time torch.nn.functional.linear (hipBLASLt)")
    ap.add_argument("--qkv", action="store_true", help="as_qkv_fwd: parity of q / k / V^T against gemm.hip + timing")
    ap.add_argument("--only", default="")
    ap.add_argument("--variants", default="0,a0,a1,b0,b1", help="0 = gemm.hip; a / b = tile shape, then the VAR digit")
    a = ap.parse_args()
    import torch
    from attentionshift_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    variants = a.variants.split(",")

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3

    for name, M, N, K, act in SHAPES:
        if a.only and name not in a.only.split(","):
            continue
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        b = torch.rand(N, device="cuda", generator=g) * 2 - 1
        ref = torch.nn.functional.linear(x.float(), w.float(), b)
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        elif act == 4:
            ref = torch.relu(ref)
        scale = float(ref.abs().max())
        outs, calls = {}, {}
        for v in variants:
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)

            def call(v=v, out=out):
                os.environ["AS_GEMM_PP"] = v[0]
                os.environ["AS_GEMM_PP_SK"] = v[1:] or "1"
                rc = lib.as_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 1, act, st)
                assert rc == 0, (rc, lib.as_last_error())
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            outs[v], calls[v] = out, call
        if a.lib:
            calls["lib"] = lambda: torch.nn.functional.linear(x, w)
            for _ in range(3):
                calls["lib"]()
        times = {v: [] for v in calls}
        for _ in range(a.rounds):
            for v in calls:
                times[v].append(timed(calls[v]))
        for v in calls:
            rec = dict(shape=name, M=M, N=N, K=K, act=act, variant=v, us_med=round(statistics.median(times[v]), 1),
                       us_min=round(min(times[v]), 1), tflops=round(2.0 * M * N * K / min(times[v]) / 1e6))
            if v in outs:
                o = outs[v].float()
                rec["err"] = round(float((o - ref).abs().max()) / scale, 5)
                rec["nan"] = int(torch.isnan(o).sum())
                if v != "0" and "0" in outs:
                    d = (o - outs["0"].float()).abs()
                    rec["vs_old_max"] = round(float(d.max()) / scale, 5)
                    rec["vs_old_frac_diff"] = round(float((d > 0).float().mean()), 5)
                # run-to-run bitwise reproducibility
                first = outs[v].clone()
                calls[v]()
                torch.cuda.synchronize()
                rec["bitwise_repro"] = bool(torch.equal(first, outs[v]))
            print(json.dumps(rec), flush=True)

    if a.qkv:
        for (B, N, D, h) in [(2, 4197, 768, 12), (1, 6501, 1024, 16), (3, 333, 256, 4)]:
            Npad = lib.as_npad(N)
            g = torch.Generator(device="cuda").manual_seed(B * N + D)
            x = (torch.rand(B, N, D, device="cuda", generator=g) * 2 - 1).bfloat16()
            w = ((torch.rand(3 * D, D, device="cuda", generator=g) * 2 - 1) * 0.05).bfloat16()
            b = torch.rand(3 * D, device="cuda", generator=g) * 2 - 1
            res, calls = {}, {}
            for v in ("0", "b0", "b1"):
                q = torch.zeros(B, h, Npad, 64, device="cuda", dtype=torch.bfloat16)
                k = torch.zeros_like(q)
                vt = torch.zeros(B, h, 64, Npad, device="cuda", dtype=torch.bfloat16)

                def call(v=v, q=q, k=k, vt=vt):
                    os.environ["AS_GEMM_PP"] = v[0]
                    os.environ["AS_GEMM_PP_SK"] = v[1:] or "1"
                    rc = lib.as_qkv_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, N, D, h, 1, st)
                    assert rc == 0, (rc, lib.as_last_error())
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                res[v], calls[v] = (q, k, vt), call
            times = {v: [] for v in calls}
            for _ in range(a.rounds):
                for v in calls:
                    times[v].append(timed(calls[v]))
            # fp32 reference of k / v (plain layouts); q is compared between the two kernels (same fragment-major layout)
            y = torch.nn.functional.linear(x.float(), w.float(), b).reshape(B, N, 3, h, 64)
            kref = y[:, :, 1].permute(0, 2, 1, 3)
            vref = y[:, :, 2].permute(0, 2, 3, 1)
            rec = dict(shape=f"qkv B{B} N{N} D{D}", us_old=round(min(times["0"]), 1), us_pp0=round(min(times["b0"]), 1),
                       us_pp1=round(min(times["b1"]), 1))
            for v in ("0", "b0", "b1"):
                q, k, vt = res[v]
                rec[f"k_err_{v}"] = round(float((k[:, :, :N].float() - kref).abs().max() / kref.abs().max()), 5)
                rec[f"vt_err_{v}"] = round(float((vt[:, :, :, :N].float() - vref).abs().max() / vref.abs().max()), 5)
            rec["b0_eq_b1"] = all(bool(torch.equal(x0, x1)) for x0, x1 in zip(res["b0"], res["b1"]))
            dq = (res["0"][0].float() - res["b1"][0].float()).abs()
            rec["q_max_diff_vs_old"] = round(float(dq.max() / res["0"][0].float().abs().max()), 5)
            rec["q_frac_diff"] = round(float((dq > 0).float().mean()), 5)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
