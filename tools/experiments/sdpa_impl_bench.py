"""A/B of the attention-forward kernels behind as_sdpa_fwd on the GPU box: max error against an fp32 torch reference on
the same bf16 operands, agreement of lse, and interleaved timing rounds (median / min).

A variant is IMPL[@BUILD]: IMPL = AS_SDPA_IMPL (0 = sdpa_fwd_glds_kernel; sdpa_fwd_pipe_kernel: 1 = <NQ 1, MODE 0>,
2 = <2, 0>, 3 = <1, 1>, 4 = <2, 1>), BUILD = a library under tools/experiments/_build/ made by `build` with extra -D flags.

    python tools/experiments/sdpa_impl_bench.py build nosgb=-DAS_SDPA_SGB=0 ...      (compiles libsdpa_<name>.so; no GPU needed)
    python tools/experiments/sdpa_impl_bench.py [--variants 0,3,4,4@nosgb] [--rounds 7] [--shapes 2x12x4197,1x16x6501]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
sys.path.insert(0, ROOT)


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("sdpa.hip")]
    for spec in specs:
        name, flags = spec.split("=", 1)
        o = os.path.join(OUT, f"sdpa_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + flags.split() +
                              ["-c", os.path.join(CS, "sdpa.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libsdpa_{name}.so"), o] + objs)
        print("built", name, flags, flush=True)


def reference(ops, q_rows, k, vt, N):
    import torch
    q, kk, v = q_rows[:, :, :N].float(), k[:, :, :N].float(), vt[:, :, :, :N].float().transpose(-1, -2)
    s = (q @ kk.transpose(-1, -2)) * ops.LN2             # q is stored pre-scaled by log2(e) / 8
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ v                       # [B,h,N,64]
    return o.permute(0, 2, 1, 3).reshape(q.shape[0], N, -1), lse


class Variant:
    def __init__(self, spec, ops):
        self.spec = spec
        impl, _, bld = spec.partition("@")
        impl, *envs = impl.split(":")                  # "6:AS_SDPA_SK_DEBUG=1": extra environment for this variant's calls
        self.env = dict(e.split("=", 1) for e in envs)
        self.impl = impl
        from attentionshift_amd import _lib
        path = os.path.join(OUT, f"libsdpa_{bld}.so") if bld else _lib.LIB_PATH
        self.lib = ctypes.CDLL(path)
        self.lib.as_sdpa_fwd.restype = ctypes.c_int
        self.lib.as_sdpa_fwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        self.lib.as_sdpa_fwd_workspace_bytes.restype = ctypes.c_size_t
        self.lib.as_sdpa_fwd_workspace_bytes.argtypes = [ctypes.c_int] * 4

    def prepare(self, q, k, vt, N):
        import torch
        B, h = q.shape[0], q.shape[1]
        self.o = torch.empty(B, N, h * 64, device="cuda", dtype=torch.bfloat16)
        self.lse = torch.empty(B, h, N, device="cuda", dtype=torch.float32)
        nws = self.lib.as_sdpa_fwd_workspace_bytes(B, N, h, 1)
        nws = max(nws, 96 << 20)                        # room for every AS_SDPA_SK_GRID a variant may force
        self.ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
        st = torch.cuda.current_stream().cuda_stream
        args = (q.data_ptr(), k.data_ptr(), vt.data_ptr(), self.o.data_ptr(), self.lse.data_ptr(), self.ws.data_ptr(), nws, B, N, h, 1, st)

        self.ws.zero_()

        def call():
            os.environ["AS_SDPA_IMPL"] = self.impl
            for k_ in ("AS_SDPA_SK_DEBUG", "AS_SDPA_SK_GRID"):
                os.environ.pop(k_, None)
            os.environ.update(self.env)
            rc = self.lib.as_sdpa_fwd(*args)
            assert rc == 0, rc
        self.call = call


def main():
    if sys.argv[1:2] == ["build"]:
        return build(sys.argv[2:])
    import torch
    from attentionshift_amd import ops
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,3,4")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", default="2x12x4197")
    ap.add_argument("--scale", type=float, default=0.06)
    a = ap.parse_args()
    variants = [Variant(v, ops) for v in a.variants.split(",")]
    for shp in a.shapes.split(","):
        B, h, N = (int(v) for v in shp.split("x"))
        D = 64 * h
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, N, D, generator=g).cuda().bfloat16()
        w = (torch.randn(3 * D, D, generator=g) * a.scale).cuda().bfloat16()
        bias = (torch.randn(3 * D, generator=g) * 0.1).cuda()
        q, k, vt = ops.qkv_fwd(x, w, bias, h)
        if os.environ.get("SDPA_ZEROS"):                     # power / clock probe: same instruction stream, operands all zero
            q.zero_(); k.zero_(); vt.zero_()
        ref_o, ref_lse = reference(ops, ops.q_from_fragment_major(q), k, vt, N)
        flops = 4.0 * B * h * N * N * 64
        for v in variants:
            v.prepare(q, k, vt, N)
            v.call()
            torch.cuda.synchronize()
            err = (v.o.float() - ref_o).abs().max().item() / max(ref_o.abs().max().item(), 1e-30)
            lerr = (v.lse - ref_lse).abs().max().item()
            print(f"[{shp}] {v.spec:10s}: max err / range {err:.3e}   lse max abs err {lerr:.3e}   finite {bool(torch.isfinite(v.o.float()).all())}", flush=True)
        times = {v.spec: [] for v in variants}
        for r in range(a.rounds):
            for v in variants:
                for _ in range(3):
                    v.call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    v.call()
                e1.record()
                torch.cuda.synchronize()
                times[v.spec].append(e0.elapsed_time(e1) / a.reps)
        # stale-data check AFTER the timing loop (caches warm, workspace full of the previous inputs' partials): new operands
        # written into the SAME tensors, one call per variant, against a fresh reference
        g2 = torch.Generator().manual_seed(1)
        x2 = torch.randn(B, N, D, generator=g2).cuda().bfloat16()
        q2, k2, vt2 = ops.qkv_fwd(x2, w, bias, h)
        q.copy_(q2); k.copy_(k2); vt.copy_(vt2)
        ref_o, ref_lse = reference(ops, ops.q_from_fragment_major(q), k, vt, N)
        for v in variants:
            for _ in range(3):
                v.call()
            torch.cuda.synchronize()
            err = (v.o.float() - ref_o).abs().max().item() / max(ref_o.abs().max().item(), 1e-30)
            print(f"[{shp}] {v.spec:10s}: after new operands: max err / range {err:.3e}  lse err {(v.lse - ref_lse).abs().max().item():.3e}", flush=True)
        for v in variants:
            t = sorted(times[v.spec])
            med, mn = t[len(t) // 2], t[0]
            print(f"[{shp}] {v.spec:10s}: median {med * 1e3:7.1f} us ({flops / med / 1e9:6.0f} TFLOP/s)   min {mn * 1e3:7.1f} us "
                  f"({flops / mn / 1e9:6.0f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
