"""A/B of compile-time variants of csrc/sdpa_bwd.hip on as_sdpa_bwd (config-2 shape), GPU box:
    python tools/experiments/bwd_variant_bench.py build occ2=-DAS_BWD_DQ_OCC=2 nst3=-DAS_BWD_NST=3      (here or on the box)
    python tools/experiments/bwd_variant_bench.py run --variants base,occ2,nst3"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("sdpa_bwd.hip")]
    for spec in specs:
        name, _, flags = spec.partition("=")
        o = os.path.join(OUT, f"bwdvar_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                               os.path.join(CS, "sdpa_bwd.hip"), "-o", o] + flags.split())
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libbwdvar_{name}.so"), o] + objs)
        print("built", name, flags, flush=True)


def run(variants, B=2, N=4197, h=12):
    import torch
    sys.path.insert(0, ROOT)
    from attentionshift_amd import ops
    D = 64 * h
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    w = (torch.randn(3 * D, D, generator=g) * 0.06).cuda().bfloat16()
    q, k, vt = ops.qkv_fwd(x, w, None, h)
    o, lse = ops.sdpa_fwd(q, k, vt, N)
    d_o = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    ref = None
    for name in variants:
        path = os.path.join(ROOT, "attentionshift_amd", "libattnshift_hip.so") if name == "base" else os.path.join(OUT, f"libbwdvar_{name}.so")
        lib = ctypes.CDLL(path)
        lib.as_sdpa_bwd_workspace_bytes.restype = ctypes.c_size_t
        lib.as_sdpa_bwd_workspace_bytes.argtypes = [ctypes.c_int] * 4
        lib.as_sdpa_bwd.restype = ctypes.c_int
        lib.as_sdpa_bwd.argtypes = [ctypes.c_void_p] * 8 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        nb = lib.as_sdpa_bwd_workspace_bytes(B, N, h, 1)
        ws = torch.empty(nb, device="cuda", dtype=torch.uint8)
        dqkv = torch.full((B, N, 3 * D), float("nan"), device="cuda", dtype=torch.bfloat16)   # unwritten rows show
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.as_sdpa_bwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
                                       dqkv.data_ptr(), ws.data_ptr(), nb, B, N, h, 1, st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        if ref is None:
            ref = dqkv.float().clone()
        err = float((dqkv.float() - ref).abs().max() / ref.abs().max())
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ts.sort()
        print(f"{name:10s}: median {ts[2] * 1e3:7.1f} us   min {ts[0] * 1e3:7.1f} us   max |diff| vs first variant / range {err:.2e}", flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build(sys.argv[2:])
    else:
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd")
        ap.add_argument("--variants", default="base")
        ap.add_argument("--shapes", default="2x4197x12", help="comma list of BxNxh")
        a = ap.parse_args()
        for shp in a.shapes.split(","):
            B, N, h = (int(v) for v in shp.split("x"))
            print(f"--- B={B} N={N} h={h}", flush=True)
            run(a.variants.split(","), B, N, h)
