"""A/B of compile-time variants of csrc/gemm.hip (tile shape, ring depth, ...) on as_linear_fwd, with a refcheck.

    python tools/experiments/gemm_variant_bench.py build s3=-DAS_GEMM_WIDE_STAGES=3 s5=-DAS_GEMM_WIDE_STAGES=5   (here or on the box)
    python tools/experiments/gemm_variant_bench.py run --variants base,base@wide,s5@wide [--shapes MxNxK,...]     (GPU box)
`name@wide` / `name@wide2` run the variant with AS_GEMM_WIDE=1 / 2 (the 256 x 256 tile on 16 waves of 64 x 64 / 8 waves of 128 x 64); `base` is the in-tree library.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
SHAPES = [(8394, 2304, 768), (8394, 768, 768), (8394, 3072, 768), (8394, 768, 3072), (4096, 4096, 4096), (8394, 4096, 1024),
          (8394, 1024, 4096)]


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("gemm.hip")]
    for spec in specs:
        name, _, flags = spec.partition("=")
        o = os.path.join(OUT, f"gemmvar_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                               os.path.join(CS, "gemm.hip"), "-o", o] + flags.split())
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libgemmvar_{name}.so"), o] + objs)
        print("built", name, flags, flush=True)


def run_one(name, shapes, act):
    import torch
    path = os.path.join(ROOT, "attentionshift_amd", "libattnshift_hip.so") if name == "base" else os.path.join(OUT, f"libgemmvar_{name}.so")
    lib = ctypes.CDLL(path)
    lib.as_linear_fwd.restype = ctypes.c_int
    lib.as_linear_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in shapes:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        b = torch.rand(N, device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        call = lambda: lib.as_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 1, act, st)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        ref = torch.nn.functional.linear(x.float(), w.float(), b)
        if act:
            ref = torch.nn.functional.gelu(ref)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(json.dumps(dict(variant=name + {"1": "@wide", "2": "@wide2"}.get(os.environ.get("AS_GEMM_WIDE"), ""), M=M, N=N, K=K,
                              us=round(ms * 1e3, 1), tflops=round(2.0 * M * N * K / ms / 1e9), err=round(err, 5))), flush=True)
        assert err < 2e-2, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run", "_one"])
    ap.add_argument("specs", nargs="*")
    ap.add_argument("--variants", default="base")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--act", type=int, default=0)
    a = ap.parse_args()
    shapes = [tuple(int(v) for v in s.split("x")) for s in a.shapes.split(",")] if a.shapes else SHAPES
    if a.cmd == "build":
        build(a.specs)
    elif a.cmd == "_one":
        run_one(a.variants, shapes, a.act)
    else:
        for v in a.variants.split(","):
            name, _, mode = v.partition("@")
            env = dict(os.environ, AS_GEMM_WIDE={"wide": "1", "wide2": "2"}.get(mode, "0"))
            subprocess.call([sys.executable, os.path.abspath(__file__), "_one", "--variants", name, "--act", str(a.act)]
                            + (["--shapes", a.shapes] if a.shapes else []), env=env)


if __name__ == "__main__":
    main()
