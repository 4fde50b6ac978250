"""A/B of compile-time variants of csrc/gemm.hip (tile shape, ring depth, ...) on as_linear_fwd, with a refcheck.

    python tools/experiments/gemm_variant_bench.py build s3=-DAS_GEMM_K64_STAGES=3 ...                      (here or on the box)
    python tools/experiments/gemm_variant_bench.py run --variants base,base@wide64,s3@tall64 [--shapes MxNxK,...]  (GPU box)
`name@tile` runs the variant with AS_GEMM_TILE=tile (short | tall | tall64 | wide64 forced, csrc/gemm.hip launch_gemm_glds);
`base` is the in-tree library.  GEMM_ZEROS=1: zero-filled operands (power / clock probe), GEMM_LIB=1: also time F.linear.
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
        if os.environ.get("GEMM_ZEROS"):                      # power / DVFS probe: same instruction stream, no toggling
            x.zero_(); w.zero_()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        call = lambda: lib.as_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 1, act, st)
        if os.environ.get("GEMM_SK"):                         # the stream-K entry point (falls back to the plain grid by itself)
            lib.as_linear_sk_workspace_bytes.restype = ctypes.c_size_t
            lib.as_linear_sk_workspace_bytes.argtypes = [ctypes.c_int] * 3
            lib.as_linear_sk_fwd.restype = ctypes.c_int
            lib.as_linear_sk_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            nb = lib.as_linear_sk_workspace_bytes(M, N, K)
            wsb = torch.empty(max(nb, 16), device="cuda", dtype=torch.uint8)
            call = lambda: lib.as_linear_sk_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 1, act,
                                                wsb.data_ptr(), wsb.numel(), st)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        ref = torch.nn.functional.linear(x.float(), w.float(), b)
        if act:
            ref = torch.nn.functional.gelu(ref)
        err = float((out.float() - ref).abs().max() / ref.abs().max().clamp(min=1e-6))
        lib_tf = None
        if os.environ.get("GEMM_LIB"):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                torch.nn.functional.linear(x, w)
            t0.record()
            for _ in range(50):
                torch.nn.functional.linear(x, w)
            t1.record()
            torch.cuda.synchronize()
            lib_tf = round(2.0 * M * N * K / (t0.elapsed_time(t1) / 50) / 1e9)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(json.dumps(dict(variant=name + "@" + os.environ.get("AS_GEMM_TILE", "auto") + "+" + os.environ.get("AS_GEMM_STAGGER", "0") + ("/sk" if os.environ.get("GEMM_SK") else ""), M=M, N=N, K=K,
                              us=round(ms * 1e3, 1), tflops=round(2.0 * M * N * K / ms / 1e9), err=round(err, 5), lib_tflops=lib_tf)), flush=True)
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
            mode, _, stag = mode.partition("+")              # name@tile+N: AS_GEMM_STAGGER=N (phase offset of the second workgroup per CU)
            env = dict(os.environ)
            env.pop("AS_GEMM_TILE", None)
            env.pop("AS_GEMM_STAGGER", None)
            if stag:
                env["AS_GEMM_STAGGER"] = stag
            if mode:
                env["AS_GEMM_TILE"] = mode
            subprocess.call([sys.executable, os.path.abspath(__file__), "_one", "--variants", name, "--act", str(a.act)]
                            + (["--shapes", a.shapes] if a.shapes else []), env=env)


if __name__ == "__main__":
    main()
