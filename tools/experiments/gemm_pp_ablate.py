"""Timing ablations / compile-time variants of csrc/gemm_pp.hip (results of an ablated build are WRONG; timing only).

    python tools/experiments/gemm_pp_ablate.py build base= abl1=-DAS_PP_ABLATE=1 ...      (here: hipcc cross-compiles)
    python tools/experiments/gemm_pp_ablate.py run --variants base,abl1 --cfgs a0,b0      (GPU box; interleaved rounds)
"""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
SHAPES = {"sq4096": (4096, 4096, 4096, 0), "fc1": (8394, 3072, 768, 0), "fc1_gelu": (8394, 3072, 768, 1), "fc2": (8394, 768, 3072, 0),
          "proj": (8394, 768, 768, 0), "qkv": (8394, 2304, 768, 0)}


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("gemm_pp.hip")]
    for spec in specs:
        name, _, flags = spec.partition("=")
        o = os.path.join(OUT, f"ppvar_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                               os.path.join(CS, "gemm_pp.hip"), "-o", o] + flags.split())
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libppvar_{name}.so"), o] + objs)
        ctypes.CDLL(os.path.join(OUT, f"libppvar_{name}.so"))             # (unresolved symbols show up here, not on the GPU box)
        print("built", name, flags, flush=True)


def run(variants, cfgs, shapes, rounds, reps):
    os.environ["AS_GEMM_PP_DYN"] = "1"
    import torch
    st = torch.cuda.current_stream().cuda_stream
    libs = {}
    for v in variants:
        lib = ctypes.CDLL(os.path.join(OUT, f"libppvar_{v}.so"))
        lib.as_linear_fwd.restype = ctypes.c_int
        lib.as_linear_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        libs[v] = lib
    for sname in shapes:
        M, N, K, act = SHAPES[sname]
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
        b = torch.rand(N, device="cuda", generator=g) * 2 - 1
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        arms = [(v, c) for v in variants for c in cfgs]

        def call(v, c):
            os.environ["AS_GEMM_PP"] = c[0]
            os.environ["AS_GEMM_PP_VAR"] = c[1:] or "0"
            assert libs[v].as_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 1, act, st) == 0
        times = {arm: [] for arm in arms}
        for arm in arms:
            for _ in range(3):
                call(*arm)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for arm in arms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    call(*arm)
                e1.record()
                torch.cuda.synchronize()
                times[arm].append(e0.elapsed_time(e1) / reps * 1e3)
        for arm in arms:
            print(json.dumps(dict(shape=sname, variant=arm[0], cfg=arm[1], us_min=round(min(times[arm]), 1),
                                  us_med=round(statistics.median(times[arm]), 1),
                                  tflops=round(2.0 * M * N * K / min(times[arm]) / 1e6))), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("specs", nargs="*")
    ap.add_argument("--variants", default="base")
    ap.add_argument("--cfgs", default="a0")
    ap.add_argument("--shapes", default="sq4096,fc1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    if a.cmd == "build":
        build(a.specs)
    else:
        run(a.variants.split(","), a.cfgs.split(","), a.shapes.split(","), a.rounds, a.reps)
