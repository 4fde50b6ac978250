"""A/B of csrc/sdpa.hip variants built as separate libraries (tools/experiments/_build/libsdpa_<name>.so), interleaved rounds in
ONE process; outputs compared bitwise with the first variant's.

    python tools/experiments/sdpa_store_ab.py build wide= narrow=-DAS_SDPA_WIDE_STORE=0        (here: hipcc cross-compiles)
    python tools/experiments/sdpa_store_ab.py run --variants narrow,wide                        (GPU box)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("sdpa.hip")]
    for spec in specs:
        name, _, flags = spec.partition("=")
        o = os.path.join(OUT, f"sdpa_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CS, "sdpa.hip"), "-o", o] + flags.split())
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libsdpa_{name}.so"), o] + objs)
        ctypes.CDLL(os.path.join(OUT, f"libsdpa_{name}.so"))
        print("built", name, flags, flush=True)


def run(variants, rounds, reps):
    import torch
    from attentionshift_amd import ops
    st = torch.cuda.current_stream().cuda_stream
    libs = {}
    for v in variants:
        lib = ctypes.CDLL(os.path.join(OUT, f"libsdpa_{v}.so"))
        lib.as_sdpa_fwd.restype = ctypes.c_int
        lib.as_sdpa_fwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        lib.as_sdpa_fwd_workspace_bytes.restype = ctypes.c_size_t
        lib.as_sdpa_fwd_workspace_bytes.argtypes = [ctypes.c_int] * 4
        libs[v] = lib
    for (B, N, h) in [(2, 4197, 12), (1, 6501, 16), (2, 4096, 12)]:
        D = h * 64
        g = torch.Generator(device="cuda").manual_seed(B * N + h)
        x = ((torch.rand(B, N, D, device="cuda", generator=g) * 2 - 1)).bfloat16()
        w = ((torch.rand(3 * D, D, device="cuda", generator=g) * 2 - 1) * 0.05).bfloat16()
        b = torch.rand(3 * D, device="cuda", generator=g) * 2 - 1
        q, k, vt = ops.qkv_fwd(x, w, b, h)
        outs, calls = {}, {}
        for v in variants:
            o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
            lse = torch.empty(B, h, N, device="cuda", dtype=torch.float32)
            nb = libs[v].as_sdpa_fwd_workspace_bytes(B, N, h, 1)
            ws = torch.empty(max(nb, 1), device="cuda", dtype=torch.uint8)

            def call(v=v, o=o, lse=lse, ws=ws, nb=nb):
                assert libs[v].as_sdpa_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), lse.data_ptr(), ws.data_ptr() if nb else None, nb, B, N, h, 1, st) == 0
            for _ in range(3):
                call()
            outs[v], calls[v] = (o, lse), call
        torch.cuda.synchronize()
        times = {v: [] for v in variants}
        for _ in range(rounds):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    calls[v]()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / reps * 1e3)
        for v in variants:
            same = all(bool(torch.equal(a, c)) for a, c in zip(outs[v], outs[variants[0]]))
            print(json.dumps(dict(shape=f"B{B} N{N} h{h}", variant=v, us_min=round(min(times[v]), 1), us_med=round(sorted(times[v])[len(times[v]) // 2], 1),
                                  tflops=round(4.0 * B * h * N * N * 64 / min(times[v]) / 1e6), equal_first=same)), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("specs", nargs="*")
    ap.add_argument("--variants", default="narrow,wide")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    if a.cmd == "build":
        build(a.specs)
    else:
        run(a.variants.split(","), a.rounds, a.reps)
