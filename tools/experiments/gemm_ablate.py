"""Timing ablation of the bf16 LDS-DMA GEMM (csrc/gemm.hip, AS_GEMM_ABLATE hooks): where the K loop's time goes.
Builds one library per variant (results of variants != 0 are WRONG by construction) and times as_linear_fwd.

    python tools/experiments/gemm_ablate.py build      (build container or GPU box: compiles the variants)
    python tools/experiments/gemm_ablate.py run        (GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
VARIANTS = {0: "baseline", 1: "no LDS-DMA / vmcnt in the loop", 2: "no MFMAs", 3: "no fragment reads", 4: "no barrier"}
SHAPES = [(4096, 4096, 4096), (8394, 3072, 768), (8394, 768, 3072)]


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("gemm.hip")]
    for v in VARIANTS:
        o = os.path.join(OUT, f"gemm_v{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DAS_GEMM_ABLATE={v}",
                               "-c", os.path.join(CS, "gemm.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libgemm_v{v}.so"), o] + objs)
        print("built variant", v, flush=True)


def run():
    import torch
    for (M, N, K) in SHAPES:
        x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
        w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for v, name in VARIANTS.items():
            lib = ctypes.CDLL(os.path.join(OUT, f"libgemm_v{v}.so"))
            lib.as_linear_fwd.restype = ctypes.c_int
            lib.as_linear_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: lib.as_linear_fwd(x.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, 1, 0, st)
            for _ in range(5):
                assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print(f"{M}x{N}x{K} variant {v} ({name:32s}): {ms * 1e3:7.1f} us  ({2.0 * M * N * K / ms / 1e9:6.0f} TFLOP/s-equivalent)", flush=True)


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
