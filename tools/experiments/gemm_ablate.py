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
VARIANTS = {0: "baseline", 1: "no LDS-DMA / vmcnt in the loop", 2: "no MFMAs", 3: "no fragment reads", 4: "no barrier",
            5: "QKV only: no V^T tile stores", 6: "staged epilogue without the global stores", 7: "no epilogue",
            8: "epilogue without conversion + staging writes", 9: "epilogue without copy-out + stores"}
SHAPES = [(4096, 4096, 4096), (8394, 3072, 768), (8394, 3072, 1536), (8394, 3072, 3072), (8394, 768, 3072)]
if os.environ.get("GEMM_ABLATE_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["GEMM_ABLATE_SHAPES"].split(",")]
if os.environ.get("GEMM_ABLATE_VARIANTS"):
    VARIANTS = {int(v): VARIANTS[int(v)] for v in os.environ["GEMM_ABLATE_VARIANTS"].split(",")}


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


def run_qkv():
    """as_qkv_fwd at config 2 (B=2, N=4197, D=768, h=12) for variants 0 and 5: what the V^T epilogue costs."""
    import torch
    B, N, D, h = 2, 4197, 768, 12
    Npad = -(-N // 64) * 64
    x = (torch.rand(B * N, D, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(3 * D, D, device="cuda") * 2 - 1).bfloat16()
    q = torch.empty(B * h * Npad * 64, device="cuda", dtype=torch.bfloat16)
    k, vt = torch.empty_like(q), torch.empty_like(q)
    for v in (0, 5):
        lib = ctypes.CDLL(os.path.join(OUT, f"libgemm_v{v}.so"))
        lib.as_qkv_fwd.restype = ctypes.c_int
        lib.as_qkv_fwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.as_qkv_fwd(x.data_ptr(), w.data_ptr(), None, q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, N, D, h, 1, st)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"as_qkv_fwd variant {v} ({VARIANTS[v]}): {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    {"build": build, "qkv": run_qkv}.get(sys.argv[1] if len(sys.argv) > 1 else "", run)()
