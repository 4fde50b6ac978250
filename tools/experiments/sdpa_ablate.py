"""Timing ablation of the bf16 flash-attention forward (csrc/sdpa.hip, AS_SDPA_ABLATE hooks): which part of the tile
loop the time goes to.  Builds one library per variant (results of variants != 0 are WRONG by construction) and times
as_sdpa_fwd at BASELINE config-2 shape (B=2, h=12, N=4197).

    python tools/experiments/sdpa_ablate.py build      (build container or GPU box: compiles the variants)
    python tools/experiments/sdpa_ablate.py run        (GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
VARIANTS = {0: "baseline", 1: "exp2 -> multiply", 2: "no softmax VALU", 3: "no P.V MFMAs", 4: "no Q.K MFMAs",
            5: "no LDS-DMA in the loop", 6: "no per-tile barrier"}
if os.environ.get("SDPA_ABLATE_PIPE"):        # the shipped kernel (sdpa_fwd_pipe_kernel<2, 0, 1>): hooks 11 .. 17
    VARIANTS = {0: "baseline", 11: "exp2 -> multiply", 12: "no softmax VALU", 13: "no P.V MFMAs", 14: "no Q.K MFMAs",
                15: "no LDS-DMA in the loop", 16: "no per-tile barrier", 17: "no LDS fragment reads"}


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("sdpa.hip")]
    for v in VARIANTS:
        o = os.path.join(OUT, f"sdpa_v{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DAS_SDPA_ABLATE={v}",
                               "-c", os.path.join(CS, "sdpa.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libablate_v{v}.so"), o] + objs)
        print("built variant", v, flush=True)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from attentionshift_amd import ops
    B, N, D, h = 2, 4197, 768, 12
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    w = (torch.randn(3 * D, D, generator=g) * 0.06).cuda().bfloat16()
    q, k, vt = ops.qkv_fwd(x, w, torch.zeros(3 * D, device="cuda"), h)
    o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, h, N, device="cuda", dtype=torch.float32)
    flops = 4.0 * B * h * N * N * 64
    for v, name in VARIANTS.items():
        lib = ctypes.CDLL(os.path.join(OUT, f"libablate_v{v}.so"))
        lib.as_sdpa_fwd.restype = ctypes.c_int
        lib.as_sdpa_fwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        lib.as_sdpa_fwd_workspace_bytes.restype = ctypes.c_size_t
        lib.as_sdpa_fwd_workspace_bytes.argtypes = [ctypes.c_int] * 4
        nws = lib.as_sdpa_fwd_workspace_bytes(B, N, h, 1)
        ws = torch.empty(max(nws, 1), device="cuda", dtype=torch.uint8)
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.as_sdpa_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), lse.data_ptr(), ws.data_ptr(),
                                       nws, B, N, h, 1, st)
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
        print(f"variant {v} ({name:24s}): {ms * 1e3:7.1f} us  ({flops / ms / 1e9:6.0f} TFLOP/s-equivalent)", flush=True)


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
