"""Timing ablation of the roll-out step kernel (csrc/rollout.hip, AS_ROLLOUT_ABLATE hooks).  Variants != 0 give WRONG
results by construction.  Shape: BASELINE config 2 (B=2, h=12, N=4197, 101 roll-out rows).

    python tools/experiments/rollout_ablate.py build
    python tools/experiments/rollout_ablate.py run        (GPU box)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
VARIANTS = {0: "baseline", 11: "step4: no LDS-DMA / vmcnt in the loop", 12: "step4: no exp2", 13: "step4: no q.k MFMAs",
            14: "step4: no barrier", 15: "step4: no R.Pbar MFMAs"}     # (1-4: the same hooks in rollout_step3, AS_ROLLOUT_V3=1)


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj")) if f.endswith(".o") and not f.startswith("rollout.hip")]
    for v in VARIANTS:
        o = os.path.join(OUT, f"rollout_v{v}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DAS_ROLLOUT_ABLATE={v}",
                               "-c", os.path.join(CS, "rollout.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"librollout_v{v}.so"), o] + objs)
        print("built variant", v, flush=True)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from attentionshift_amd import _lib, ops
    B, N, D, h, T = 2, 4197, 768, 12, 101
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    states = []
    for l in range(3):
        wq = (torch.randn(3 * D, D, generator=g) * 0.06).cuda().bfloat16()
        wp = (torch.randn(D, D, generator=g) * 0.03).cuda().bfloat16()
        _, st = ops.attention_fwd(x, wq, torch.zeros(3 * D, device="cuda"), wp, torch.zeros(D, device="cuda"), h, keep_state=True)
        states.append(st)
    real = _lib.load
    for v, name in VARIANTS.items():
        lib = ctypes.CDLL(os.path.join(OUT, f"librollout_v{v}.so"))
        for sym, (res, args) in _lib.SIGNATURES.items():
            if hasattr(lib, sym):
                getattr(lib, sym).restype, getattr(lib, sym).argtypes = res, args
        _lib.load = lambda lib=lib: lib
        try:
            for _ in range(3):
                ops.rollout_rows(states, T - 1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.rollout_rows(states, T - 1)
            e1.record()
            torch.cuda.synchronize()
            print(f"variant {v} ({name:34s}): {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per 3-layer roll-out (2 step launches)", flush=True)
        finally:
            _lib.load = real


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
