"""The MLP's GELU inside the GEMM epilogues (autograd.MlpFn's kernels) against the separate ATen passes, ViT-B layer shape:
    python tools/experiments/mlp_fused_bench.py [M D H]          (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attentionshift_amd import ops  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    M, D, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8394, 768, 3072)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, D, generator=g).cuda().bfloat16()
    w1 = (torch.randn(H, D, generator=g) / D ** 0.5).cuda().bfloat16()
    b1 = torch.zeros(H).cuda()
    w2 = (torch.randn(D, H, generator=g) / H ** 0.5).cuda().bfloat16()
    dy = torch.randn(M, D, generator=g).cuda().bfloat16()
    a, pre = ops.linear_gelu(x, w1, b1)
    rows = [
        ("fc1 (as_linear_fwd)", lambda: ops.linear(x, w1, b1)),
        ("ATen gelu", lambda: torch.nn.functional.gelu(pre)),
        ("fc1 + GELU, pre kept (as_linear_gelu_fwd)", lambda: ops.linear_gelu(x, w1, b1)),
        ("fc1 + GELU, inference epilogue (act = 1)", lambda: ops.linear(x, w1, b1, act="gelu")),
        ("fc2 backward (as_linear_bwd: dx, dW, db)", lambda: ops.linear_bwd(a, w2, dy, True, True, True, dw_dtype=torch.float32)),
        ("ATen gelu_backward", lambda: torch.ops.aten.gelu_backward(pre, pre)),
        ("fc2 backward with GELU' (as_linear_bwd_dgelu)", lambda: ops.linear_bwd(a, w2, dy, True, True, True, dw_dtype=torch.float32, gelu_pre=pre)),
    ]
    for name, fn in rows:
        print(f"{name:50s} {timeit(fn):8.1f} us", flush=True)


if __name__ == "__main__":
    main()
