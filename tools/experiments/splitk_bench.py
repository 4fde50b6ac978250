"""Split-K weight-gradient GEMM (as_linear_splitk_fwd) on the shapes of a training step: time per call and TFLOP/s.
AS_SPLITK_SHORT=1 forces the 128 x 128 tile.    python tools/experiments/splitk_bench.py     (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from attentionshift_amd import ops  # noqa: E402

SHAPES = [("dW_qkv", 2304, 768, 8448), ("dW_proj", 768, 768, 8448), ("dW_fc1", 3072, 768, 8448), ("dW_fc2", 768, 3072, 8448),
          ("head fc1", 1024, 256, 51200), ("head qkv", 768, 256, 51200), ("head proj", 256, 256, 51200),
          ("head embed", 256, 768, 50176)]
torch.manual_seed(0)
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(3):
        ops.linear_splitk(x, w, torch.float32)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        ops.linear_splitk(x, w, torch.float32)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{name:10s} [{M} x {N}] K={K}: {ms * 1e3:7.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
