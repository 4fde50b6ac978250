"""Run torch.nn.functional.linear (hipBLASLt) on the backbone's four GEMM shapes so that a rocprofv3 kernel trace of this
script names the library's chosen solutions (macro tile, stream-K / GSU, wave tiling are part of the Cijk_* kernel name).

    PROF_NAME_WIDTH=400 tools/prof_cmd.sh r06_hipblaslt_solutions python tools/experiments/hipblaslt_names.py
"""
import torch

SHAPES = [("qkv", 8394, 2304, 768), ("proj", 8394, 768, 768), ("fc1", 8394, 3072, 768), ("fc2", 8394, 768, 3072)]
for name, M, N, K in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.rand(N, device="cuda").bfloat16()
    for _ in range(20):
        torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    # a marker kernel between the shapes: the summary groups by kernel name, the order of first appearance is the shape order
    torch.zeros(1, device="cuda").add_(1.0)
print("done")
