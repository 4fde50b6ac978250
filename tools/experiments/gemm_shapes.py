"""Time as_linear_fwd (bf16) on square and ViT/Swin shapes: python tools/experiments/gemm_shapes.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attentionshift_amd import ops

SHAPES = [(7168, 2304, 768), (5376, 3072, 768), (3584, 2304, 768), (8394, 768, 1536), (16384, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192), (8394, 2304, 768), (8394, 768, 768), (8394, 3072, 768), (8394, 768, 3072),
          (8192, 1536, 512), (8192, 2048, 512), (8192, 512, 2048), (131072, 384, 128), (131072, 512, 128), (131072, 128, 512)]


def main():
    dev = torch.device("cuda")
    shapes = SHAPES
    if sys.argv[1:2] == ["sweep"]:
        # time vs M at fixed (N, K): a staircase in rounds of 256 one-per-CU tiles = tile quantisation is the cost;
        # a straight line = the chip runs at a fixed total rate (power / clock cap) and idle CUs are not lost time
        N, K = int(sys.argv[2]), int(sys.argv[3])
        shapes = [(256 * m, N, K) for m in range(int(sys.argv[4]), int(sys.argv[5]) + 1)]
    for (M, N, K) in shapes:
        x = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
        w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
        b = torch.zeros(N, device=dev)
        for _ in range(3):
            ops.linear(x, w, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.linear(x, w, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        ref = torch.nn.functional.linear(x, w)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(n):
            torch.nn.functional.linear(x, w)
        t1.record()
        torch.cuda.synchronize()
        lib = t0.elapsed_time(t1) / n
        err = float((ops.linear(x, w, b).float() - ref.float()).abs().max() / ref.float().abs().max())
        assert err < 2e-2, err
        print(json.dumps(dict(tile=os.environ.get("AS_GEMM_TILE", "auto"), M=M, N=N, K=K, ms=round(ms, 4), tflops=round(2 * M * N * K / ms / 1e9, 1),
                              lib_ms=round(lib, 4), lib_tflops=round(2 * M * N * K / lib / 1e9, 1),
                              gbs=round((M * K + N * K + M * N) * 2 / ms / 1e6, 0))))


if __name__ == "__main__":
    main()
