"""A/B of as_qkv_fwd (config 2: B = 2, N = 4197, D = 768, h = 12) between the in-tree library and variant builds under
tools/experiments/_build/ (gemm_variant_bench.py build name=-Dflags).   python tools/experiments/qkv_ab.py base vtdirect ..."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
B, N, D, h = 2, 4197, 768, 12
Npad = -(-N // 64) * 64
x = (torch.rand(B * N, D, device="cuda") * 2 - 1).bfloat16()
w = (torch.rand(3 * D, D, device="cuda") * 2 - 1).bfloat16()
bias = torch.rand(3 * D, device="cuda")
names = sys.argv[1:] or ["base"]
libs, bufs = {}, {}
for name in names:
    path = os.path.join(ROOT, "attentionshift_amd", "libattnshift_hip.so") if name == "base" else os.path.join(OUT, f"libgemmvar_{name}.so")
    lib = ctypes.CDLL(path)
    lib.as_qkv_fwd.restype = ctypes.c_int
    lib.as_qkv_fwd.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    libs[name] = lib
    bufs[name] = [torch.zeros(B * h * Npad * 64, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
st = torch.cuda.current_stream().cuda_stream
for rnd in range(3):
    for name in names:
        q, k, vt = bufs[name]
        call = lambda: libs[name].as_qkv_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr(), B, N, D, h, 1, st)
        for _ in range(5):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd} {name:10s}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
ref = bufs[names[0]]
for name in names[1:]:
    print(name, "equal to", names[0], [bool(torch.equal(a, b)) for a, b in zip(ref, bufs[name])])
