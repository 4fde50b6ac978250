import os, sys, json, torch
sys.path.insert(0, os.getcwd())
os.environ["AS_GEMM_PP_DYN"] = "1"
from attentionshift_amd import ops
def t(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K, cfg) in [(8394, 2304, 768, "b"), (8394, 3072, 768, "a"), (8394, 768, 3072, "b"), (8394, 768, 768, "b")]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).bfloat16()
    b = torch.rand(N, device="cuda", generator=g)
    os.environ["AS_GEMM_PP"] = cfg
    res = {"bias": [], "nobias": []}
    for _ in range(3):
        ops.linear(x, w, b); ops.linear(x, w, None)
    for r in range(5):
        res["bias"].append(t(lambda: ops.linear(x, w, b)))
        res["nobias"].append(t(lambda: ops.linear(x, w, None)))
    print(json.dumps(dict(shape=f"{M}x{N}x{K}", cfg=cfg, bias=round(min(res["bias"]), 1), nobias=round(min(res["nobias"]), 1))))
