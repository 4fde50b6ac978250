"""as_linear_bwd / as_attn_bwd with the transpose-free weight gradient (default) -- run once more with AS_BWD_TRANSPOSED=1 for
the round-3 route.  python tools/experiments/dw_tn_bench.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attentionshift_amd import ops


def timeit(fn, reps=20):
    for _ in range(3):
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
    tag = "transposed" if os.environ.get("AS_BWD_TRANSPOSED") else "tn"
    g = torch.Generator().manual_seed(0)
    for (M, N, K, wf32) in ((8394, 3072, 768, False), (8394, 768, 3072, False), (8394, 768, 768, False), (51200, 1024, 256, True),
                            (51200, 256, 1024, True)):
        x = torch.randn(M, K, generator=g).cuda().bfloat16()
        w = torch.randn(N, K, generator=g).cuda().bfloat16()
        dy = torch.randn(M, N, generator=g).cuda().bfloat16()
        dt = torch.float32 if wf32 else torch.bfloat16
        t_all = timeit(lambda: ops.linear_bwd(x, w, dy, True, True, True, dw_dtype=dt))
        t_dx = timeit(lambda: ops.linear_bwd(x, w, dy, True, False, False, dw_dtype=dt))
        print(json.dumps(dict(route=tag, M=M, N=N, K=K, us_bwd=round(t_all, 1), us_dx_only=round(t_dx, 1), us_dw_db=round(t_all - t_dx, 1))), flush=True)
    B, N, D, h = 2, 4197, 768, 12
    xa = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    wq = (torch.randn(3 * D, D, generator=g) * 0.03).cuda().bfloat16()
    wp = (torch.randn(D, D, generator=g) * 0.03).cuda().bfloat16()
    bq, bp = torch.zeros(3 * D).cuda(), torch.zeros(D).cuda()
    out, st = ops.attention_fwd(xa, wq, bq, wp, bp, h, keep_state=True, keep_o=True)
    dout = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    t = timeit(lambda: ops.attention_bwd(xa, wq, wp, dout, st, want_bias=(True, True)), reps=10)
    print(json.dumps(dict(route=tag, kernel="attention_bwd(module)", us=round(t, 1))), flush=True)


if __name__ == "__main__":
    main()
