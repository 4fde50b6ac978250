"""as_window_attn_bwd on bf16 tensors: the matrix-core kernel against the fp32-arithmetic one (AS_WINDOW_BWD_VALU=1, read
once per process) on the four stages of Swin-B at 1024^2, batch 2.      python tools/experiments/window_bwd_bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import torch
    sys.path.insert(0, ROOT)
    from attentionshift_amd import ops
    for (hw, h) in ((256, 4), (128, 8), (64, 16), (32, 32)):
        B, C = 2, 32 * h
        g = torch.Generator().manual_seed(hw)
        qkv = torch.randn(B, hw, hw, 3 * C, generator=g).cuda().bfloat16()
        bq = torch.zeros(3 * C).cuda()
        table = (torch.randn(169, h, generator=g) * 0.3).cuda()
        d_out = torch.randn(B, hw, hw, C, generator=g).cuda().bfloat16()
        for shift in (0, 3):
            fn = lambda: ops.window_attention_bwd(qkv, bq, table, d_out, h, 7, shift)   # noqa: E731
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"{os.environ.get('AS_WINDOW_BWD_VALU', 'mfma'):5s} tokens {hw}x{hw} heads {h:2d} shift {shift}: "
                  f"{e0.elapsed_time(e1) / 10 * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for env in ({}, {"AS_WINDOW_BWD_VALU": "1"}):
            subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env))
