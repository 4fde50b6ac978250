"""Forward + backward of one trainable Swin block (bf16) at the four Swin-B stage sizes, 1024^2 input, batch 2; run a second
time with AS_WINDOW_BWD_VALU=1 for the fp32-arithmetic window-attention backward.   python tools/experiments/swin_block_bwd_bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import torch
    sys.path.insert(0, ROOT)
    from attentionshift_amd.swin import SwinTransformerBlock
    for (hw, C, heads) in ((256, 128, 4), (128, 256, 8), (64, 512, 16), (32, 1024, 32)):
        for shift in (0, 3):
            blk = SwinTransformerBlock(C, (hw, hw), heads, window_size=7, shift_size=shift, compute_dtype=torch.bfloat16).cuda().train()
            x = torch.randn(2, hw * hw, C, device="cuda", requires_grad=True)
            w = torch.randn(2, hw * hw, C, device="cuda")

            def step():
                y, _ = blk(x)
                (y * w).sum().backward()
                x.grad = None
                for p in blk.parameters():
                    p.grad = None
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record()
            torch.cuda.synchronize()
            print(f"{'valu' if os.environ.get('AS_WINDOW_BWD_VALU') else 'mfma'}  tokens {hw}x{hw} C {C:4d} shift {shift}: "
                  f"fwd+bwd {e0.elapsed_time(e1) / 10:7.3f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one()
    else:
        for env in ({}, {"AS_WINDOW_BWD_VALU": "1"}):
            subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, **env))
