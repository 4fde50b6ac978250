"""Does running the two images of a batch as two independent kernel sequences (one HIP stream each) beat the batched
sequence?  The batched GEMMs leave a partly filled last round of workgroups (QKV 594 tiles on 512 slots, fc1 396 on 256,
proj / fc2 198 on 256); two independent half-size sequences let one image's tail overlap the other's next kernel.
Times `depth` ViT-B blocks (add+LN, QKV + SDPA + proj, add+LN, fc1+GELU, fc2) on random weights, N = 4197, bf16.

    python tools/experiments/per_image_streams.py [depth]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attentionshift_amd import ops


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    B, N, D, h = 2, 4197, 768, 12
    g = torch.Generator().manual_seed(0)
    cd = torch.bfloat16
    w = dict(qkv=(torch.randn(3 * D, D, generator=g) * 0.03).cuda().to(cd), bqkv=torch.zeros(3 * D).cuda(),
             proj=(torch.randn(D, D, generator=g) * 0.03).cuda().to(cd), bproj=torch.zeros(D).cuda(),
             fc1=(torch.randn(4 * D, D, generator=g) * 0.03).cuda().to(cd), b1=torch.zeros(4 * D).cuda(),
             fc2=(torch.randn(D, 4 * D, generator=g) * 0.02).cuda().to(cd), b2=torch.zeros(D).cuda(),
             g1=torch.ones(D).cuda(), be1=torch.zeros(D).cuda())
    x0 = torch.randn(B, N, D, generator=g).cuda()

    def blocks(x):
        delta = None
        for _ in range(depth):
            x, y = ops.add_layernorm(x, delta, w["g1"], w["be1"], 1e-6, cd)
            a, _st = ops.attention_fwd(y, w["qkv"], w["bqkv"], w["proj"], w["bproj"], h, keep_state=True)
            x, z = ops.add_layernorm(x, a, w["g1"], w["be1"], 1e-6, cd)
            z = ops.linear(z, w["fc1"], w["b1"], act="gelu")
            delta = ops.linear(z, w["fc2"], w["b2"])
        return x

    def batched():
        return blocks(x0)

    streams = [torch.cuda.Stream() for _ in range(B)]

    def per_image(interleave):
        main_s = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main_s)
        if not interleave:                                  # image 0's whole chain queued, then image 1's
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    blocks(x0[i:i + 1])
        else:                                               # layer by layer, alternating streams (what a backbone loop would do)
            xs = [x0[i:i + 1] for i in range(B)]
            deltas = [None] * B
            for _ in range(depth):
                for i, s in enumerate(streams):
                    with torch.cuda.stream(s):
                        x, y = ops.add_layernorm(xs[i], deltas[i], w["g1"], w["be1"], 1e-6, cd)
                        a, _st = ops.attention_fwd(y, w["qkv"], w["bqkv"], w["proj"], w["bproj"], h, keep_state=True)
                        x, z = ops.add_layernorm(x, a, w["g1"], w["be1"], 1e-6, cd)
                        z = ops.linear(z, w["fc1"], w["b1"], act="gelu")
                        deltas[i] = ops.linear(z, w["fc2"], w["b2"])
                        xs[i] = x
        for s in streams:
            main_s.wait_stream(s)

    def timeit(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / reps * 1e3
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, host

    for name, fn in (("batched B=2, one stream", batched), ("per image, chains back to back on 2 streams", lambda: per_image(False)),
                     ("per image, interleaved per layer on 2 streams", lambda: per_image(True)), ("batched B=2, one stream (again)", batched)):
        dev, host = timeit(fn)
        print(f"{name:50s}: device {dev:7.3f} ms   host queueing {host:6.3f} ms   ({depth} blocks)", flush=True)


if __name__ == "__main__":
    main()
