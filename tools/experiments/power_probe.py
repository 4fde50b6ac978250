"""Is a kernel running into the chip's power / clock limit?  Loops one of the hot kernels for a few seconds on a side thread
while the main thread samples `rocm-smi` (socket power, sclk) a few times; prints one JSON line per kernel.

    python tools/experiments/power_probe.py [sdpa|gemm|sdpa_zeros|idle ...]
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from attentionshift_amd import ops


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        power = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((str(v) for k, v in card.items() if k.lower().startswith("sclk")), None)
        m = re.search(r"(\\d+)\\s*Mhz", sclk or "", re.I)
        return power, int(m.group(1)) if m else sclk
    except Exception as e:                                   # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"[:80]


def main():
    names = sys.argv[1:] or ["idle", "sdpa", "sdpa_zeros", "sdpa_bwd", "sdpa_bwd_zeros", "fc2", "gemm"]
    B, h, N, D = 2, 12, 4197, 768
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    w = (torch.randn(3 * D, D, generator=g) * 0.06).cuda().bfloat16()
    bias = (torch.randn(3 * D, generator=g) * 0.1).cuda()
    q, k, vt = ops.qkv_fwd(x, w, bias, h)
    qz, kz, vz = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(vt)
    a = (torch.rand(8192, 4096, device="cuda") * 2 - 1).bfloat16()
    wb = (torch.rand(4096, 4096, device="cuda") * 2 - 1).bfloat16()
    bz = torch.zeros(4096, device="cuda")
    o, lse = ops.sdpa_fwd(q, k, vt, N)
    d_o = torch.randn(B, N, D, generator=g).cuda().bfloat16()
    oz, lz, dz = torch.zeros_like(o), torch.zeros_like(lse), torch.zeros_like(d_o)
    xf = torch.randn(8394, 3072, generator=g).cuda().bfloat16()
    w2 = (torch.randn(768, 3072, generator=g) * 0.03).cuda().bfloat16()
    b2 = torch.zeros(768, device="cuda")
    work = {"sdpa": lambda: ops.sdpa_fwd(q, k, vt, N), "sdpa_zeros": lambda: ops.sdpa_fwd(qz, kz, vz, N),
            "sdpa_bwd": lambda: ops.sdpa_bwd(q, k, vt, o, d_o, lse, N),
            "sdpa_bwd_zeros": lambda: ops.sdpa_bwd(qz, kz, vz, oz, dz, lz, N),
            "fc2": lambda: ops.linear(xf, w2, b2),
            "gemm": lambda: ops.linear(a, wb, bz), "idle": None}
    for name in names:
        fn = work[name]
        stop = threading.Event()
        count = [0]

        def loop():
            torch.cuda.set_device(0)
            while not stop.is_set():
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                count[0] += 50

        th = None
        if fn is not None:
            th = threading.Thread(target=loop)
            th.start()
            time.sleep(1.0)
        c0, t0 = count[0], time.time()
        samples = [smi() for _ in range(4)]
        dt, n = time.time() - t0, count[0] - c0
        stop.set()
        if th is not None:
            th.join()
        print(json.dumps({"kernel": name, "calls_per_s": round(n / dt, 1) if fn else 0, "us_per_call": round(dt / max(n, 1) * 1e6, 1) if fn else None,
                          "power_w": [s[0] for s in samples], "sclk_mhz": [s[1] for s in samples]}), flush=True)


if __name__ == "__main__":
    main()
