"""In-kernel timeline of the mean-shift iteration kernels (csrc/cosine_shift.hip built with -DAS_SHIFT_STAMPS).

    python tools/experiments/shift_timeline.py build      (build container or GPU box)
    python tools/experiments/shift_timeline.py run [--md gpurun_out/r05_shift_timeline.md]     (GPU box)

Thread 0 of every workgroup stamps s_memrealtime (100 MHz, one counter for the chip) at the phase boundaries of
shift_sim / shift_assign / shift_aggregate; the script runs as_cosine_shift on the config-2 inputs of tools/kernel_bench.py
(B = 2, 3 objects per image, P = 20, C = 768, 64 x 64 patches, S = 5), reads the stamps of the last of several calls and
prints, per launch: when the first workgroup entered (relative to the call's first stamp), the median / maximum of every
phase over the workgroups that do work, when the last workgroup finished, and the gap to the next launch's first entry.
The instrumented library is ~10 % slower than the shipped one (the drains before the 'landed' stamps); it is an experiment
build under tools/experiments/_build/, never the product library.
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "attentionshift_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
LIB = os.path.join(OUT, "libshift_stamps.so")
NL, NB, NS = 16, 1024, 8
PHASES = {
    "shift_sim": ["entry->box", "box->operands landed", "operands->MFMA chain issued", "MFMA->LDS exchange", "exchange->results stored"],
    "shift_assign": ["entry->box", "box->blind loads landed", "loads->tau/max", "tau/max->Z", "Z->assignment stored"],
    "shift_aggregate": ["entry->box", "box->first loads landed", "loads->accumulation done", "accumulation->stored"],
}
KERNELS = ["shift_sim", "shift_assign", "shift_aggregate"]


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CS, "_obj", f) for f in os.listdir(os.path.join(CS, "_obj"))
            if f.endswith(".o") and not f.startswith("cosine_shift.hip")]
    o = os.path.join(OUT, "cosine_shift_stamps.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DAS_SHIFT_STAMPS", "-c",
                           os.path.join(CS, "cosine_shift.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, o] + objs)
    print("built", LIB)


def run(md):
    import torch
    sys.path.insert(0, ROOT)
    from attentionshift_amd import synthetic
    dev = torch.device("cuda", 0)
    lib = ctypes.CDLL(LIB)
    vp, ci, cd, cz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    lib.as_cosine_shift_workspace_bytes.restype = cz
    lib.as_cosine_shift_workspace_bytes.argtypes = [ci] * 6
    lib.as_cosine_shift.restype = ci
    lib.as_cosine_shift.argtypes = [vp] * 5 + [cd, cd, ci] + [vp] * 4 + [cz] + [ci] * 6 + [vp]
    lib.as_shift_stamps_read.restype = ci
    lib.as_shift_stamps_read.argtypes = [vp, cz, ci]
    B, D, hp, wp, S, P = 2, 768, 64, 64, 5, 20
    feats, boxes, prots, obj = [], [], [], []
    for b in range(B):
        inp = synthetic.shift_inputs(100 + b, hp, wp, D, 3, 1)
        f = inp["vit_feat"].flatten(1).t().contiguous()
        feats.append(f)
        pb = inp["patch_boxes"].int()
        boxes.append(pb)
        for gi in range(3):
            x0, y0, x1, y1 = pb[gi].tolist()
            ys = torch.linspace(y0, y1, 5).long()
            xs = torch.linspace(x0, x1, 4).long()
            prots.append(f[(ys[:, None] * wp + xs[None, :]).flatten()])
            obj.append(b)
    feat = torch.stack(feats).to(dev)
    box_patch = torch.cat(boxes).to(dev)
    prot = torch.stack(prots).to(dev).contiguous()
    obj_img = torch.tensor(obj, dtype=torch.int32, device=dev)
    G = len(obj)
    Np = hp * wp
    wsb = lib.as_cosine_shift_workspace_bytes(B, D, hp, wp, G, P)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    pout = torch.empty_like(prot)
    sim = torch.empty(G, P, Np, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = lib.as_cosine_shift(feat.data_ptr(), box_patch.data_ptr(), obj_img.data_ptr(), prot.data_ptr(), pout.data_ptr(), 0.1, 0.1, S,
                                 sim.data_ptr(), None, None, ws.data_ptr(), wsb, B, D, hp, wp, G, P, st)
        assert rc == 0, rc

    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    per_call = e0.elapsed_time(e1) / 20 * 1e3
    buf = np.zeros((NL, NB, NS), dtype=np.uint64)
    assert lib.as_shift_stamps_read(buf.ctypes.data, buf.nbytes, 1) == 0
    call()                                              # ONE call after clearing: its stamps only
    torch.cuda.synchronize()
    assert lib.as_shift_stamps_read(buf.ctypes.data, buf.nbytes, 0) == 0
    t = buf.astype(np.float64) * 0.01                   # 100 MHz ticks -> us
    t[buf == 0] = np.nan
    origin = np.nanmin(t)
    t -= origin
    lines = ["# r05 in-kernel timeline of the mean-shift iteration (s_memrealtime stamps, csrc/cosine_shift.hip -DAS_SHIFT_STAMPS)", "",
             f"config 2 inputs of tools/kernel_bench.py: B = {B}, {G} objects, P = {P}, C = {D}, {hp} x {wp} patches, S = {S}; instrumented "
             f"library {per_call:.1f} us per call (back-to-back, HIP events).  Times in us relative to the first stamp of the call; "
             "phases are per workgroup (thread 0), `med / max` over the workgroups that do work (in-box tiles / channel blocks).", "",
             "| launch | kernel | workgroups with work | first entry | last entry | " + "phase durations, med / max | last exit | gap to next first entry |",
             "|---|---|---|---|---|---|---|---|"]
    ends = []
    rows = []
    for L in range(3 * S):
        kern = KERNELS[L % 3]
        nph = len(PHASES[kern])
        x = t[L]
        last_slot = nph                                 # slots 0..nph
        work = ~np.isnan(x[:, last_slot])
        entered = ~np.isnan(x[:, 0])
        if not work.any():
            continue
        xe = x[work]
        ph = []
        for k in range(nph):
            d = xe[:, k + 1] - xe[:, k]
            ph.append(f"{PHASES[kern][k]} {np.nanmedian(d):.2f} / {np.nanmax(d):.2f}")
        rows.append((L, kern, int(work.sum()), float(np.nanmin(x[entered, 0])), float(np.nanmax(x[entered, 0])), ph,
                     float(np.nanmax(xe[:, last_slot]))))
    for i, (L, kern, nw, fe, le, ph, lx) in enumerate(rows):
        gap = rows[i + 1][3] - lx if i + 1 < len(rows) else float("nan")
        lines.append(f"| {L} | `{kern}` it {L // 3} | {nw} | {fe:.2f} | {le:.2f} | " + "; ".join(ph) + f" | {lx:.2f} | {gap:.2f} |")
    # per-kernel averages over iterations 1..S-1 (iteration 0 has no previous assignment / norm partials)
    lines += ["", "Averages over iterations 1.." + str(S - 1) + " (us): span = last exit - first entry of the launch; critical = median "
              "workgroup's entry -> exit; boundary = next launch's first entry - this launch's last exit.", "",
              "| kernel | span | median workgroup entry->exit | boundary after it |", "|---|---|---|---|"]
    for kern in KERNELS:
        sp, cr, bd = [], [], []
        for i, (L, k2, nw, fe, le, ph, lx) in enumerate(rows):
            if k2 != kern or L < 3:
                continue
            sp.append(lx - fe)
            x = t[L]
            work = ~np.isnan(x[:, len(PHASES[kern])])
            cr.append(float(np.nanmedian(x[work, len(PHASES[kern])] - x[work, 0])))
            if i + 1 < len(rows):
                bd.append(rows[i + 1][3] - lx)
        lines.append(f"| `{kern}` | {np.mean(sp):.2f} | {np.mean(cr):.2f} | {np.mean(bd) if bd else float('nan'):.2f} |")
    total = rows[-1][6] - rows[0][3]
    lines += ["", f"First entry of iteration 0 -> last exit of iteration {S - 1}'s aggregation: {total:.1f} us (the final similarity "
              "launch follows)."]
    text = "\n".join(lines) + "\n"
    print(text)
    if md:
        open(md, "w").write(text)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["build", "run"])
    ap.add_argument("--md", default="")
    a = ap.parse_args()
    build() if a.cmd == "build" else run(a.md)
