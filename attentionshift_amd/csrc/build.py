"""Build libattnshift_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m attentionshift_amd.csrc.build [--force]

One object per source (so an edit rebuilds one file), then one shared library next to the package:
attentionshift_amd/libattnshift_hip.so.  Objects/so are git-ignored but travel with gpurun snapshots.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(PKG, "libattnshift_hip.so")

SOURCES = ["capi.cpp", "gemm.hip", "gemm_pp.hip", "linear_small.hip", "sdpa.hip", "sdpa_bwd.hip", "attn_bwd.hip", "window_attn.hip", "layernorm.hip", "rollout.hip", "cosine_shift.hip", "shift_final.hip", "ccl.hip", "refine.hip", "chamfer.hip", "small_attn.hip", "roi_align.hip", "mt19937.hip", "gemm_tn.hip"]
# files whose floating-point steps must round exactly like ATen's (no implicit fma contraction)
NO_CONTRACT = {"ccl.hip", "refine.hip"}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".inc")))
    headers.append(os.path.join(os.path.dirname(PKG), "include", "attnshift.h"))
    objs, rebuilt = [], False
    for src in SOURCES:
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            continue
        flags = list(COMMON) + (["-ffp-contract=off"] if src in NO_CONTRACT else [])
        obj = os.path.join(OBJ, src + ".o")
        stamp = obj + ".sha"
        dig = _digest([path] + headers, " ".join(flags))
        if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != dig:
            cmd = [hipcc] + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(dig)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
