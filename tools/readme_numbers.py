"""Regenerate README.md's "Measured" table from the round's committed artifacts, so that no figure in the README can
disagree with profiles/.    python tools/readme_numbers.py r05      (rewrites the block between the measured:begin / end markers)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = os.path.join(ROOT, "profiles")
rec = json.loads(open(os.path.join(P, f"{R}_bench_final.json")).read().strip().splitlines()[-1])
kb = {}
for line in open(os.path.join(P, f"{R}_kernel_bench.md")):
    m = re.match(r"\| `(.*?)` \| ([\d.]+) \| ([\d.]*) \| ([\d.]*) \|", line)
    if m:
        kb[m.group(1)] = (float(m.group(2)), m.group(3), m.group(4))
ab_path = os.path.join(P, f"{R}_ab_vs_r04.txt")                   # (the same-box A/B against an older tree: round 5 only)
ab = [l.split() for l in open(ab_path) if l.startswith(("r04 ", "r05 "))] if os.path.exists(ab_path) else []
old = [float(x[1]) for x in ab if x[0] == "r04"]
new = [float(x[1]) for x in ab if x[0] == "r05"]
ab_text = (f"same box, alternating runs against the round-4 tree: {min(old):.1f}–{max(old):.1f} → **{min(new):.1f}–{max(new):.1f}** "
           f"(`profiles/{R}_ab_vs_r04.txt`); ") if old and new else ""
ro, blk, aff, tr = rec["roofline"], rec.get("roofline_attention_block", {}), rec["roofline_affinity"], rec.get("train", {})
oc = rec.get("other_configs", {})
cpu = rec.get("cpu_baseline", {})
rows = [
    ("hot path, BASELINE config 2 (ViT-B, 1024², 2 images/GPU, 3 objects), `python bench.py`",
     f"**{rec['value']:.1f} images/s** ({rec['ms_per_step']:.3f} ms/step); " + ab_text +
     f"{rec.get('images_per_sec_fp32_parity_path', float('nan'))} with `compute_dtype=float32` (the 1e-3 parity path); "
     f"{rec.get('images_per_sec_reference_rng', float('nan')):.1f} with the reference's literal RNG stream, "
     f"{rec.get('images_per_sec_mil_selector', float('nan')):.1f} with the MIL head choosing the roll-out depth"),
    ("config 4 (ViT-L, 1280², 1 image, 7 objects) / config 5 (Swin-B backbone, 1024², 2 images): legs of the default run",
     f"{oc.get('vitl', {}).get('images_per_sec', float('nan')):.1f} images/s (SDPA at {oc.get('vitl', {}).get('sdpa_fwd', {}).get('frac', float('nan')):.3f} of peak) / "
     f"{oc.get('swinb', {}).get('images_per_sec', float('nan')):.1f} images/s"),
    ("flash attention forward (`as_sdpa_fwd`), HIP events inside the timed region",
     f"{ro['ms_per_launch']:.4f} ms per layer = {ro['achieved']:.0f} TFLOP/s = **{ro['frac']:.3f}** of the dense bf16 peak"),
    ("attention block QKV + SDPA + proj (SURVEY §8d-ii)",
     f"{blk.get('ms_per_layer', {}).get('sum', float('nan')):.4f} ms per layer = **{blk.get('frac', float('nan')):.3f}** of peak "
     f"(by kernel: {blk.get('frac_by_kernel')})"),
    ("mean-shift token affinity (`as_cosine_shift`, both images)",
     f"{aff.get('ms_per_call_alone', float('nan')):.4f} ms alone on the device = {aff.get('frac_alone', float('nan')):.3f} of HBM peak by the algorithmic "
     f"byte count; {aff['ms_per_call']:.4f} ms inside the step, where it shares the device with the mask-point kernels; "
     f"{kb.get('cosine_shift_S5', (float('nan'),))[0]:.4f} ms isolated back-to-back"),
    ("GEMMs at M = 8394 (`tools/kernel_bench.py`, isolated): QKV / proj / fc1 + GELU / fc2; hipBLASLt fc1 / fc2",
     " / ".join(f"{kb.get(k, (float('nan'),))[0] * 1e3:.1f}" for k in ("qkv_gemm_bf16", "proj_gemm_bf16", "fc1_gelu_gemm_bf16", "fc2_gemm_bf16"))
     + " µs; " + " / ".join(f"{kb.get(k, (float('nan'),))[0] * 1e3:.1f}" for k in ("torch_hipblaslt_fc1_bf16", "torch_hipblaslt_fc2_bf16")) + " µs"),
    ("roll-out (7 layers, matched rows) / attention backward (`as_sdpa_bwd`) / attention module backward",
     f"{kb.get('rollout_rows_7layers_bf16_matched3', (float('nan'),))[0]:.3f} ms / {kb.get('sdpa_bwd_bf16(prep+dkv+dq)', (float('nan'),))[0]:.3f} ms / "
     f"{kb.get('attention_bwd_bf16(module)', (float('nan'),))[0]:.3f} ms per layer"),
    ("training step (fwd + attention shift + RoI-head losses + bwd + one-rank RCCL all-reduce + AdamW)",
     f"{tr.get('ms_per_step', float('nan')):.1f} ms ({tr.get('images_per_sec', float('nan')):.1f} images/s)"),
    (f"CPU oracle on {cpu.get('cores')} of {cpu.get('host_cores')} host cores", f"{cpu.get('value')} images/s"),
]
block = [f"## Measured (one MI355X, bf16; every figure from `profiles/{R}_*` by `tools/readme_numbers.py`; boxes differ by up to 8 %)", "",
         "| | |", "|---|---|"] + [f"| {a} | {b} |" for a, b in rows]
text = "\n".join(block) + "\n"
readme = os.path.join(ROOT, "README.md")
s = open(readme).read()
if "<!-- measured:begin -->" in s:
    s = re.sub(r"<!-- measured:begin -->.*?<!-- measured:end -->", "<!-- measured:begin -->\n" + text + "<!-- measured:end -->", s, flags=re.S)
else:
    i0 = s.index("## Measured")
    i1 = s.index("No multi-GPU scaling curve")
    s = s[:i0] + "<!-- measured:begin -->\n" + text + "<!-- measured:end -->\n\n" + s[i1:]
open(readme, "w").write(s)
print(text)
