"""Copy a round's measurement summaries from gpurun_out/ (scratch, merged back by gpurun) into profiles/ (tracked), and
render the kernel_bench JSON lines as a markdown table.    python tools/collect_profiles.py r05

Nothing is computed here: every number is what tools/round_artifacts.sh measured on the GPU box."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
copied = []
for name in sorted(os.listdir(src)):
    p = os.path.join(src, name)
    if not os.path.isfile(p):
        continue
    if name.startswith(R + "_") and name.endswith((".md", ".json", ".txt", ".jsonl")) and "kernel_bench.jsonl" not in name \
            and not name.startswith(R + "_bench") or name in (f"{R}_bench_final.json", f"{R}_bench_kernel_stats_final.md", f"{R}_bench_wall.txt"):
        shutil.copy(p, os.path.join(dst, name))
        copied.append(name)
    elif name.startswith(f"pmc_sdpa_{R}_") and name.endswith(".md"):          # PMC summaries: pmc_sdpa_r05_<kernel>.md -> r05_<kernel>_pmc.md
        out = f"{R}_{name[len('pmc_sdpa_' + R + '_'):-3]}_pmc.md"
        shutil.copy(p, os.path.join(dst, out))
        copied.append(out)
    elif name == f"{R}_ab_vs_r04.log" or name == f"{R}_glue_sites.log":
        shutil.copy(p, os.path.join(dst, name.replace(".log", ".txt")))
        copied.append(name)
kb = os.path.join(src, f"{R}_kernel_bench.jsonl")
if os.path.exists(kb):
    rows = []
    for line in open(kb):
        line = line.strip()
        if line.startswith("{"):
            rows.append(json.loads(line))
    with open(os.path.join(dst, f"{R}_kernel_bench.md"), "w") as f:
        f.write(f"# {R} kernel_bench (tools/kernel_bench.py --reps 20, one MI355X box; HIP events, back-to-back calls, fastest of three batches)\n\n")
        f.write("| kernel | ms | TFLOP/s or GB/s | fraction of peak | note |\n|---|---|---|---|---|\n")
        for r in rows:
            rate = r.get("tflops", r.get("gbps", ""))
            f.write(f"| `{r['kernel']}` | {r['ms']} | {rate} | {r.get('frac', '')} | {r.get('note', '')} |\n")
    copied.append(f"{R}_kernel_bench.md")
print("\n".join(copied))
