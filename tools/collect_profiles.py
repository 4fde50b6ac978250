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

# the SDPA launch's HBM bytes as bench.py reads them (roofline.traffic): from the round's PMC summary of the shipped kernel
pmc = os.path.join(dst, f"{R}_sdpa_pmc.md")
if os.path.exists(pmc):
    import re
    txt = open(pmc).read()
    rd = re.search(r"hbm_read_bytes_corrected = .*? = ([\d.e+]+) B", txt)
    fs = re.search(r"\| FETCH_SIZE \| ([\d.e+]+) \|", txt)
    ws = re.search(r"\| WRITE_SIZE \| ([\d.e+]+) \|", txt)
    if fs and ws:
        read_b, write_b = float(fs.group(1)) * 1024 * 2, float(ws.group(1)) * 1024
        with open(os.path.join(dst, f"{R}_sdpa_traffic.json"), "w") as f:
            json.dump({"source": f"profiles/{R}_sdpa_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only; "
                                 "FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE uncalibrated)",
                       "note": f"round {R[1:].lstrip('0')}, shipped kernel sdpa_fwd_pipe_kernel<2,0,1,8> (512-row workgroups of eight waves), unchanged since round 4",
                       "shape": "B=2 h=12 N=4197 bf16, one launch per as_sdpa_fwd call",
                       "hbm_read_bytes": read_b, "hbm_write_bytes": write_b, "per_as_sdpa_fwd_call_bytes": read_b + write_b,
                       "algorithmic_bytes": 53700000.0}, f, indent=1)
        print(f"{R}_sdpa_traffic.json")
