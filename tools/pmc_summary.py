"""Summarise rocprofv3 --pmc result DBs of one kernel into a markdown table.

    python tools/pmc_summary.py gpurun_out/pmc_sdpa sdpa_fwd_glds profiles/r01_sdpa_pmc.md "title"
Per counter: mean over dispatches of the SUM over all instances (SE / XCD dimensions) of the matching kernel.
FETCH_SIZE / WRITE_SIZE are reported raw (KB) and as bytes; the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md ("HBM") is applied in the `hbm_read_bytes_corrected` line.
"""
import glob
import sqlite3
import sys


def main(d, pat, out, title):
    lines = [f"# {title}", "", f"source: `{d}/p*_results.db` (rocprofv3 --pmc ... --kernel-trace, one pass per counter group); "
             f"kernel filter `{pat}`", "", "| counter | mean per dispatch (sum over instances) | dispatches | avg kernel us |", "|---|---|---|---|"]
    vals = {}
    for db in sorted(glob.glob(f"{d}/*_results.db")):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
        cname = "counter_name" if "counter_name" in cols else "pmc_name" if "pmc_name" in cols else None
        vcol = "counter_value" if "counter_value" in cols else "value"
        if cname is None:
            cname = [x for x in cols if "name" in x and x != "name"][0]
        q = (f"select {cname}, dispatch_id, sum({vcol}), max(duration) from pmc_events where name like ? "
             f"group by {cname}, dispatch_id")
        per = {}
        for n, disp, v, dur in c.execute(q, (f"%{pat}%",)):
            per.setdefault(n, []).append((v, dur))
        for n, lst in per.items():
            lst = lst[len(lst) // 2:]                # drop warm-up dispatches
            m = sum(v for v, _ in lst) / len(lst)
            du = sum(u for _, u in lst) / len(lst) / 1e3
            vals[n] = m
            lines.append(f"| {n} | {m:.6g} | {len(lst)} | {du:.1f} |")
    lines.append("")
    if "FETCH_SIZE" in vals:
        lines.append(f"hbm_read_bytes_corrected = FETCH_SIZE[KB] x 1024 x 2 = {vals['FETCH_SIZE'] * 2048:.4g} B per launch")
    if "WRITE_SIZE" in vals:
        lines.append(f"hbm_write_bytes (uncalibrated) = WRITE_SIZE[KB] x 1024 = {vals['WRITE_SIZE'] * 1024:.4g} B per launch")
    if "SQ_WAVE_CYCLES" in vals:
        w = vals["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in vals:
                lines.append(f"{k} / SQ_WAVE_CYCLES = {vals[k] / w:.3f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "PMC summary")
