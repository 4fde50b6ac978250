"""Summarise a rocprofv3 results .db (kernel trace) into a markdown table under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db profiles/r01_bench_kernel_stats.md "title"
"""
import sqlite3
import sys


def main(db, out, title):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"# {title}", "", f"source: `{db}` (rocprofv3 --kernel-trace --stats); durations in microseconds", "",
             "| kernel | calls | total us | avg us | min us | max us | % | VGPR | LDS B |", "|---|---|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx, vg, lds in rows[:60]:
        width = int(__import__("os").environ.get("PROF_NAME_WIDTH", "110"))
        short = name if len(name) < width else name[:width - 3] + "..."
        lines.append(f"| `{short}` | {n} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | "
                     f"{100.0 * tot / total:.1f} | {vg} | {lds} |")
    lines.append("")
    lines.append(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    if __import__("os").environ.get("PROF_TIMELINE"):        # kernel sequence of the last N ms of the trace -> <out>.timeline.tsv
        cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
        sc, ec = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
        ev = c.execute(f"select {sc}, {ec}, name, {qcol} from kernels order by {sc}").fetchall()
        t_end = max(e[1] for e in ev)
        win = float(__import__("os").environ["PROF_TIMELINE"]) * 1e6
        ev = [e for e in ev if e[0] >= t_end - win]
        t0 = ev[0][0]
        with open(out.replace(".md", "") + ".timeline.tsv", "w") as f:
            f.write("start_us\tdur_us\tgap_us\tqueue\tkernel\n")
            prev_end = t0
            for st, en, name, q in ev:
                f.write(f"{(st - t0) / 1e3:.1f}\t{(en - st) / 1e3:.1f}\t{(st - prev_end) / 1e3:.1f}\t{q}\t{name[:90]}\n")
                prev_end = max(prev_end, en)
    if __import__("os").environ.get("PROF_GAPS"):            # device idle analysis: union of busy intervals, largest gaps
        cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
        sc, ec = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        ev = c.execute(f"select {sc}, {ec}, name from kernels order by {sc}").fetchall()
        n = len(ev)
        ev = ev[n // 3:]                                   # skip warm-up / model build
        busy, cur_end, gaps = 0, ev[0][0], []
        for st, en, name in ev:
            if st > cur_end:
                gaps.append((st - cur_end, name))
                busy += en - st
                cur_end = en
            elif en > cur_end:
                busy += en - cur_end
                cur_end = en
        span = cur_end - ev[0][0]
        lines += ["", f"## device idle (last two thirds of the trace): span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms "
                      f"({100.0 * busy / span:.1f} %), {len(gaps)} gaps"]
        big = sorted(gaps, reverse=True)[:int(__import__("os").environ.get("PROF_GAPS"))]
        agg = {}
        for g, name in gaps:
            if g > 3000:
                a = agg.setdefault(name[:70], [0, 0])
                a[0] += 1
                a[1] += g
        lines += ["", "| gap before kernel (gaps > 3 us) | count | total us |", "|---|---|---|"]
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            lines.append(f"| `{k}` | {v[0]} | {v[1] / 1e3:.1f} |")
        open(out, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines[-32:]))
        return
    pat = __import__("os").environ.get("PROF_BY_GRID")        # e.g. PROF_BY_GRID=gemm: split matching kernels by grid size
    if pat:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
        gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
        if gcol:
            lines += ["", f"## kernels matching `{pat}` by grid size", "", "| kernel | grid x | calls | avg us | min us | max us |",
                      "|---|---|---|---|---|---|"]
            q = (f"select name, {gcol}, count(*), avg(duration), min(duration), max(duration) from kernels where name like ? "
                 f"group by name, {gcol} order by sum(duration) desc")
            for name, gx, n, avg, mn, mx in c.execute(q, (f"%{pat}%",)).fetchall()[:40]:
                lines.append(f"| `{name[:60]}` | {gx} | {n} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} |")
        else:
            lines += ["", f"(no grid column among {cols})"]
    open(out, "w").write("\n".join(lines) + "\n")
    if pat:
        print("\n".join(lines[-44:]))
        return
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")
