"""GPU idle-gap analysis of one bench step from a rocprofv3 kernel-trace DB: span, busy time, the largest gaps and which
kernels bracket them.  python tools/gap_analysis.py <results.db>"""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select start,end,name from kernels order by start").fetchall()
    names = [r[2] for r in rows]
    marks = [i for i, n in enumerate(names) if "attn_mean_rows_kernel" in n]
    if len(marks) < 4:
        print("not enough steps in trace")
        return
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 3
    i0, i1 = marks[k], marks[k + 1]        # one step: roll-out top to the next roll-out top
    seg = rows[i0:i1]
    span = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    print(f"step span {span:.2f} ms, GPU busy {busy:.2f} ms, {len(seg)} kernels")
    # split at the first patch-embed GEMM after the RoI kernels = backbone start
    gaps = [((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1)]
    tot = sum(g[0] for g in gaps if g[0] > 0) / 1e3
    big = [g for g in gaps if g[0] > 15]
    print(f"total gap {tot:.2f} ms; gaps > 15 us: {len(big)} totalling {sum(g[0] for g in big) / 1e3:.2f} ms")
    for g in sorted(gaps, key=lambda g: -g[0])[:25]:
        print(f"{g[0]:8.1f} us  after {g[1]:60s} before {g[2]}")


if __name__ == "__main__":
    main(sys.argv[1])
