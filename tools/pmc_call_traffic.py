"""Per-CALL HBM traffic of a multi-launch C entry point from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE result DBs.

    python tools/pmc_call_traffic.py <dir with *_results.db> <marker kernel: last launch of a call> <pattern,pattern> <out prefix>
Dispatches are cut into calls at every dispatch of the marker kernel; the counters of all dispatches whose kernel name
contains one of the patterns are summed per call and averaged over the second half of the calls (warm).  FETCH_SIZE is
doubled (gfx950 correction, MI355X_MICROARCH.md "HBM"); WRITE_SIZE is reported uncalibrated."""
import glob
import json
import sqlite3
import sys


def main(d, marker, patterns, out):
    pats = patterns.split(",")
    res, detail = {}, {}
    for db in sorted(glob.glob(f"{d}/*_results.db")):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
        cname = "counter_name" if "counter_name" in cols else "pmc_name" if "pmc_name" in cols else \
            [x for x in cols if "name" in x and x != "name"][0]
        vcol = "counter_value" if "counter_value" in cols else "value"
        rows = list(c.execute(f"select {cname}, dispatch_id, name, sum({vcol}) from pmc_events group by {cname}, dispatch_id "
                              f"order by dispatch_id"))
        for counter in sorted({r[0] for r in rows}):
            calls, cur, per_kernel = [], 0.0, {}
            seq = [(disp, name, v) for n, disp, name, v in rows if n == counter and any(p in name for p in pats)]
            n_calls = sum(1 for _, name, _ in seq if marker in name)
            seen = 0
            for disp, name, v in seq:
                cur += v
                if seen >= n_calls // 2:
                    key = next(p for p in ("shift_final_sim", "shift_sim_full", "shift_sim", "shift_assign", "shift_aggregate", "prot_norm2", name)
                               if p in name)
                    per_kernel[key] = per_kernel.get(key, 0.0) + v
                if marker in name:
                    calls.append(cur)
                    cur = 0.0
                    seen += 1
            warm = calls[len(calls) // 2:]
            res[counter] = sum(warm) / max(len(warm), 1)
            detail[counter] = {k: v / max(len(warm), 1) for k, v in per_kernel.items()}
            detail[counter]["calls_averaged"] = len(warm)
    rd = res.get("FETCH_SIZE", float("nan")) * 1024 * 2
    wr = res.get("WRITE_SIZE", float("nan")) * 1024
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) on tools/kernel_bench.py "
                     "--only shift; FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE uncalibrated",
           "shape": "2 images x 3 objects, 64x64 patches, C=768 fp32, 20 prototypes, S=5",
           "hbm_read_bytes_per_call": rd, "hbm_write_bytes_per_call": wr, "per_call_bytes": rd + wr,
           "raw_kb_per_call": res, "raw_kb_per_call_by_kernel": detail}
    json.dump(rec, open(out + ".json", "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
