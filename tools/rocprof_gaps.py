#!/usr/bin/env python3
"""Where the GPU idles inside a step: from a rocprofv3 --kernel-trace results .db, the gaps between consecutive kernel dispatches (end of one to
start of the next), summed by the kernel that precedes the gap. Usage: rocprof_gaps.py results.db out.json [n_steps]"""
import json
import sqlite3
import sys


def main(db_path, out_path, steps):
    db = sqlite3.connect(db_path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {view} order by start").fetchall()
    gaps, total_gap, busy = {}, 0, 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        busy += e0 - s0
        g = s1 - e0
        if g <= 0:
            continue
        short = n0.split("(")[0].replace("void ", "")
        rec = gaps.setdefault(short, {"count": 0, "total_us": 0.0, "max_us": 0.0})
        rec["count"] += 1
        rec["total_us"] += g / 1e3
        rec["max_us"] = max(rec["max_us"], g / 1e3)
        total_gap += g
    out = {"dispatches": len(rows), "busy_ms": round(busy / 1e6, 3), "idle_ms": round(total_gap / 1e6, 3), "steps": steps,
           "idle_ms_per_step": round(total_gap / 1e6 / steps, 3),
           "after": {k: {"count": v["count"], "total_us": round(v["total_us"], 1), "mean_us": round(v["total_us"] / v["count"], 2), "max_us": round(v["max_us"], 1)}
                     for k, v in sorted(gaps.items(), key=lambda kv: -kv[1]["total_us"])}}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("dispatches", "busy_ms", "idle_ms", "idle_ms_per_step")}))
    for k, v in list(out["after"].items())[:14]:
        print(k, v)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
