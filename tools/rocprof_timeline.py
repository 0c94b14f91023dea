#!/usr/bin/env python3
"""The dispatch timeline of the LAST proof in a rocprofv3 --kernel-trace results .db: per kernel name the launches, the time they run and
the idle time in front of them (end of the previous dispatch to the start of this one), and — with a third argument — every dispatch in
order (start offset, duration, gap before, name). Proofs are cut at `merkle::grind` as in tools/rocprof_gaps.py.
Usage: rocprof_timeline.py results.db out.json [dispatches.csv]"""
import json
import sqlite3
import sys


def main(db_path, out_path, csv_path=None):
    db = sqlite3.connect(db_path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {view} order by start").fetchall()
    short = lambda n: n.split("(")[0].replace("void ", "")
    cuts = [i for i, r in enumerate(rows) if short(r[0]) == "merkle::grind"]
    if len(cuts) < 2:
        raise SystemExit("fewer than two proofs in the trace")
    lo = cuts[-2] + 1
    while lo < len(rows) and short(rows[lo][0]).startswith("open::gather"):
        lo += 1
    hi = cuts[-1] + 1
    while hi < len(rows) and short(rows[hi][0]).startswith("open::gather"):
        hi += 1
    pr = rows[lo:hi]
    t0 = pr[0][1]
    per, order, cur_end = {}, [], pr[0][1]
    for n, s, e in pr:
        gap = max(0, s - cur_end)
        rec = per.setdefault(short(n), {"launches": 0, "run_us": 0.0, "idle_before_us": 0.0})
        rec["launches"] += 1
        rec["run_us"] += (e - s) / 1e3
        rec["idle_before_us"] += gap / 1e3
        order.append(((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short(n)))
        cur_end = max(cur_end, e)
    span = (cur_end - t0) / 1e3
    out = {"dispatches": len(pr), "span_us": round(span, 1), "run_us": round(sum(r["run_us"] for r in per.values()), 1),
           "idle_us": round(sum(r["idle_before_us"] for r in per.values()), 1),
           "by_kernel": {k: {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()}
                         for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["run_us"] + kv[1]["idle_before_us"]))}}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    if csv_path:
        with open(csv_path, "w") as f:
            f.write("start_us,duration_us,idle_before_us,kernel\n")
            for s, d, g, n in order:
                f.write(f"{s:.1f},{d:.1f},{g:.1f},{n}\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
