#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of one shard proof, from a rocprofv3 --kernel-trace results .db: where the stream waits for the host.
Usage: gap_analysis.py results.db [min_gap_us]"""
import sqlite3
import sys


def main(path, min_gap=8.0):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
    t = [x for x in tables if x.startswith("kernels")] or [x for x in tables if "kernel_dispatch" in x]
    cols = [r[1] for r in db.execute(f"pragma table_info({t[0]})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {t[0]} order by start").fetchall()
    # the last proof: from the last hash_leaves-after-long-idle ... simply take the final third of the dispatches
    rows = rows[len(rows) * 2 // 3:]
    busy = sum(e - s for _, s, e in rows) / 1e3
    span = (rows[-1][2] - rows[0][1]) / 1e3
    gaps = []
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1e3
        if g >= min_gap:
            gaps.append((g, n0.split("(")[0][-40:], n1.split("(")[0][-40:]))
    print(f"{len(rows)} kernels, span {span:.0f} us, busy {busy:.0f} us, idle {span - busy:.0f} us; gaps >= {min_gap} us: {len(gaps)} totalling {sum(g for g, _, _ in gaps):.0f} us")
    agg = {}
    for g, a, b in gaps:
        k = (a, b)
        agg[k] = (agg.get(k, (0, 0))[0] + g, agg.get(k, (0, 0))[1] + 1)
    for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"{g:8.0f} us  x{c:3d}   {a}  ->  {b}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0)
