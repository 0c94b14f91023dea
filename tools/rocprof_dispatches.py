#!/usr/bin/env python3
"""Per-dispatch durations of the named kernels inside the LAST proof of a rocprofv3 --kernel-trace results .db (proofs cut at merkle::grind).
Usage: rocprof_dispatches.py results.db kernel_substring [kernel_substring ...]"""
import sqlite3
import sys


def main(db_path, *subs):
    db = sqlite3.connect(db_path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else "kernel_name"
    extra = [c for c in ("grid_size", "grid_size_x", "workgroup_size", "workgroup_size_x", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_block_size", "scratch_size") if c in cols]
    rows = db.execute(f"select {name_col}, start, end{''.join(', ' + c for c in extra)} from {view} order by start").fetchall()
    short = lambda n: n.split("(")[0].replace("void ", "")
    cuts = [i for i, r in enumerate(rows) if short(r[0]) == "merkle::grind"]
    lo = cuts[-2] + 1 if len(cuts) >= 2 else 0
    last = rows[lo:cuts[-1] + 1] if cuts else rows
    t0 = last[0][1]
    for r in last:
        n = short(r[0])
        if any(s in n for s in subs):
            print(f"{(r[1] - t0) / 1e6:9.3f} ms  {n:34s} {(r[2] - r[1]) / 1e3:9.1f} us  " + " ".join(f"{c}={v}" for c, v in zip(extra, r[3:])))


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
