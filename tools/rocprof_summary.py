#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite, `--kernel-trace --stats`) into the per-kernel CSV
summary committed under profiles/. Usage: rocprof_summary.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            w.writerow([short, calls, round(total, 3), round(avg, 3), round(pct, 3)])
    print(f"{len(rows)} kernels -> {out_path}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
