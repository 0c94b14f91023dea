#!/usr/bin/env python3
"""Per-DISPATCH counter table of one kernel from a rocprofv3 --pmc results database (the per-kernel sums of pmc_sq_summary.py hide which
launches of a kernel that is launched 58 times per proof are the inefficient ones).
Usage: pmc_per_dispatch.py results.db kernel_substring out.csv
One row per dispatch in launch order: grid size (work-items), duration, waves, VALU instructions per wave, VALU pipe busy
(100 * SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024): both from the same pass), mean resident waves per SIMD."""
import collections
import csv
import sqlite3
import sys

N_SIMD = 1024


def main(db_path, pattern, out_path):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ident = "dispatch_id" if "dispatch_id" in cols else "id"
    grid = "grid_size" if "grid_size" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    start = "start" if "start" in cols else ident
    rows = db.execute(f"select {ident}, kernel_name, counter_name, sum(value), max(duration), max({grid}), min({start}) from counters_collection "
                      f"where kernel_name like ? group by {ident}, counter_name", (f"%{pattern}%",)).fetchall()
    per = collections.OrderedDict()
    for did, k, c, v, dur, g, st in sorted(rows, key=lambda r: (r[6], r[0])):
        d = per.setdefault(did, {"dur": dur, "grid": g, "name": k.split("(")[0].replace("void ", "")})
        d[c] = v
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["dispatch", "kernel", "grid_work_items", "duration_us", "waves", "valu_insts_per_wave", "valu_pipe_busy_pct", "mean_waves_per_simd"])
        for did, d in per.items():
            gui = d.get("GRBM_GUI_ACTIVE", 0)
            waves = max(d.get("SQ_WAVES", 0), 1)
            busy = round(100 * d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (gui / 8 * N_SIMD), 1) if gui else ""
            res = round(d.get("SQ_WAVE_CYCLES", 0) * 4 / (gui / 8 * N_SIMD), 2) if gui else ""
            w.writerow([did, d["name"], d["grid"], round(d["dur"] / 1e3, 1), int(d.get("SQ_WAVES", 0)), round(d.get("SQ_INSTS_VALU", 0) / waves), busy, res])
    print("wrote", out_path, len(per), "dispatches; columns seen:", cols)


if __name__ == "__main__":
    main(*sys.argv[1:4])
