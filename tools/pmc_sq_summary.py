#!/usr/bin/env python3
"""Per-kernel SQ counter summary from a rocprofv3 --pmc results .db: where wave cycles go.
Usage: pmc_sq_summary.py results.db out.csv"""
import collections
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration), max(vgpr_count), "
                      "max(lds_block_size) from counters_collection group by kernel_name, counter_name").fetchall()
    agg, meta = collections.defaultdict(dict), {}
    for k, c, v, n, dur, vg, lds in rows:
        short = k.split("(")[0].replace("void ", "")
        agg[short][c] = v
        meta[short] = (n, dur, vg, lds)
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Dispatches", "TotalDurationUs", "VGPRs", "LDSBytes", "WaitAnyPct(memory/barrier)",
                    "WaitInstAnyPct(issue stall)", "ValuActivePct", "LdsActivePct", "ValuInstsPerWave"])
        for k, d in sorted(agg.items(), key=lambda kv: -meta[kv[0]][1]):
            n, dur, vg, lds = meta[k]
            wc = d.get("SQ_WAVE_CYCLES", 0) or 1
            pct = lambda c: round(100 * d.get(c, 0) / wc, 1)
            w.writerow([k, n, round(dur / 1e3, 1), vg, lds, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"),
                        pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"),
                        round(d.get("SQ_INSTS_VALU", 0) / max(d.get("SQ_WAVES", 1), 1))])
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
