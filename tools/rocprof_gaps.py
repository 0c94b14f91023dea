#!/usr/bin/env python3
"""Where the GPU idles inside a proof: from a rocprofv3 --kernel-trace results .db, the gaps between consecutive kernel dispatches (end of one to
start of the next), summed by the kernel that precedes the gap.

A proof ends with its queries right after the one `merkle::grind` dispatch, so the trace is cut at the grind kernels: proof i = the dispatches
after grind i-1's trailing gather up to and including grind i. Set-up work (first-use tables, uploads of the benchmark's traces, warm-up
compilation) lies before the first cut or inside the first proofs; the figures reported are over the LAST `n_last` proofs only (default: all
but the first two), per proof: dispatches, busy ms, idle ms (gaps between dispatches of the same proof).
Usage: rocprof_gaps.py results.db out.json [n_last]"""
import json
import sqlite3
import sys


def main(db_path, out_path, n_last=None):
    db = sqlite3.connect(db_path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {view} order by start").fetchall()
    short = lambda n: n.split("(")[0].replace("void ", "")
    # cut after each grind (+ the gather kernels that follow it inside the same proof)
    cuts = [i for i, r in enumerate(rows) if short(r[0]) == "merkle::grind"]
    proofs, lo = [], 0
    for c in cuts:
        hi = c + 1
        while hi < len(rows) and short(rows[hi][0]).startswith("open::gather"):
            hi += 1
        proofs.append(rows[lo:hi])
        lo = hi
    if not proofs:
        proofs = [rows]
    n_last = n_last or max(1, len(proofs) - 2)
    sel = proofs[-n_last:]
    gaps, total_gap, busy, ndisp = {}, 0, 0, 0
    for pr in sel:
        ndisp += len(pr)
        # kernels of the main and the side stream overlap (round 4): busy = the union of the dispatch intervals, a gap = time no kernel runs,
        # charged to the kernel that ended last before it
        cur_end, last = pr[0][1], pr[0][0]
        for n1, s1, e1 in pr:
            if s1 > cur_end:
                g = s1 - cur_end
                rec = gaps.setdefault(short(last), {"count": 0, "total_us": 0.0, "max_us": 0.0})
                rec["count"] += 1
                rec["total_us"] += g / 1e3
                rec["max_us"] = max(rec["max_us"], g / 1e3)
                total_gap += g
                busy += 0
            if e1 > cur_end:
                busy += e1 - max(cur_end, s1)
                cur_end, last = e1, n1
    k = len(sel)
    counts = {}
    for pr in sel:
        for n, _, _ in pr:
            counts[short(n)] = counts.get(short(n), 0) + 1
    out = {"proofs_in_trace": len(proofs), "proofs_measured": k, "dispatches_per_proof": round(ndisp / k, 1),
           "busy_ms_per_proof": round(busy / 1e6 / k, 3), "idle_ms_per_proof": round(total_gap / 1e6 / k, 3),
           "span_ms_per_proof": round(sum(pr[-1][2] - pr[0][1] for pr in sel) / 1e6 / k, 3),
           "dispatches_per_proof_by_kernel": {n: round(c / k, 1) for n, c in sorted(counts.items(), key=lambda kv: -kv[1])},
           "gap_after": {n: {"per_proof": round(v["count"] / k, 1), "total_us_per_proof": round(v["total_us"] / k, 1),
                             "mean_us": round(v["total_us"] / v["count"], 2), "max_us": round(v["max_us"], 1)}
                         for n, v in sorted(gaps.items(), key=lambda kv: -kv[1]["total_us"])}}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({x: out[x] for x in ("proofs_in_trace", "proofs_measured", "dispatches_per_proof", "busy_ms_per_proof", "idle_ms_per_proof", "span_ms_per_proof")}))
    for n, v in list(out["gap_after"].items())[:16]:
        print(n, v)
    print(out["dispatches_per_proof_by_kernel"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
