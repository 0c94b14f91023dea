#!/usr/bin/env python3
"""GPU busy time inside the claim-queue phase of a rocprofv3 --kernel-trace results .db of `bench.py` (two lanes: the lanes' kernels
interleave, so proofs cannot be cut apart; the window is what is measured). The window: from the start of the (skip + 1)-th
tracegen::cpu_rows dispatch (one per shard; `skip` = the warm-up and priming shards) to the end of the lanes' last proofs (the second merkle::grind behind the
queue phase's last cpu_rows; the phase ends where the next trace generation is more than 0.3 s away: the verification pause). Prints and writes: shards in the window, ms per shard of the
window, the union of all dispatch intervals (busy), the idle remainder, and per kernel the summed duration per shard (they overlap: the
sum may exceed the window).   rocprof_queue_window.py results.db out.json [skip]"""
import json
import sqlite3
import sys


def main(db_path, out_path, skip):
    db = sqlite3.connect(db_path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    view = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, start, end from {view} order by start").fetchall()
    short = lambda n: n.split("(")[0].replace("void ", "")
    rows = [(short(n), s, e) for n, s, e in rows]
    cpu = [i for i, r in enumerate(rows) if r[0] == "tracegen::cpu_rows"]
    if len(cpu) <= skip + 1:
        raise SystemExit(f"only {len(cpu)} tracegen::cpu_rows dispatches: nothing behind the first {skip}")
    t0 = rows[cpu[skip]][1]
    # the queue phase's trace generations follow each other within tens of milliseconds; the verification pause (hundreds) ends it
    last = skip
    while last + 1 < len(cpu) and rows[cpu[last + 1]][1] - rows[cpu[last]][1] < 300e6:
        last += 1
    cpu = cpu[:last + 1]
    grinds = [r[2] for r in rows if r[0] == "merkle::grind" and r[1] > rows[cpu[-1]][1]]
    # every lane's last proof ends with a grind + queries; the resident leg's grinds come later: take the first `lanes` grinds after the last cpu_rows
    t1 = sorted(grinds)[1] if len(grinds) > 1 else grinds[0]
    inside = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    shards = sum(1 for r in inside if r[0] == "tracegen::cpu_rows")
    busy, cur_s, cur_e = 0, None, None
    for _, s, e in sorted(inside, key=lambda r: r[1]):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    per = {}
    for n, s, e in inside:
        per[n] = per.get(n, 0) + (e - s)
    window = t1 - t0
    out = {"shards_in_window": shards, "window_ms": window / 1e6, "ms_per_shard": window / 1e6 / shards, "busy_ms_per_shard": busy / 1e6 / shards,
           "idle_ms_per_shard": (window - busy) / 1e6 / shards, "busy_fraction": busy / window, "dispatches_per_shard": len(inside) / shards,
           "summed_kernel_ms_per_shard": {k: round(v / 1e6 / shards, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:24]},
           "summed_kernel_ms_per_shard_total": sum(per.values()) / 1e6 / shards}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "summed_kernel_ms_per_shard"}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
