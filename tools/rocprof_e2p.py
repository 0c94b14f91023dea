#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace --memory-copy-trace results .db of tools/bench_e2p.py: per shard of the pipelined loop (cut at the grind
kernels) the span, the kernels' busy time, the idle gaps, and the host-to-device copies — how many bytes, how long the copy engine was
busy, how much of that lay under kernels of the running proof, and how long the proof's kernels took while a copy was in flight versus
the same kernels without one.   Usage: rocprof_e2p.py results.db out.json [n_last]"""
import json
import sqlite3
import sys


def intervals_overlap(a, b):
    """total length of the intersection of two sorted interval lists"""
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def merge(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def main(db_path, out_path, n_last=None):
    db = sqlite3.connect(db_path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    kview = "kernels" if "kernels" in tables else next(t for t in tables if "kernel_dispatch" in t)
    kcols = [r[1] for r in db.execute(f"pragma table_info({kview})")]
    kname = "name" if "name" in kcols else "kernel_name"
    kern = db.execute(f"select {kname}, start, end from {kview} order by start").fetchall()
    mview = "memory_copies" if "memory_copies" in tables else next((t for t in tables if "memory_copy" in t), None)
    copies = []
    if mview:
        mcols = [r[1] for r in db.execute(f"pragma table_info({mview})")]
        size_col = "size" if "size" in mcols else next((c for c in mcols if "size" in c or "bytes" in c), None)
        name_col = "name" if "name" in mcols else next((c for c in mcols if "kind" in c or "direction" in c or "name" in c), None)
        copies = db.execute(f"select {name_col}, start, end, {size_col} from {mview} order by start").fetchall()
    short = lambda s: s.split("(")[0].replace("void ", "")
    cuts = [i for i, r in enumerate(kern) if short(r[0]) == "merkle::grind"]
    proofs, lo = [], 0
    for c in cuts:
        hi = c + 1
        while hi < len(kern) and short(kern[hi][0]).startswith("open::gather"):
            hi += 1
        proofs.append(kern[lo:hi])
        lo = hi
    n_last = n_last or max(1, len(proofs) - 2)
    sel = proofs[-n_last:]
    big = [(s, e, sz, nm) for nm, s, e, sz in copies if sz and sz >= (1 << 20) and "HOST_TO_DEVICE" in str(nm).upper().replace(" ", "_")]
    res = []
    for pr in sel:
        t0, t1 = pr[0][1], pr[-1][2]
        kiv = merge([(s, e) for _, s, e in pr])
        busy = sum(e - s for s, e in kiv)
        civ = merge([(max(s, t0), min(e, t1)) for s, e, _, _ in big if e > t0 and s < t1])
        cbytes = sum(sz for s, e, sz, _ in big if e > t0 and s < t1)
        cbusy = sum(e - s for s, e in civ)
        under = intervals_overlap(kiv, civ)
        res.append({"span_ms": (t1 - t0) / 1e6, "kernel_busy_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6, "h2d_bytes": cbytes, "h2d_busy_ms": cbusy / 1e6,
                    "h2d_under_kernels_ms": under / 1e6, "h2d_GBps": cbytes / max(cbusy, 1) if cbusy else None, "dispatches": len(pr)})
    k = len(res)
    avg = {key: round(sum(r[key] or 0 for r in res) / k, 3) for key in res[0]}
    # the proofs' own kernels (hashing, LDE...): mean duration by name, for comparison with a resident-trace run of the same build
    by = {}
    for pr in sel:
        for nm, s, e in pr:
            a = by.setdefault(short(nm), [0, 0])
            a[0] += e - s
            a[1] += 1
    out = {"shards_in_trace": len(proofs), "shards_measured": k, "per_shard": avg,
           "h2d_copies_seen": len(big), "copy_kinds": sorted({str(c[0]) for c in copies})[:8],
           "kernel_ms_per_shard": {n: round(v[0] / 1e6 / k, 3) for n, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]}}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else None)
