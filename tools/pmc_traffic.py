#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: they do not fit one pass).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/f -o f -- python bench.py --log-rows 22 --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/w -o w -- python bench.py --log-rows 22 --steps 1 --warmup 0 --no-cpu-baseline
  python tools/pmc_traffic.py out/f/f_results.db out/w/w_results.db profiles/r01_syn22_hbm_traffic.json

Units: the counters are in KiB. gfx950 correction (guide, section HBM): FETCH_SIZE reports half the bytes of a
wide coalesced stream, so reads are doubled; WRITE_SIZE is taken as is (it matches the digest bytes the tree
kernels write to within 2 %)."""
import json
import sqlite3
import sys


def load(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? "
                      "group by kernel_name", (counter,)).fetchall()
    return {k.split("(")[0].replace("void ", ""): (v, n) for k, v, n in rows}


def main(fetch_db, write_db, out_path, tag="syn22", proofs=1):
    f, w = load(fetch_db, "FETCH_SIZE"), load(write_db, "WRITE_SIZE")
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench              # the same digest bench.py checks before it quotes a figure from this file
    commit = os.environ.get("ZKM_COMMIT") or None       # the GPU box's snapshot has no .git: the caller passes the commit it snapshotted
    if commit is None:
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except OSError:
            commit = None
    proofs = int(proofs)
    out = {"workload": f"{tag}: {proofs} shard proof(s) of bench.py's workload of that tag (tools/profile_r04.sh)", "steps": proofs, "csrc_digest": bench.csrc_digest(), "commit": commit,
           "correction": "read bytes = 2 * FETCH_SIZE * 1024, write bytes = WRITE_SIZE * 1024", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fv, fn = f.get(k, (0, 0))
        wv, wn = w.get(k, (0, 0))
        launches = max(fn, wn, 1)
        total = 2 * fv * 1024 + wv * 1024
        out["kernels"][k] = {"launches": launches, "fetch_size_kib": fv, "write_size_kib": wv,
                             "hbm_bytes_total": total, "hbm_bytes_per_launch": total / launches}
    json.dump(out, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main(*sys.argv[1:6])
