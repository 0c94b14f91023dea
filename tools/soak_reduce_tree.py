#!/usr/bin/env python3
"""Soak of the recursion-tree reduce: the same 32-leaf tree R times on two lanes of one GPU; wall time per tree, the contexts' pool sizes
and the device's used memory after every tree — flat after the first few means nothing leaks through the handle table, the failing-call
give-back or the events' prefetch buffers.   python tools/soak_reduce_tree.py [--trees 30] [--leaves 32]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=30)
    ap.add_argument("--leaves", type=int, default=32)
    args = ap.parse_args()
    import ctypes as C
    import bench_reduce_tree as B
    from ziren_amd import farm, field as F, lib, prover, reduce as RD
    ctxs = [prover.Context(0), prover.Context(0)]
    lanes = [RD.ReduceLane(c) for c in ctxs]
    for c in ctxs:
        lib.load().zkm_ctx_set_host_wait(c.h, C.c_int(1))
    tree = RD.ReduceTree(RD.TreePlan(1, 0, 0), B.device_permute(ctxs[0]))
    f = farm.Farm()
    import re
    import subprocess

    def vram_used_mb():           # the device's used memory as the driver reports it (rocm-smi; a second HIP runtime cannot be loaded beside the library's)
        try:
            out = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True, timeout=30).stdout
            m = re.search(r"Used Memory \(B\): (\d+)", out)
            return int(m.group(1)) >> 20 if m else None
        except (OSError, subprocess.SubprocessError):
            return None
    rows, first = [], None
    for k in range(args.trees):
        core = np.random.default_rng(k).integers(0, F.P, (args.leaves, RD.CHILD_WORDS), dtype=np.uint64)
        t0 = time.perf_counter()
        streams, words = tree.run(f, lanes, core)
        dt = time.perf_counter() - t0
        digest = words[-1][0][24:].tolist()
        if k == 0:
            first = digest
        rows.append({"tree": k, "wall_ms": round(dt * 1e3, 2), "pool_MB": [c.memory_held() >> 20 for c in ctxs], "device_used_MB": vram_used_mb() if k % 5 == 4 or k < 4 else None})
    # the first tree again: a different tree in between changes nothing of a later one
    core = np.random.default_rng(0).integers(0, F.P, (args.leaves, RD.CHILD_WORDS), dtype=np.uint64)
    again = tree.run(f, lanes, core)[1][-1][0][24:].tolist()
    out = {"what": f"{args.trees} trees of {args.leaves} leaves back to back ({sum(len(n) for *_, n in tree.layers(args.leaves))} recursion shards each), two lanes, one MI355X",
           "wall_ms": {"first": rows[0]["wall_ms"], "median_after_the_third": float(np.median([r["wall_ms"] for r in rows[3:]])), "max_after_the_third": max(r["wall_ms"] for r in rows[3:])},
           "pool_MB_after_tree": {"3": rows[3]["pool_MB"], "last": rows[-1]["pool_MB"]}, "device_used_MB_after_tree": {"3": rows[3]["device_used_MB"], "last": rows[-1]["device_used_MB"]},
           "same_root_digest_for_the_same_inputs": again == first, "rows": rows}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
