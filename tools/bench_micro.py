#!/usr/bin/env python3
"""Single-matrix micro-benchmarks of the commit path: coset LDE and Poseidon2 Merkle
commitment of one 2^k x 64 matrix resident in HBM, reported against the HBM roofline with the algorithmic byte
counts LDE = 12 n w and Merkle = 8 n w + 128 n (blow-up 2). Also the BASELINE config-2 shape (SYN-20, commit only).

  python tools/bench_micro.py [--log-rows 20 21 22] [--width 64] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from ziren_amd import prover, synth, field as F

HBM_PEAK = 8000.0


def kernel_ms(ctx, names):
    t = {n: ms for n, ms, _, _ in ctx.kernel_timings()}
    return sum(t.get(n, 0.0) for n in names)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, nargs="+", default=[20, 21, 22])
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    print(json.dumps(micro_bench(prover.Context(0), args.log_rows, args.width, args.reps), indent=1))


def micro_bench(ctx, log_rows=(20, 21, 22), width=64, reps=5, config2=True):
    """{"micro": [per 2^k x width matrix: LDE and Merkle separately], "config2": SYN-20 commit only} on `ctx` (bench.py carries it as `micro`)."""
    from ziren_amd import lib
    import ctypes as C
    args = argparse.Namespace(log_rows=list(log_rows), width=width, reps=reps)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    out = []
    rng = F.SplitMix64(0xC0FFEE)
    for k in args.log_rows:
        n, w = 1 << k, args.width
        m = F.to_monty(rng.uniform_field((n, w)))
        dm = ctx.upload(m)
        lde_ms, tree_ms, wall = [], [], []
        for _ in range(args.reps + 1):
            t0 = time.perf_counter()
            d = prover.pcs_commit(ctx, [dm], 1)
            wall.append((time.perf_counter() - t0) * 1e3)
            lde_ms.append(kernel_ms(ctx, ["lde_rows", "lde_cols_inverse", "lde_cols_forward"]))
            tree_ms.append(kernel_ms(ctx, ["hash_leaves", "hash_leaves_tree", "hash_rows", "compress_layer", "compress_layer_rowdig", "compress_small", "compress_tail"]))
            d.free()
        lde, tree = float(np.median(lde_ms[1:])), float(np.median(tree_ms[1:]))
        lde_bytes, tree_bytes = 12.0 * n * w, 8.0 * n * w + 128.0 * n
        rec = {"matrix": f"2^{k} x {w}", "lde_ms": round(lde, 3), "lde_GBps": round(lde_bytes / lde / 1e6, 1),
               "lde_frac_of_hbm_peak": round(lde_bytes / lde / 1e6 / HBM_PEAK, 4),
               "merkle_ms": round(tree, 3), "merkle_GBps": round(tree_bytes / tree / 1e6, 1),
               "merkle_frac_of_hbm_peak": round(tree_bytes / tree / 1e6 / HBM_PEAK, 4),
               "commit_wall_ms": round(float(np.median(wall[1:])), 3),
               "poseidon2_permutations": int(2 * n * ((w + 7) // 8) + 2 * n - 1),
               "Gperm_per_s": round((2 * n * ((w + 7) // 8) + 2 * n - 1) / tree / 1e6, 2)}
        out.append(rec)
        dm.free()
        ctx.trim()
    if not config2:
        lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2))
        return {"micro": out}
    # BASELINE config 2: SYN-20 main traces, commit only
    sh = synth.syn_shard(20)
    hp = prover.HipProver(sh.chips, __import__("ziren_amd.abi", fromlist=["abi"]).FriConfig(1, 84, 16), synth.NUM_PV_ELTS, ctx=ctx)
    traces = hp.upload_traces([c.trace for c in sh.chips])
    walls = []
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        data = hp.commit(sh.public_values, traces)
        walls.append((time.perf_counter() - t0) * 1e3)
        lib.load().zkm_main_data_free(ctx.h, data.handle)
    cells = sum(c.main_width << c.log_height for c in sh.chips)
    rows = sum(1 << c.log_height for c in sh.chips)
    cfg2 = {"workload": "SYN-20 main traces, MachineProver::commit only (BASELINE config 2)",
            "commit_ms": round(float(np.median(walls[1:])), 3), "cells": cells,
            "algorithmic_bytes": int(20 * cells + 128 * rows),
            "GBps": round((20 * cells + 128 * rows) / float(np.median(walls[1:])) / 1e6, 1)}
    for t in traces:
        t.free()
    ctx.trim()
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2))
    return {"micro": out, "config2": cfg2,
            "note": "one matrix resident in HBM, blow-up 2, medians of per-kernel HIP-event time (LDE = the three LDE kernels, Merkle = leaves + every tree level); "
                    "algorithmic bytes LDE 12 n w, Merkle 8 n w + 128 n (SURVEY 8d)"}


if __name__ == "__main__":
    main()
