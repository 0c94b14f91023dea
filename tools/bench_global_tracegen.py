#!/usr/bin/env python3
"""Global chip trace generation on the device at scale: 2^k random GlobalLookupEvents -> the 99-column trace (lift_x per row, the
curve-point scan, the accumulation columns). Reports the per-kernel times the context records.

  python tools/bench_global_tracegen.py [--log-events 20] [--steps 5]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from ziren_amd import lib, miniexec as M, prover


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-events", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    n = 1 << args.log_events
    rng = np.random.default_rng(1)
    ev = np.zeros(n, dtype=M.GLOBAL_LOOKUP_EVENT)
    ev["message"] = rng.integers(0, 1 << 24, size=(n, 7))
    ev["message"][:, 0] = 1
    ev["message"][:, 3:] = rng.integers(0, 256, size=(n, 4))
    ev["is_receive"] = rng.integers(0, 2, size=n)
    ev["kind"] = 1
    ctx = prover.Context(0)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    blu = ctx.byte_lookups()
    ctx.tracegen_global(ev, -1, blu).free()
    wall, kern = [], {}
    for _ in range(args.steps):
        t0 = time.perf_counter()
        m = ctx.tracegen_global(ev, -1, blu)
        wall.append((time.perf_counter() - t0) * 1e3)
        for name, ms, calls, _ in ctx.kernel_timings():
            kern.setdefault(name, []).append(ms)
        m.free()
    blu.free()
    k = {name: round(float(np.mean(v)), 3) for name, v in kern.items()}
    total = sum(k.values())
    print(json.dumps({"workload": f"GLOBAL-{args.log_events}: 2^{args.log_events} GlobalLookupEvents -> {n} x 99 trace", "wall_ms": round(float(np.mean(wall)), 3),
                      "kernels_ms": k, "kernel_total_ms": round(total, 3), "rows_per_s": round(n / total * 1e3), "trace_bytes": n * 99 * 4,
                      "note": "compute-bound integer work: ~2 curve candidates per row, each a norm (6 Frobenius maps + 6 products) and, when it is a "
                              "square, a 31-bit exponentiation in the septic extension; the scan costs two inversions per point"}, indent=1))


if __name__ == "__main__":
    main()
