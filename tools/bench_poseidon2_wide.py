#!/usr/bin/env python3
"""Trace generation of the recursion machine's Poseidon2Wide chip (degree 3) on the device: 2^k permutations -> 313-column rows.
Events (input, output) are made with the library's own batch permutation. Reports kernel time, rows/s and the HBM write rate
(4 * 313 bytes per row against the 8 TB/s peak)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from ziren_amd import field as F, lib, prover, recursion as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    n = 1 << args.log_rows
    ctx = prover.Context(0)
    inputs = F.to_monty(F.SplitMix64(7).uniform_field(16 * n)).reshape(n, 16)
    outputs = prover.poseidon2_permute_batch(ctx, inputs.copy())
    events = ctx.host_alloc((n * 32,))
    events.reshape(n, 32)[:, :16] = inputs
    events.reshape(n, 32)[:, 16:] = outputs
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    ks = []
    for it in range(args.steps + 1):
        m = ctx.tracegen_poseidon2_wide(events, args.log_rows)
        k = sum(ms for name, ms, _, _ in ctx.kernel_timings() if name == "tracegen_poseidon2_wide")
        if it == 0:   # the output columns are the events' outputs
            got = m.to_host()
            assert np.array_equal(got[:, 156:172], outputs) and np.array_equal(got[:, :16], inputs)
        m.free()
        if it:
            ks.append(k)
    k = float(np.mean(ks))
    nbytes = 4.0 * R.POSEIDON2_WIDE_WIDTH * n + 128.0 * n
    print(json.dumps({"workload": f"Poseidon2Wide (degree 3) trace generation, 2^{args.log_rows} permutations", "kernel_ms": round(k, 4),
                      "rows_per_s": round(n / (k * 1e-3), 1), "GBps": round(nbytes / k / 1e6, 1), "frac_of_hbm_peak": round(nbytes / k / 1e6 / 8000.0, 4)}))


if __name__ == "__main__":
    main()
