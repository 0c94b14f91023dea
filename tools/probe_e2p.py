#!/usr/bin/env python3
"""Where the pipelined events -> proof loop spends host time beyond the proof: per shard the prefetch calls, trace generation, the proof, frees."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ziren_amd import abi, lib, prover

wl = bench.FibWorkload("shaped", int(sys.argv[1]) if len(sys.argv) > 1 else 21)
ctx = prover.Context(0)
fri = abi.FriConfig(1, 84, 16)
hp, pk, ch0 = wl.setup(ctx, fri, True)
lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
ds = wl.ds
ds.pin(ctx)
out = np.zeros(1 << 22, dtype=np.uint32)
born = ds.traces(ctx)
for _ in range(3):
    hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
t0 = time.perf_counter()
for _ in range(6):
    hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
print(f"resident: {(time.perf_counter() - t0) / 6 * 1e3:.3f} ms per proof")
for t in born:
    t.free()
pre = ds.prefetch(ctx)
acc = {"prefetch": 0.0, "traces": 0.0, "prove": 0.0, "free": 0.0}
n = 8
T0 = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter()
    nxt = ds.prefetch(ctx) if i + 1 < n else None
    t1 = time.perf_counter()
    born = ds.traces(ctx, pre)
    t2 = time.perf_counter()
    hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
    t3 = time.perf_counter()
    for t in born:
        t.free()
    t4 = time.perf_counter()
    pre = nxt
    if i >= 2:
        acc["prefetch"] += t1 - t0; acc["traces"] += t2 - t1; acc["prove"] += t3 - t2; acc["free"] += t4 - t3
print({k: round(v / (n - 2) * 1e3, 3) for k, v in acc.items()}, "ms per shard; total", round((time.perf_counter() - T0) / n * 1e3, 3))
