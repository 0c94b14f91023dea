#!/usr/bin/env python3
"""events -> device traces -> proof, pipelined (shard i+1's events queued on the DMA stream right before shard i's proof), alone — the
loop bench.py's `events_to_proof.pipelined` times, for a rocprofv3 trace (tools/profile_e2p.sh).  python tools/bench_e2p.py [shards] [log2 SHARD_SIZE] [resident]
`resident`: the same number of proofs on resident traces instead (the comparison trace)."""
import ctypes as C
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ziren_amd import abi, lib, prover

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = bench.FibWorkload("shaped", int(sys.argv[2]) if len(sys.argv) > 2 else 21)
resident = len(sys.argv) > 3 and sys.argv[3] == "resident"
ctx = prover.Context(0)
fri = abi.FriConfig(1, 84, 16)
hp, pk, ch0 = wl.setup(ctx, fri, True)
lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
ds = wl.ds
ds.pin(ctx)
out = np.zeros(1 << 22, dtype=np.uint32)


def one(pre):
    born = ds.traces(ctx, pre)
    proof = hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
    for t in born:
        t.free()
    return proof


one(ds.prefetch(ctx))
ctx.synchronize()
if resident:
    born = ds.traces(ctx)
    hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
    t0 = time.perf_counter()
    for i in range(n):
        hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
    ctx.synchronize()
else:
    pre = ds.prefetch(ctx)
    t0 = time.perf_counter()
    for i in range(n):
        nxt = ds.prefetch(ctx) if i + 1 < n else None
        one(pre)
        pre = nxt
    ctx.synchronize()
print(json.dumps({"mode": "resident" if resident else "pipelined", "shards": n, "ms_per_shard": round((time.perf_counter() - t0) / n * 1e3, 3), "event_bytes": ds.event_bytes()}))
