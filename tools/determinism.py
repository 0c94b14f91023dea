#!/usr/bin/env python3
"""Two lanes on one GPU prove the benchmarked shard forty times each: every proof must be the same words (a race between the main and the
side stream, or between contexts, would show as a differing proof), and the restated verifier accepts it.   gpurun -- 'python tools/determinism.py'"""
import os, sys, threading
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import numpy as np
import bench
from ziren_amd import abi, prover
fri = abi.FriConfig(1, 84, 16)
wl = bench.FibWorkload("shaped", 21)
lanes = []
for j in range(2):
    ctx = prover.Context(0)
    hp, pk, ch0 = wl.setup(ctx, fri, True)
    lanes.append((ctx, hp, pk, ch0, wl.resident_traces(ctx), np.zeros(1 << 22, dtype=np.uint32)))
ref = [None, None]
bad = []
def loop(j, n):
    ctx, hp, pk, ch0, tr, out = lanes[j]
    for i in range(n):
        p = hp.prove_shard(pk, wl.public_values, tr, ch0.copy(), out=out).copy()
        if ref[j] is None: ref[j] = p
        elif not np.array_equal(p, ref[j]): bad.append((j, i))
ts = [threading.Thread(target=loop, args=(j, 40)) for j in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
print("identical across lanes:", np.array_equal(ref[0], ref[1]), "mismatches:", bad)
bench.verify_or_die(wl, fri, lanes[0][3], ref[0], "determinism")
print("verified")
