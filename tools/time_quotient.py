#!/usr/bin/env python3
"""Per-kernel time of the benchmarked shard's proof WITHOUT verification — for experiments whose generated kernels compute something else
than the constraints (tools/ab_uniforms.sh times the quotient kernels with their wave-uniform arithmetic taken out: the proof is garbage,
the kernel's duration is what a table of precomputed uniform values would give).  python tools/time_quotient.py [proofs]"""
import ctypes as C
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ziren_amd import abi, lib, prover

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = bench.FibWorkload("shaped", 21)
ctx = prover.Context(0)
fri = abi.FriConfig(1, 84, 16)
hp, pk, ch0 = wl.setup(ctx, fri, True)
L = lib.load()
L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(0))
born = wl.resident_traces(ctx)
out = np.zeros(1 << 22, dtype=np.uint32)
L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2))
acc = {}
for i in range(n):
    hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
    for name, ms, calls, nbytes in ctx.kernel_timings():
        a = acc.setdefault(name, [0.0, 0])
        a[0] += ms; a[1] += calls
print(json.dumps({k: {"ms": round(v[0] / n, 3), "launches": v[1] // n} for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]}))
