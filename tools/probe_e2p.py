#!/usr/bin/env python3
"""Where the events -> proof path spends host time beyond the proof (one shaped fibonacci shard, events prefetched): per tracegen call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ziren_amd import abi, prover, lib
import ctypes as C

wl = bench.FibWorkload("shaped", int(sys.argv[1]) if len(sys.argv) > 1 else 21)
ctx = prover.Context(0)
fri = abi.FriConfig(1, 84, 16)
hp, pk, ch0 = wl.setup(ctx, fri, True)
lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
ds = wl.ds
ds.pin(ctx)
out = np.zeros(1 << 22, dtype=np.uint32)
for rep in range(3):
    pre = ds.prefetch(ctx)
    ctx.synchronize()
    for d in pre.values():
        lib.load().zkm_matrix_wait  # noqa
    time.sleep(0.05)   # the copies have landed
    t0 = time.perf_counter()
    blu = ctx.byte_lookups()
    t_blu = time.perf_counter() - t0
    marks = []
    born = []
    orig = {}
    import types
    # time ds.traces as a whole and the proof
    t1 = time.perf_counter()
    born = ds.traces(ctx, pre)
    t2 = time.perf_counter()
    proof = hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
    t3 = time.perf_counter()
    for t in born:
        t.free()
    t4 = time.perf_counter()
    print(f"rep {rep}: byte_lookups {t_blu*1e3:.3f} ms, traces() {1e3*(t2-t1):.3f} ms, prove {1e3*(t3-t2):.3f} ms, free {1e3*(t4-t3):.3f} ms")
    blu.free()
# per call
pre = ds.prefetch(ctx)
time.sleep(0.05)
blu = ctx.byte_lookups()
for name, ev, lh, _ in ds.work:
    e = pre.get(name, ev)
    t0 = time.perf_counter()
    if name == "Cpu":
        m, pm = ctx.tracegen_cpu_and_program(e, ds.machine.program, ds.machine.pc_base, ds.shard_no, lh, ds.plh, blu)
    elif name in bench.__dict__.get("_ALU", {}) or name in ("AddSub", "Bitwise", "Lt", "ShiftLeft", "ShiftRight", "CloClz"):
        from ziren_amd import fibfast
        m = ctx.tracegen_alu(fibfast._ALU[name], e, lh, blu)
    else:
        continue
    dt = time.perf_counter() - t0
    print(f"  {name:14s} 2^{lh} rows, {len(ev)} events: {dt*1e3:.3f} ms")
