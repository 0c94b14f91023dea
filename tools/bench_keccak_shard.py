#!/usr/bin/env python3
"""The Keccak precompile shard of a run (BASELINE config 4's workload shape: examples/keccak-precompile) end to end on the device: a
program that hashes `--calls` messages with KECCAK_SPONGE is executed by ziren_amd/miniexec.py; the deferred precompile shard
(SyscallPrecompile, KeccakSponge, MemoryLocal, Global, Byte, Program) is generated on the device from its events and proven.

  python tools/bench_keccak_shard.py [--calls 1500] [--steps 3]

KeccakSponge is 3531 columns wide and 24 rows per 36-word block: the shard is dominated by that one matrix. Its constraint program
(114 324 instructions) is too long for a straight-line kernel and runs in the library's bytecode interpreter."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from ziren_amd import abi, chips, field as F, lib, miniexec as M, prover, synth

import machine_lib as ML


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=1500)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    t0 = time.perf_counter()
    m = M.run_machine(8 * args.calls, seed=2, keccak_calls=args.calls)
    exec_s = time.perf_counter() - t0
    k = [i for i, s in enumerate(m.shards) if s.kind == "precompile"][-1]
    rec = m.shards[k].record
    ctx = prover.Context(0)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    fri = abi.FriConfig(1, 84, 16)
    zero = F.to_monty(np.array(chips.SEPTIC_START_X + chips.SEPTIC_START_Y, dtype=np.uint64)).astype(np.uint32)
    pvs = ML.shard_public_values(m.shards[k])
    # the KeccakSponge matrix alone: events -> 65536 x 3531 on the device
    kec_ms = []
    for _ in range(4):
        t0 = time.perf_counter()
        born = ctx.tracegen_keccak_sponge(rec.keccak_sponge)
        ctx.synchronize()
        kec_ms.append(((time.perf_counter() - t0) * 1e3, sum(ms for nm, ms, _, _ in ctx.kernel_timings() if nm.startswith("tracegen"))))
        kec_bytes = 4 * born.height * born.width
        born.free()
    kec_wall, kec_kernel = min(kec_ms[1:])
    res = []
    hp = pk = None
    for step in range(args.steps + 1):
        t0 = time.perf_counter()
        dev = ML.Device(ctx)
        dcs = ML.build_shard(dev, m, k)
        ctx.synchronize()
        t1 = time.perf_counter()
        tg = {nm: round(ms, 3) for nm, ms, _, _ in ctx.kernel_timings() if nm.startswith("tracegen")}
        if hp is None:
            hp = prover.HipProver(dcs, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=True)
            pk = hp.setup([ctx.tracegen_byte_table(), ctx.tracegen_program(m.program, m.pc_base, dcs[-1].log_height)], [0, 0], F.to_monty(m.pc_base), zero)
        ch = prover.new_challenger()
        pk.observe_into(ch)
        t2 = time.perf_counter()
        proof = hp.prove_shard(pk, pvs, [c.trace for c in dcs], ch)
        t3 = time.perf_counter()
        phases = dict(ctx.last_timings())
        kern = {nm: (round(ms, 3), calls) for nm, ms, calls, _ in ctx.kernel_timings()}
        cells = sum(c.trace.height * c.trace.width for c in dcs)
        shape = {c.name: [c.trace.height, c.trace.width] for c in dcs}
        for c in dcs:
            c.trace.free()
        dev.blu.free()
        if step:
            res.append({"tracegen_ms": (t1 - t0) * 1e3, "prove_ms": (t3 - t2) * 1e3, "phases": phases, "kernels": kern, "tracegen_kernels": tg,
                        "proof_words": int(len(proof))})
    best = min(res, key=lambda r: r["prove_ms"])
    blocks = len(rec.keccak_sponge)
    out = {"workload": f"KECCAK-{args.calls}: {args.calls} KECCAK_SPONGE calls ({blocks} blocks of 36 words) -> the deferred precompile shard",
           "executor_seconds_python": round(exec_s, 1), "chips": shape, "committed_cells": int(cells), "blocks": blocks,
           "shard_tracegen_ms_with_python_event_packing": round(best["tracegen_ms"], 3),
           "keccak_sponge_tracegen": {"wall_ms": round(kec_wall, 3), "kernel_ms": round(kec_kernel, 3), "trace_bytes": kec_bytes,
                                      "GBps": round(kec_bytes / max(kec_kernel, 1e-9) / 1e6, 1)}, "prove_ms": round(best["prove_ms"], 3),
           "keccak_blocks_per_s_proven": round(blocks / (best["prove_ms"] + kec_wall) * 1e3, 1), "phases_ms": best["phases"],
           "kernels_ms": best["kernels"], "proof_words": best["proof_words"],
           "note": "the KeccakSponge quotient runs in the bytecode interpreter (its 114 324-instruction program is not compiled to a straight-line kernel)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
