#!/usr/bin/env python3
"""Shard of the reference's real chips — six ALU chips, Jump, MovCond, Branch, Mul, DivRem and Byte (recorded AIRs, ziren_amd/chips.py): events -> device traces ->
commit + open, timed per phase and per kernel. Real constraint programs and lookup shapes instead of the SYN stand-ins.

  python tools/bench_alu_shard.py [--log-rows 21] [--steps 3]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from ziren_amd import abi, chips, events as E, field as F, lib, prover, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=21)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--interpreter", action="store_true")
    args = ap.parse_args()
    k = args.log_rows
    spec = [(E.CHIP_ADD_SUB, k), (E.CHIP_BITWISE, k - 1), (E.CHIP_LT, k - 1), (E.CHIP_SHIFT_LEFT, k - 2), (E.CHIP_SHIFT_RIGHT, k - 2),
            (E.CHIP_CLO_CLZ, k - 3)]
    other = [("jump", k - 3, E.synthetic_jump_events, chips.record_jump_chip), ("mov_cond", k - 3, E.synthetic_mov_cond_events, chips.record_mov_cond_chip),
             ("branch", k - 3, E.synthetic_branch_events, chips.record_branch_chip), ("mul", k - 2, E.synthetic_mul_events, chips.record_mul_chip),
             ("divrem", k - 3, E.synthetic_divrem_events, chips.record_divrem_chip)]
    fill = 0.6
    streams = {c: E.synthetic_alu_events(c, int((1 << lh) * fill)) for c, lh in spec}
    ostreams = {name: gen(int((1 << lh) * fill)) for name, lh, gen, _ in other}
    # the events the executor derives from these instructions (crates/core/executor/src/dependencies.rs)
    lt_dep, add_dep = E.branch_dependencies(ostreams["branch"])
    div_add, div_mul, div_lt = E.divrem_dependencies(ostreams["divrem"])
    ostreams["mul"] = np.concatenate([ostreams["mul"][:int((1 << (k - 2)) * 0.5)], div_mul])
    streams[E.CHIP_SHIFT_RIGHT] = np.concatenate([streams[E.CHIP_SHIFT_RIGHT], E.cloclz_dependencies(streams[E.CHIP_CLO_CLZ])])
    streams[E.CHIP_ADD_SUB] = np.concatenate([streams[E.CHIP_ADD_SUB], E.jump_dependencies(ostreams["jump"]), add_dep, div_add])
    streams[E.CHIP_LT] = np.concatenate([streams[E.CHIP_LT][:int((1 << (k - 1)) * 0.35)], lt_dep, div_lt])
    recs = [chips.record_chip(c, lh) for c, lh in spec] + [rec(lh) for _, lh, _, rec in other] + [chips.record_byte_chip(0)]
    ctx = prover.Context(0)

    def pin(ev):   # the executor's event vectors, in page-locked host memory
        pinned = ctx.host_alloc((len(ev) * (ev.dtype.itemsize // 4),))
        pinned[...] = ev.view(np.uint32).reshape(-1)
        return pinned.view(ev.dtype)

    evs = [(c, pin(streams[c]), lh) for c, lh in spec] + [(name, pin(ostreams[name]), lh) for name, lh, _, _ in other]
    fri = abi.FriConfig(1, 84, 16)
    hp = prover.HipProver(recs, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=not args.interpreter)
    pv = F.to_monty(F.SplitMix64(3).uniform_field(synth.PROOF_MAX_NUM_PVS))
    pv[synth.NUM_PV_ELTS:] = 0
    pk = hp.setup([ctx.tracegen_byte_table()], [0], F.to_monty(0x400000), F.to_monty(F.SplitMix64(4).uniform_field(14)))
    ch0 = prover.new_challenger()
    pk.observe_into(ch0)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    out = np.zeros(1 << 22, dtype=np.uint32)
    res = []
    for it in range(args.steps + 1):
        t0 = time.perf_counter()
        blu = ctx.byte_lookups()
        born = []
        for c, ev, lh in evs:
            if c == "jump":
                born.append(ctx.tracegen_jump(ev, lh))
            elif c == "mov_cond":
                born.append(ctx.tracegen_mov_cond(ev, lh))
            elif c == "branch":
                born.append(ctx.tracegen_branch(ev, lh, blu))
            elif c == "mul":
                born.append(ctx.tracegen_mul(ev, lh, blu))
            elif c == "divrem":
                born.append(ctx.tracegen_divrem(ev, lh, blu))
            else:
                born.append(ctx.tracegen_alu(c, ev, lh, blu))
        born.append(ctx.tracegen_byte_mults(blu))
        blu.free()
        t1 = time.perf_counter()
        proof = hp.prove_shard(pk, pv, born, ch0.copy(), out=out)
        t2 = time.perf_counter()
        phases = dict(ctx.last_timings())
        kern = {n: (round(ms, 3), calls) for n, ms, calls, _ in ctx.kernel_timings()}
        for m in born:
            m.free()
        if it:
            res.append({"tracegen_ms": (t1 - t0) * 1e3, "prove_ms": (t2 - t1) * 1e3, "phases": phases, "kernels": kern})
    r = res[-1]
    cells = sum((1 << c.log_height) * (c.main_width + 4 * c.perm_ext_width + 8) for c in recs)
    print(json.dumps({"workload": f"CORE11-{k}: AddSub 2^{k}, Bitwise/Lt 2^{k-1}, ShiftLeft/ShiftRight/Mul 2^{k-2}, CloClz/Jump/MovCond/Branch/DivRem 2^{k-3}, "
                                  "Byte 2^16; 60% filled plus the executor's dependency events",
                      "tracegen_ms": round(float(np.mean([x["tracegen_ms"] for x in res])), 3),
                      "prove_ms": round(float(np.mean([x["prove_ms"] for x in res])), 3),
                      "committed_cells": cells, "proof_words": int(len(proof)),
                      "chips": {c.name: {"rows": 1 << c.log_height, "main": c.main_width, "perm_ext": c.perm_ext_width,
                                         "constraints": c.num_constraints, "lookups": len(c.sends) + len(c.receives)} for c in recs},
                      "phases_ms": {n: round(v, 3) for n, v in r["phases"].items()}, "kernels_ms": r["kernels"]}, indent=1))


if __name__ == "__main__":
    main()
