#!/usr/bin/env python3
"""A coherent core shard end to end on the device: a program executed by ziren_amd/miniexec.py (the reference's executor is
Rust and cannot run here) -> CpuEvents + per-chip events in pinned host memory -> device trace generation for Cpu, Program,
the instruction chips, MemoryLocal, Global and Byte -> commit + open with the recorded AIRs (generated quotient kernels).

  python tools/bench_core_shard.py [--log-cycles 18] [--steps 3]

Every lookup of the shard has both of its sides in real chips (the local cumulative sum is zero; the Global chip's curve sum is the
proof's global_cumulative_sum); tests/test_cpu_shard.py verifies the same shard shape against the oracle."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from ziren_amd import abi, chips, events as E, field as F, lib, miniexec as M, prover, synth

PC_BASE, SHARD = 0x1000, 1


def log2_rows(n):
    h = 16
    while h < n:
        h <<= 1
    return h.bit_length() - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-cycles", type=int, default=18)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=1, help="shards in flight on the GPU (one context + host thread each): events -> proof end to end")
    args = ap.parse_args()
    n = 1 << args.log_cycles
    t0 = time.perf_counter()
    prog, rec0, pv = M.run(n - 40, seed=1, shard=SHARD, pc_base=PC_BASE, halt=True)   # + 35 or 36 cycles of commits and halt: 2^k Cpu rows
    rec = M.add_dependencies(rec0)
    exec_s = time.perf_counter() - t0
    class Lane:
        """One prover context with its own pinned copy of the record: events -> traces -> proof, start to end."""

        def __init__(self):
            ctx = self.ctx = prover.Context(0)

            def pin(ev):   # the executor's event vectors, in page-locked host memory
                pinned = ctx.host_alloc((max(len(ev), 1) * (ev.dtype.itemsize // 4),))
                pinned[:len(ev) * (ev.dtype.itemsize // 4)] = ev.view(np.uint32).reshape(-1)
                return pinned[:len(ev) * (ev.dtype.itemsize // 4)].view(ev.dtype)

            work = [("cpu", pin(rec.cpu), chips.record_cpu_chip)]
            work += [(c, pin(rec.alu[c]), None) for c in sorted(E.CHIP_NAMES)]
            work += [("jump", pin(rec.jump), chips.record_jump_chip), ("mov_cond", pin(rec.mov_cond), chips.record_mov_cond_chip),
                     ("branch", pin(rec.branch), chips.record_branch_chip), ("mul", pin(rec.mul), chips.record_mul_chip),
                     ("divrem", pin(rec.divrem), chips.record_divrem_chip),
                     ("memory_instrs", pin(rec.mem_instr), chips.record_memory_instrs_chip),
                     ("syscall_instrs", pin(rec.syscall), chips.record_syscall_instrs_chip), ("misc_instrs", pin(rec.misc), chips.record_misc_instrs_chip),
                     ("memory_local", pin(rec.memory_local), chips.record_memory_local_chip),
                     ("global", pin(M.global_lookup_events(rec.memory_local)), chips.record_global_chip)]
            self.work = work
            self.heights = [log2_rows(-(-len(ev) // 4) if c == "memory_local" else len(ev)) for c, ev, _ in work]
            self.recs = [chips.record_chip(c, lh) if rc is None else rc(lh) for (c, _, rc), lh in zip(work, self.heights)]
            self.plh = log2_rows(len(prog))
            self.recs += [chips.record_byte_chip(0), chips.record_program_chip(self.plh, 1)]
            self.pprog = pin(prog)
            self.hp = prover.HipProver(self.recs, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=True)
            self.pk = self.hp.setup([ctx.tracegen_byte_table(), ctx.tracegen_program(self.pprog, PC_BASE, self.plh)], [0, 0], F.to_monty(PC_BASE),
                                    F.to_monty(np.zeros(14, dtype=np.uint64)))
            self.ch0 = prover.new_challenger()
            self.pk.observe_into(self.ch0)
            self.out = np.zeros(1 << 22, dtype=np.uint32)

        def shard(self):
            ctx = self.ctx
            t0 = time.perf_counter()
            blu = ctx.byte_lookups()
            born = []
            tg_kernel = 0.0
            for (c, ev, _), lh in zip(self.work, self.heights):
                if c == "cpu":
                    cpu_trace, program_mults = ctx.tracegen_cpu_and_program(ev, self.pprog, PC_BASE, SHARD, lh, self.plh, blu)
                    born.append(cpu_trace)
                elif c == "jump":
                    born.append(ctx.tracegen_jump(ev, lh))
                elif c == "mov_cond":
                    born.append(ctx.tracegen_mov_cond(ev, lh))
                elif c == "branch":
                    born.append(ctx.tracegen_branch(ev, lh, blu))
                elif c == "mul":
                    born.append(ctx.tracegen_mul(ev, lh, blu))
                elif c == "divrem":
                    born.append(ctx.tracegen_divrem(ev, lh, blu))
                elif c == "memory_local":
                    born.append(ctx.tracegen_memory_local(ev, lh))
                elif c == "global":
                    born.append(ctx.tracegen_global(ev, lh, blu))
                elif c == "memory_instrs":
                    born.append(ctx.tracegen_memory_instrs(ev, lh, blu))
                elif c == "syscall_instrs":
                    born.append(ctx.tracegen_syscall_instrs(ev, lh))
                elif c == "misc_instrs":
                    born.append(ctx.tracegen_misc_instrs(ev, lh, blu))
                else:
                    born.append(ctx.tracegen_alu(c, ev, lh, blu))
                tg_kernel += sum(ms for name, ms, _, _ in ctx.kernel_timings() if name.startswith("tracegen"))
            born.append(ctx.tracegen_byte_mults(blu))
            born.append(program_mults)
            blu.free()
            t1 = time.perf_counter()
            proof = self.hp.prove_shard(self.pk, pvs, born, self.ch0.copy(), out=self.out)
            t2 = time.perf_counter()
            phases = dict(ctx.last_timings())
            kern = {nm: (round(ms, 3), calls) for nm, ms, calls, _ in ctx.kernel_timings()}
            for m in born:
                m.free()
            return {"tracegen_ms": (t1 - t0) * 1e3, "tracegen_kernel_ms": tg_kernel, "prove_ms": (t2 - t1) * 1e3, "phases": phases, "kernels": kern,
                    "proof_words": int(len(proof))}

    fri = abi.FriConfig(1, 84, 16)
    pvs = M.public_values(pv)
    lanes = [Lane() for _ in range(max(1, args.inflight))]
    for lane in lanes:
        lib.load().zkm_ctx_set_kernel_timing(lane.ctx.h, C.c_int(1))
        lane.shard()     # warm-up
    res = [lanes[0].shard() for _ in range(args.steps)]
    pipelined = None
    if len(lanes) > 1:   # free-running lanes: one lane's event upload (copy engine) overlaps another lane's proving (compute)
        import threading
        t0 = time.perf_counter()
        ts = [threading.Thread(target=lambda ln=ln: [ln.shard() for _ in range(args.steps)]) for ln in lanes]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        wall = time.perf_counter() - t0
        pipelined = {"lanes": len(lanes), "shards": len(lanes) * args.steps, "ms_per_shard": round(wall / (len(lanes) * args.steps) * 1e3, 3),
                     "shards_per_s": round(len(lanes) * args.steps / wall, 3)}
    recs, proof_words = lanes[0].recs, res[-1]["proof_words"]
    work = lanes[0].work
    r = res[-1]
    cells = sum((1 << c.log_height) * (c.main_width + 4 * c.perm_ext_width + 8) for c in recs)
    event_bytes = sum(len(ev) * ev.dtype.itemsize for _, ev, _ in work)
    print(json.dumps({"workload": f"CORE-{args.log_cycles}: 2^{args.log_cycles} executed instructions (Cpu rows) of a generated program; Cpu, Program, "
                                  "AddSub, Bitwise, Lt, ShiftLeft, ShiftRight, CloClz, Mul, DivRem, Branch, Jump, MovCond, MemoryInstrs, SyscallInstrs, MiscInstrs, MemoryLocal, Global, Byte",
                      "executor_seconds_python": round(exec_s, 1), "program_instructions": int(len(prog)), "event_bytes": int(event_bytes),
                      "tracegen_ms": round(float(np.mean([x["tracegen_ms"] for x in res])), 3),
                      "tracegen_kernel_ms": round(float(np.mean([x["tracegen_kernel_ms"] for x in res])), 3),
                      "prove_ms": round(float(np.mean([x["prove_ms"] for x in res])), 3),
                      "events_to_proof_ms": round(float(np.mean([x["tracegen_ms"] + x["prove_ms"] for x in res])), 3), "pipelined": pipelined,
                      "committed_cells": cells, "proof_words": proof_words,
                      "chips": {c.name: {"rows": 1 << c.log_height, "main": c.main_width, "perm_ext": c.perm_ext_width,
                                         "constraints": c.num_constraints, "lookups": len(c.sends) + len(c.receives)} for c in recs},
                      "phases_ms": {nm: round(v, 3) for nm, v in r["phases"].items()}, "kernels_ms": r["kernels"]}, indent=1))


if __name__ == "__main__":
    main()
