#!/usr/bin/env python3
"""Benchmark of the shard-prover hot path (BASELINE.json metric: shard-proofs/sec, 2^22-row trace).

A "step" is one full shard proof — MachineProver::commit + open (crates/stark/src/prover.rs:258-653) —
of a SYN-k shard (a Cpu-like chip at 2^k rows plus seven smaller chips, core FRI
parameters: blowup 2, 84 queries, 16 PoW bits) whose traces are already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--log-rows 22]

N > 1 is launched by the driver through torch.distributed.run, one process per GPU; shards are
independent (prove.rs:492-497), so every rank proves its own shards and there is no data-path
collective: scaling is weak, value = shards proven by all ranks / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # before anything initialises the HIP runtime (torch.distributed for N > 1 would): ziren_amd/lib.py

from ziren_amd import abi, lib, prover, synth

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VECTOR_TFLOPS = 78.6  # MI355X FP64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (MI355X_MICROARCH.md)


def poseidon2_isa():
    """Dynamic VALU instructions per permutation as the hardware counts them (SQ_INSTS_VALU over tools/ubench_p2's kernels,
    profiles/r03_poseidon2_isa.json, written by tools/profile_r03.sh -> tools/pmc_poseidon2.py); None when the profile is missing."""
    path = os.path.join(ROOT, "profiles", "r03_poseidon2_isa.json")
    if not os.path.exists(path):
        return None
    return json.load(open(path))


PROVER_SOURCES = ("kb31.cuh", "lde.cuh", "merkle.cuh", "open.cuh", "poseidon2.cuh", "poseidon2_constants.inc", "poseidon2_f64.cuh", "quotient_args.cuh",
                  "stark.cuh", "host_pcs.hpp", "host_open.hpp")


def csrc_digest():
    """sha256 over the sources a shard proof runs: the kernels (field, LDE, Merkle / Poseidon2, quotient, opening) and the host code that
    decides their launches (commit, open). PMC profiles record it, so a traffic figure is only quoted for the code it was measured on.
    The trace generators and the API glue are not part of it: they launch nothing inside a proof."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ziren_amd", "csrc")
    for name in PROVER_SOURCES:
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def usable_cores():
    """(cores this process may actually use, logical CPUs the host shows, why they differ). A container sees every logical CPU of the host
    (os.cpu_count) but is scheduled under a cgroup CPU quota: on the GPU boxes of this pool cpu.max is 16 CPUs' worth of time on a
    256-thread host, and running more threads than the quota makes every OpenMP loop slower (measured: SYN-18 on the restatement takes
    4.5 s with 16 threads, 6.5 s with 64, 58.8 s with 256 — profiles/r03_cpu_scaling.json), so the quota is the core count."""
    shown = os.cpu_count() or 1
    n, why = shown, None
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                n, why = q, f"cgroup cpu.max = {quota}/{period}: {q} CPUs of the host's {shown} logical CPUs"
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and quota // period < n:
                n, why = max(1, quota // period), f"cgroup cfs quota {quota}/{period}"
        except (OSError, ValueError):
            pass
    return n, shown, why


def cpu_baseline(log_rows_sample, fri):
    """Time the CPU restatement (oracle, kind 'port') proving one SYN shard of the sample size, on every core this process may use
    (usable_cores: the cgroup quota, not the host's CPU count)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    L.orc_lde_seconds.restype = C.c_double
    threads, shown, why = usable_cores()
    L.orc_set_num_threads(threads)
    sh = synth.syn_shard(log_rows_sample)
    pk = O.Pk([], [], sh.pc_start, sh.initial_global_cumulative_sum, fri.log_blowup)
    ch = O.new_challenger()
    pk.observe_into(ch)
    L.orc_lde_seconds(C.c_int(1))
    t0 = time.time()
    O.prove_shard(pk, sh.chips, [c.trace for c in sh.chips], sh.public_values, fri, synth.NUM_PV_ELTS, ch)
    wall = time.time() - t0
    return wall, float(L.orc_lde_seconds(C.c_int(0))), threads, shown, why


def fib_leg(device, fri, log_cycles, steps, specialize=True):
    """BASELINE.json's own workload beside the synthetic default: a full shard of the fibonacci guest (examples/fibonacci; the loop's
    closed-form events, ziren_amd/fibfast.py — event for event what the executor restatement gives), real chips (Cpu, AddSub, Lt, Mul,
    Branch, DivRem, MemoryLocal, Global, Byte, Program with the recorded AIRs), 2^log_cycles cycles. Two rates: `prove` = commit + open with
    the traces resident in HBM (what `value` of the main line means for SYN), and `events_to_proof` = device trace generation from the
    executor's events in (pageable) host memory + the proof, the reference's prove-a-record step (crates/core/machine/src/utils/prove.rs:484-497)."""
    from ziren_amd import fibfast, field as F, chips as CH
    t0 = time.perf_counter()
    mach = fibfast.full_shard(log_cycles)
    ds = fibfast.DeviceShard(mach)
    gen_s = time.perf_counter() - t0
    ctx = prover.Context(device)
    hp = prover.HipProver(ds.chips, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=specialize)
    zero_digest = F.to_monty(np.array(CH.SEPTIC_START_X + CH.SEPTIC_START_Y, dtype=np.uint64)).astype(np.uint32)
    pk = hp.setup(ds.preprocessed(ctx), [0, 0], F.to_monty(mach.pc_base), zero_digest)
    ch0 = prover.new_challenger()
    pk.observe_into(ch0)
    out = np.zeros(1 << 22, dtype=np.uint32)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2))
    born = ds.traces(ctx)
    hp.prove_shard(pk, ds.public_values, born, ch0.copy(), out=out)          # warm-up
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        proof = hp.prove_shard(pk, ds.public_values, born, ch0.copy(), out=out)
    ctx.synchronize()
    prove_ms = (time.perf_counter() - t0) / steps * 1e3
    phases = {n: round(ms, 3) for n, ms in ctx.last_timings()}
    kernels = {n: {"ms": round(ms, 3), "launches": calls} for n, ms, calls, _ in sorted(ctx.kernel_timings(), key=lambda t: -t[1])}
    n_words = int(len(proof))
    for t in born:
        t.free()
    pageable = None
    for pinned in (False, True):
        if pinned:
            pageable = (e2p_ms, tg / steps * 1e3)
            ds.pin(ctx)
            for t in ds.traces(ctx):
                t.free()
        e2p_ms, tg = _events_to_proof(ctx, hp, pk, ds, ch0, out, steps)
    # two lanes (a context + host thread each, the same pinned events): one lane's event upload and trace generation run under the
    # other's proof — how a host that feeds a GPU from the executor's record channel would drive it (prove.rs:484-497)
    import threading
    ctx2 = prover.Context(device)
    hp2 = prover.HipProver(ds.chips, fri, synth.NUM_PV_ELTS, ctx=ctx2, specialize=specialize)
    pk2 = hp2.setup(ds.preprocessed(ctx2), [0, 0], F.to_monty(mach.pc_base), zero_digest)
    out2 = np.zeros(1 << 22, dtype=np.uint32)
    _events_to_proof(ctx2, hp2, pk2, ds, ch0, out2, 1)
    n2 = max(2, steps)
    t0 = time.perf_counter()
    th = threading.Thread(target=_events_to_proof, args=(ctx2, hp2, pk2, ds, ch0, out2, n2))
    th.start()
    _events_to_proof(ctx, hp, pk, ds, ch0, out, n2)
    th.join()
    two_lane_ms = (time.perf_counter() - t0) / (2 * n2) * 1e3
    alg = synth.shard_algorithmic_bytes(ds)
    rec = mach.shards[0].record
    event_bytes = int(sum(a.nbytes for a in [rec.cpu, rec.divrem, rec.branch, rec.memory_local] + list(rec.alu.values())))
    res = _fib_result(log_cycles, rec, ds, n_words, prove_ms, steps, e2p_ms, tg, pageable, event_bytes, alg, phases, kernels, gen_s)
    res["events_to_proof"]["two_lanes"] = {"ms_per_shard": round(two_lane_ms, 3), "value": round(1e3 / two_lane_ms, 4), "unit": "shard-proofs/s",
                                           "note": "two contexts + host threads on the one GPU, each running events -> traces -> proof back to back"}
    return res


def _events_to_proof(ctx, hp, pk, ds, ch0, out, steps):
    t0 = time.perf_counter()
    tg = 0.0
    for _ in range(steps):
        t1 = time.perf_counter()
        born = ds.traces(ctx)
        ctx.synchronize()
        tg += time.perf_counter() - t1
        hp.prove_shard(pk, ds.public_values, born, ch0.copy(), out=out)
        for t in born:
            t.free()
    ctx.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, tg


def _fib_result(log_cycles, rec, ds, n_words, prove_ms, steps, e2p_ms, tg, pageable, event_bytes, alg, phases, kernels, gen_s):
    return {"workload": f"FIB-{log_cycles}: a middle shard of examples/fibonacci, {len(rec.cpu)} cycles, full shard proof (commit+open), blowup 2, 84 queries, 16 PoW bits",
            "chips": {c.name: c.log_height for c in ds.chips}, "committed_cells": ds.committed_cells(), "proof_words": n_words,
            "prove": {"ms_per_proof": round(prove_ms, 3), "value": round(1e3 / prove_ms, 4), "unit": "shard-proofs/s", "steps": steps,
                      "note": "traces resident in HBM (device-born), as `value` of the main line"},
            "events_to_proof": {"ms_per_shard": round(e2p_ms, 3), "value": round(1e3 / e2p_ms, 4), "unit": "shard-proofs/s",
                                "tracegen_ms": round(tg / steps * 1e3, 3), "event_bytes": event_bytes,
                                "from_pageable_memory": {"ms_per_shard": round(pageable[0], 3), "tracegen_ms": round(pageable[1], 3)},
                                "note": "device trace generation of every chip from the shard's events in page-locked host memory (the events' upload is part of "
                                        "it), then the proof; one shard at a time, nothing overlapped"},
            "whole_shard": {"algorithmic_bytes": alg, "achieved": round(alg / (prove_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                            "frac": round(alg / (prove_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "phases_ms": phases, "kernels_ms": kernels, "event_generation_s": round(gen_s, 2)}


def tracegen_bench(args):
    """`python bench.py --tracegen`: device trace generation, chip after chip, 2^log_rows events each, events
    resident in pinned host memory. A "step" is one generate_trace call per chip. value = rows per second of kernel time; the
    wall-clock rate including the H2D copy of the events is reported beside it. HBM roofline per chip: algorithmic bytes =
    event bytes + 4 h w (the column-major trace) over the kernel time. The six AluEvent chips, Mul, DivRem, Branch, Jump and
    MovCond use the synthetic per-chip streams of ziren_amd/events.py; Cpu, MemoryInstrs, MiscInstrs and SyscallInstrs a record of
    ziren_amd/miniexec.py repeated to size (the row builders do not look across rows)."""
    from ziren_amd import events as E, miniexec as M
    ctx = prover.Context(0)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    n = 1 << args.log_rows
    prog, rec, _ = M.run(1 << 13, seed=1)

    def tile(ev):
        return np.tile(ev, -(-n // len(ev)))[:n]

    chips_todo = [(E.CHIP_NAMES[c], E.synthetic_alu_events(c, n), (lambda ev, c=c: ctx.tracegen_alu(c, ev, args.log_rows))) for c in sorted(E.CHIP_NAMES)]
    chips_todo += [
        ("Mul", E.synthetic_mul_events(n), lambda ev: ctx.tracegen_mul(ev, args.log_rows)),
        ("DivRem", E.synthetic_divrem_events(n), lambda ev: ctx.tracegen_divrem(ev, args.log_rows)),
        ("Branch", E.synthetic_branch_events(n), lambda ev: ctx.tracegen_branch(ev, args.log_rows)),
        ("Jump", E.synthetic_jump_events(n), lambda ev: ctx.tracegen_jump(ev, args.log_rows)),
        ("MovCond", E.synthetic_mov_cond_events(n), lambda ev: ctx.tracegen_mov_cond(ev, args.log_rows)),
        ("Cpu", tile(rec.cpu), lambda ev: ctx.tracegen_cpu(ev, prog, 0x1000, 1, args.log_rows)),
        ("MemoryInstrs", tile(rec.mem_instr), lambda ev: ctx.tracegen_memory_instrs(ev, args.log_rows)),
        ("MiscInstrs", tile(rec.misc), lambda ev: ctx.tracegen_misc_instrs(ev, args.log_rows)),
    ]
    per_chip, tot_rows, tot_kernel_ms, tot_wall_ms, tot_bytes = {}, 0, 0.0, 0.0, 0.0
    for name, ev, gen in chips_todo:
        words = ev.dtype.itemsize // 4
        pinned = ctx.host_alloc((n * words,))
        pinned[...] = ev.view(np.uint32).reshape(-1)
        evp = pinned.view(ev.dtype)
        kms, wms, width = [], [], 0
        for it in range(args.warmup + args.steps):
            t0 = time.perf_counter()
            m = gen(evp)
            w = (time.perf_counter() - t0) * 1e3
            k = sum(ms for nm, ms, _, _ in ctx.kernel_timings() if nm.startswith("tracegen"))
            width = m.width
            m.free()
            if it >= args.warmup:
                kms.append(k)
                wms.append(w)
        ctx.host_free(pinned)
        nbytes = float(ev.dtype.itemsize) * n + 4.0 * n * width
        k, w = float(np.mean(kms)), float(np.mean(wms))
        per_chip[name] = {"kernel_ms": round(k, 4), "call_ms": round(w, 3), "GBps": round(nbytes / k / 1e6, 1), "width": width,
                          "event_bytes": ev.dtype.itemsize, "Grows_per_s": round(n / k / 1e6, 2)}
        tot_rows += n
        tot_kernel_ms += k
        tot_wall_ms += w
        tot_bytes += nbytes
    cpu = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        O.lib().orc_set_num_threads(min(16, os.cpu_count() or 1))
        ns = 1 << 21
        rows, t = 0, 0.0
        for chip in sorted(E.CHIP_NAMES):
            ev = E.synthetic_alu_events(chip, ns)
            t0 = time.perf_counter()
            O.tracegen_alu(chip, ev, 21)
            t += time.perf_counter() - t0
            rows += ns
        cpu = {"value": round(rows / t, 1), "unit": "trace rows/s", "cores": O.lib().orc_num_threads(), "kind": "port",
               "sample": "the six AluEvent chips at 2^21 events each through oracle/tracegen.hpp (row-major Montgomery output, as the reference's generate_trace returns)"}
    achieved = tot_bytes / tot_kernel_ms / 1e6
    print(json.dumps({
        "metric": "trace rows/sec", "value": round(tot_rows / (tot_kernel_ms * 1e-3), 1), "unit": "trace rows/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(tot_kernel_ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"generate_trace of {len(chips_todo)} core chips ({', '.join(nm for nm, _, _ in chips_todo)}), 2^{args.log_rows} events each, "
                               "events resident in HBM", "log_rows": args.log_rows},
        "pcie_inclusive": {"value": round(tot_rows / (tot_wall_ms * 1e-3), 1), "unit": "trace rows/s", "ms_per_step": round(tot_wall_ms, 3),
                           "note": "wall clock of the zkm_tracegen_* calls: pinned-host events -> HBM, kernel, synchronise"},
        "chips": per_chip,
        "roofline": {"bound": "hbm", "kernel": "tracegen::alu_rows / cpu_rows", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None},
        "cpu_baseline": cpu}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracegen", action="store_true", help="benchmark device trace generation of the ALU chips instead of the shard proof")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-rows", type=int, default=22)
    ap.add_argument("--cpu-sample-log-rows", type=int, default=20)
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive measurement that follows the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU baseline at the full size (SYN-22: ~80 s on 16 cores) instead of on the SYN-20 sample")
    ap.add_argument("--no-inflight2", action="store_true", help="skip the two-shards-in-flight measurement that follows the timed region")
    ap.add_argument("--inflight", type=int, default=1, help="shards proven concurrently per GPU (one context + host thread each)")
    ap.add_argument("--kernel-timing", type=int, default=3,
                    help="0 off, 1 every launch, 2 launches >= 256 KiB, 3 (default): inside the timed region only the dominant kernel (the one the roofline "
                         "is quoted on) is timed, and the per-kernel table comes from an extra pass with mode 2 after it — timing every launch costs ~2.5 %% of a step")
    ap.add_argument("--from-host", action="store_true", help="time upload (pinned host traces -> HBM) + proof: the PCIe-inclusive rate")
    ap.add_argument("--interpreter", action="store_true", help="use the bytecode interpreter instead of per-chip quotient kernels")
    ap.add_argument("--queue", type=int, default=0, metavar="SHARDS", help="deal SHARDS distinct shards (0: off; the default with --queue -1 is 4 per GPU) "
                    "through the farm's claim queue (Farm.run_queue: whichever rank is free takes the next shard, crates/core/machine/src/utils/prove.rs:484) "
                    "and gather the proof streams to rank 0, instead of K identical steps per rank; value = shards / max-over-ranks time")
    ap.add_argument("--fib", type=str, default="21,22", help="log2 cycles of the fibonacci-guest shards proven after the timed region (the `fib` object of "
                    "the line: BASELINE.json's own workload beside SYN); empty string to skip")
    ap.add_argument("--workload", choices=["syn", "fib"], default="syn", help="fib: only the fibonacci-guest leg at --log-rows, printed as the line's `fib` object "
                    "with value / ms_per_step taken from it")
    args = ap.parse_args()
    if args.tracegen:
        return tracegen_bench(args)

    from ziren_amd import farm as farm_mod
    hp_holder = {}
    farm = farm_mod.Farm(device_sync=lambda: hp_holder["hp"].ctx.synchronize() if "hp" in hp_holder else None)
    rank, local_rank, world = farm.rank, farm.local_rank, farm.world

    fri = abi.FriConfig(1, 84, 16)  # core config, crates/stark/src/kb31_poseidon2.rs:203-213
    k = args.log_rows
    if args.workload == "fib":
        # the fibonacci guest's shard as the whole job: every rank proves its own copy (weak scaling), rank 0 prints
        farm.barrier()
        leg = fib_leg(local_rank, fri, k, max(1, args.steps), not args.interpreter)
        farm.barrier()
        slowest = farm.max_over_ranks(leg["prove"]["ms_per_proof"])
        if rank == 0:
            print(json.dumps({"metric": "shard-proofs/sec", "value": round(world * 1e3 / slowest, 4), "unit": "shard-proofs/s", "n_gpus": world,
                              "steps": args.steps, "warmup": 1, "ms_per_step": round(slowest, 3), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "u32", "data": "synthetic (the guest's events in closed form, ziren_amd/fibfast.py)",
                              "config": {"workload": leg["workload"], "log_rows": k, "parallelism": f"{world} GPU(s) x 1 shard in flight, independent shards, no collective"},
                              "fib": leg}))
        farm.close()
        return
    shard = synth.syn_shard(k, seed=0x5A4B4D00 + 1000 * rank)
    import threading
    M = max(1, args.inflight)
    lanes = []
    for j in range(M):
        hpj = prover.HipProver(shard.chips, fri, synth.NUM_PV_ELTS, device=local_rank, specialize=not args.interpreter)
        pkj = hpj.setup([], [], shard.pc_start, shard.initial_global_cumulative_sum)
        chj = prover.new_challenger()
        pkj.observe_into(chj)
        if args.from_host:  # traces live in page-locked host memory; every step uploads them again
            trj = []
            for c in shard.chips:
                h = hpj.ctx.host_alloc(c.trace.shape)
                h[...] = c.trace
                trj.append(h)
        else:
            trj = hpj.upload_traces([c.trace for c in shard.chips])  # inputs resident in HBM before timing
        lib.load().zkm_ctx_set_kernel_timing(hpj.ctx.h, C.c_int(2 if args.kernel_timing == 3 else args.kernel_timing))
        lanes.append((hpj, pkj, chj, trj, np.zeros(1 << 22, dtype=np.uint32)))
    hp = lanes[0][0]
    hp_holder["hp"] = hp
    for c in shard.chips:
        c.trace = None

    def prove_on(j):
        hpj, pkj, chj, trj, outj = lanes[j]
        ch = chj.copy()  # challenger cloned per shard (prove.rs:496)
        if args.from_host:
            dev = [hpj.ctx.upload_async(h) for h in trj]   # queued tallest first; the proof waits per matrix on the device
            proof = hpj.prove_shard(pkj, shard.public_values, dev, ch, out=outj)
            for d in dev:
                d.free()
            return proof
        return hpj.prove_shard(pkj, shard.public_values, trj, ch, out=outj)

    def step():
        if M == 1:
            return prove_on(0)
        ts = [threading.Thread(target=prove_on, args=(j,)) for j in range(1, M)]
        for t in ts:
            t.start()
        prove_on(0)
        for t in ts:
            t.join()

    phase_acc = {}
    kern_acc = {}

    def timed_step():
        step()
        for name, ms in hp.ctx.last_timings():
            phase_acc[name] = phase_acc.get(name, 0.0) + ms
        for name, ms, calls, nbytes in hp.ctx.kernel_timings():
            a = kern_acc.setdefault(name, [0.0, 0, 0.0])
            a[0] += ms
            a[1] += calls
            a[2] += nbytes

    if args.queue:
        # the farm as the reference runs it: distinct shards from one queue. The traces are this rank's resident set; what makes shard i its
        # own shard is its transcript (the index is observed after the key), so every proof differs and every proof verifies
        n_shards = args.queue if args.queue > 0 else 4 * world
        hpj, pkj, chj, trj, outj = lanes[0]

        def prove_shard_i(i):
            ch = chj.copy()
            idx = np.array([i + 1], dtype=np.uint32)
            lib.load().zkm_challenger_observe(C.byref(ch), abi.as_u32p(idx), C.c_size_t(1))
            return hpj.prove_shard(pkj, shard.public_values, trj, ch, out=outj)

        for _ in range(max(1, args.warmup)):
            prove_shard_i(0)
        farm.barrier()
        t0 = time.perf_counter()
        ids, proofs = farm.run_queue(n_shards, prove_shard_i)
        gathered = farm.gather_proofs(ids, proofs, n_shards)
        farm.barrier()
        elapsed = farm.max_over_ranks(time.perf_counter() - t0)
        mine = float(np.mean(farm.host_ms)) if farm.host_ms else 0.0
        slowest = farm.max_over_ranks(mine)
        proved = farm.sum_over_ranks(float(len(ids)))
        if rank == 0:
            assert len(gathered) == n_shards and len({p.tobytes() for p in gathered}) == n_shards, "the gathered proofs are not distinct"
            print(json.dumps({"metric": "shard-proofs/sec", "value": round(n_shards / elapsed, 4), "unit": "shard-proofs/s", "n_gpus": world,
                              "steps": n_shards, "warmup": max(1, args.warmup), "ms_per_step": round(elapsed / n_shards * 1e3, 3),
                              "higher_is_better": True, "scaling": "strong" if args.queue > 0 else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                              "config": {"workload": f"SYN-{k}: {n_shards} distinct full shard proofs dealt from one queue (Farm.run_queue), proof streams gathered to rank 0 inside the timed region",
                                         "log_rows": k, "parallelism": f"{world} GPU(s), claim queue, RCCL gather of {sum(len(p) for p in gathered) * 4} proof bytes"},
                              "host_ms_per_shard": {"rank0_mean": round(mine, 3), "max_over_ranks_of_means": round(slowest, 3)},
                              "shards_proved": int(proved), "shards_proved_by_rank0": len(ids)}))
        farm.close()
        return
    for _ in range(args.warmup):
        step()
    # which kernel the roofline will be quoted on: the one with the most HIP-event time in the last warm-up step (every launch >= 256 KiB timed)
    dominant, inst_acc, inst_steps, inst_ms = None, None, 0, None
    if args.kernel_timing == 3 and M == 1 and args.warmup > 0:
        last = hp.ctx.kernel_timings()
        if last:
            dominant = max(last, key=lambda t: t[1])[0]
            lib.load().zkm_ctx_set_kernel_timing_only(hp.ctx.h, dominant.encode())
    if M == 1:
        elapsed = farm.timed(timed_step, steps=args.steps, warmup=0)
        if dominant is not None:
            # the per-kernel table: an extra pass with every launch >= 256 KiB timed, outside the timed region
            lib.load().zkm_ctx_set_kernel_timing(hp.ctx.h, C.c_int(2))
            inst_acc, inst_steps = {}, min(args.steps, 5)
            t0 = time.perf_counter()
            for _ in range(inst_steps):
                step()
                for name, ms, calls, nbytes in hp.ctx.kernel_timings():
                    a = inst_acc.setdefault(name, [0.0, 0, 0.0])
                    a[0] += ms
                    a[1] += calls
                    a[2] += nbytes
            inst_ms = (time.perf_counter() - t0) / inst_steps * 1e3
    else:
        # M independent lanes, each proving K shards back to back (free-running: one lane's upload and
        # transcript round trips overlap another lane's kernels); the timed region ends when all are done.
        def run_all():
            def loop(j):
                time.sleep(0.04 * j)
                for _ in range(args.steps):
                    prove_on(j)
            ts = [threading.Thread(target=loop, args=(j,)) for j in range(1, M)]
            for t in ts:
                t.start()
            for _ in range(args.steps):
                prove_on(0)
                for name, ms in hp.ctx.last_timings():
                    phase_acc[name] = phase_acc.get(name, 0.0) + ms
                for name, ms, calls, nbytes in hp.ctx.kernel_timings():
                    a = kern_acc.setdefault(name, [0.0, 0, 0.0])
                    a[0] += ms
                    a[1] += calls
                    a[2] += nbytes
            for t in ts:
                t.join()
        elapsed = farm.timed(run_all, steps=1, warmup=0)

    if rank == 0:
        steps = args.steps
        ms_per_step = elapsed / steps * 1e3
        value = world * M * steps / elapsed
        alg_bytes = synth.shard_algorithmic_bytes(shard)
        # dominant kernel by accumulated HIP-event time on the prover's stream
        roofline = None
        if kern_acc:
            dom = max(kern_acc.items(), key=lambda kv: kv[1][0])
            name, (ms, calls, nbytes) = dom
            per_launch_ms = ms / max(calls, 1)
            kbytes = nbytes / max(calls, 1)
            achieved = kbytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
            # HBM bytes per launch: PMC counters cannot be read from inside this process, so the figure comes from the rocprofv3 --pmc
            # passes of this same command kept under profiles/ (tools/profile_r03.sh) — and only if they were taken on these very
            # kernel sources (the profile records their digest); otherwise null, never a stale number
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "r03_syn22_hbm_traffic.json")
            short = {"compress_layer": "merkle::compress_layer", "hash_leaves": "merkle::hash_leaves", "hash_leaves_tree": "merkle::hash_leaves_tree",
                     "lde_rows": "lde::lde_rows_big"}.get(name)
            if k == 22 and short and os.path.exists(tpath):
                tj = json.load(open(tpath))
                tk = tj["kernels"].get(short)
                if tk and tj.get("csrc_digest") == csrc_digest():
                    traffic = int(tk["hbm_bytes_per_launch"])
                    traffic_source = {"file": "profiles/r03_syn22_hbm_traffic.json", "commit": tj.get("commit"), "csrc_digest": tj.get("csrc_digest")}
                else:
                    traffic_source = {"file": "profiles/r03_syn22_hbm_traffic.json", "stale": True,
                                      "note": "taken on other kernel sources (csrc digest differs): not quoted"}
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                        "launches_per_step": calls // steps, "avg_launch_ms": round(per_launch_ms, 4),
                        "algorithmic_bytes_per_launch": int(kbytes),
                        "kernel_share_of_step": round(ms / steps / ms_per_step, 3),
                        "whole_shard": {"algorithmic_bytes": alg_bytes,
                                        "achieved": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                        "frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}
        # SURVEY 8d asks for the VALU side next to the HBM fraction: the hashing kernels are bound by instruction issue, not by bytes.
        # Poseidon2 runs on the FP64 vector pipe (csrc/poseidon2_f64.cuh); its ceiling is the chip's FP64 vector issue rate divided by
        # the permutation's dynamic instruction count as the hardware counted it (SQ_INSTS_VALU, profiles/r03_poseidon2_isa.json)
        valu = None
        # the per-kernel table: from the timed region itself, or — when only the dominant kernel was timed there — from the pass after it
        table, table_steps, table_step_ms = (inst_acc, inst_steps, inst_ms) if inst_acc else (kern_acc, steps, ms_per_step)
        hashing = [n for n in ("compress_layer", "hash_leaves", "hash_leaves_tree", "hash_fri_leaves", "hash_fri_leaves_tree", "compress_small", "compress_tail")
                   if n in table]
        if hashing and M == 1:
            perms = synth.shard_poseidon2_permutations(shard, fri.log_blowup)
            hms = sum(table[n][0] for n in hashing) / table_steps
            isa = poseidon2_isa()
            per_perm = isa["fp64"]["valu_instr_per_permutation"] if isa else None
            # one wave64 FP64 instruction = 64 lanes; the vector peak counts an FMA as 2 flop: instructions/s = TFLOPS / 2 / 64 per ... lane-instr
            peak = FP64_VECTOR_TFLOPS * 1e12 / 2 / per_perm / 1e9 if per_perm else None
            valu = {"bound": "fp64-vector-issue", "kernels": hashing, "poseidon2_permutations": perms, "ms": round(hms, 3),
                    "achieved": round(perms / hms / 1e6, 3), "peak": round(peak, 2) if peak else None, "unit": "Gperm/s",
                    "frac": round(perms / hms / 1e6 / peak, 3) if peak else None, "valu_instr_per_permutation": per_perm,
                    "valu_instr_source": "profiles/r03_poseidon2_isa.json (SQ_INSTS_VALU / permutations, tools/ubench_p2)" if isa else None,
                    "share_of_step": round(hms / table_step_ms, 3)}
        # the coset LDE as a group: its three kernels' time, the algorithmic bytes 12 n w (read n w words, write 2 n w)
        # of every committed column, and — from the PMC profile of these very sources — the bytes they really move through HBM
        lde = None
        lde_names = [n for n in ("lde_rows", "lde_cols_forward", "lde_cols_inverse") if n in table]
        if lde_names and M == 1:
            lde_ms = sum(table[n][0] for n in lde_names) / table_steps
            cells = sum((1 << c.log_height) * (c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in shard.chips)
            alg = 12.0 * cells
            tr = None
            tpath = os.path.join(ROOT, "profiles", "r03_syn22_hbm_traffic.json")
            if k == 22 and os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("csrc_digest") == csrc_digest():
                    tr = sum(tj["kernels"][n]["hbm_bytes_total"] for n in ("lde::lde_rows_big", "lde::lde_cols<true>", "lde::lde_cols<false>") if n in tj["kernels"])
            lde = {"ms": round(lde_ms, 3), "alg_GB": round(alg / 1e9, 3), "alg_GBps": round(alg / lde_ms / 1e6, 1), "frac_of_hbm_peak": round(alg / lde_ms / 1e6 / HBM_PEAK_GBPS, 4),
                   "traffic_GB": round(tr / 1e9, 3) if tr else None, "ratio": round(tr / alg, 2) if tr else None,
                   "structural_floor": "36 n w: a two-level split reads and writes the column three times (strided inverse, rows, strided forward)"}
        # hardware-measured vector-pipe occupancy of the hashing kernels (rocprofv3 --pmc SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE over this
        # command: tools/profile_r03.sh, definitions in tools/pmc_sq_summary.py), next to the derived valu.frac
        if valu is not None:
            spath = os.path.join(ROOT, "profiles", "r03_syn22_sq_counters.csv")
            if os.path.exists(spath):
                import csv
                rows = {r["Name"]: r for r in csv.DictReader(open(spath))}
                valu["valu_pipe_busy_pct_measured"] = {n: float(rows[n]["ValuPipeBusyPct(of SIMD time)"]) for n in ("merkle::compress_layer", "merkle::hash_leaves")
                                                       if n in rows and rows[n].get("ValuPipeBusyPct(of SIMD time)")}
                valu["valu_pipe_busy_source"] = "profiles/r03_syn22_sq_counters.csv"
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # rank 0, N = 1 only
            ks = k if args.cpu_full else min(args.cpu_sample_log_rows, k)
            wall, lde_s, threads, shown, why = cpu_baseline(ks, fri)
            # everything but the LDEs is linear in the rows; the LDEs (n log n) grow by (k + 1) / (ks + 1) on top (rows of the extended domain)
            lin = 1 << (k - ks)
            est = (wall - lde_s) * lin + lde_s * lin * (k + 1) / (ks + 1)
            cpu = {"value": round(1.0 / est, 6), "unit": "shard-proofs/s", "cores": threads, "cores_available": threads, "host_logical_cpus": shown,
                   "cores_note": why, "kind": "port",
                   "sample": (f"oracle (CPU restatement, canonical `% p` arithmetic, OpenMP, {threads} threads) proving one SYN-{ks} shard in {wall:.2f} s "
                              f"({lde_s:.2f} s of it coset LDEs)" + ("" if ks == k else f"; scaled to SYN-{k}: x{lin} for the linear phases, x{lin}*{k + 1}/{ks + 1} for the LDEs")),
                   "measured_at_full_size": ks == k, "sample_seconds": round(wall, 3), "estimated_seconds_full_size": round(est, 2),
                   "full_size_measurement": "profiles/r03_cpu_baseline_full.json (bench.py --cpu-full, the same proof at SYN-22 timed directly)"}
        two = None
        if world == 1 and M == 1 and not args.from_host and not args.no_inflight2:
            # the same GPU with two shards in flight (a second context + host thread, its own copy of the traces): the launch gaps and
            # host round trips of one proof are filled by the other's kernels. Throughput beside `value`; latency per proof doubles.
            hp0, pk0, ch0, tr0, out0 = lanes[0]
            hp1 = prover.HipProver(shard.chips, fri, synth.NUM_PV_ELTS, device=local_rank, specialize=not args.interpreter)
            pk1 = hp1.setup([], [], shard.pc_start, shard.initial_global_cumulative_sum)
            ch1 = prover.new_challenger()
            pk1.observe_into(ch1)
            tr1 = hp1.upload_traces([t.to_host() for t in tr0])
            lib.load().zkm_ctx_set_kernel_timing(hp1.ctx.h, C.c_int(args.kernel_timing))
            out1 = np.zeros(1 << 22, dtype=np.uint32)
            both = ((hp0, pk0, ch0, tr0, out0), (hp1, pk1, ch1, tr1, out1))

            def loop(lane, n):
                hpj, pkj, chj, trj, outj = lane
                for _ in range(n):
                    hpj.prove_shard(pkj, shard.public_values, trj, chj.copy(), out=outj)

            loop(both[1], 1)                       # warm the second context
            n2 = max(2, min(steps, 8))
            hp0.ctx.synchronize(); hp1.ctx.synchronize()
            t0 = time.perf_counter()
            th = threading.Thread(target=loop, args=(both[1], n2))
            th.start()
            loop(both[0], n2)
            th.join()
            hp0.ctx.synchronize(); hp1.ctx.synchronize()
            dt = time.perf_counter() - t0
            two = {"value": round(2 * n2 / dt, 4), "unit": "shard-proofs/s", "shards": 2 * n2, "ms_per_shard": round(dt / (2 * n2) * 1e3, 3),
                   "note": "two contexts on the one GPU, each proving back to back; the default line keeps one shard in flight (lowest latency)"}
            for t in tr1:
                t.free()
        pcie = None
        if world == 1 and M == 1 and not args.from_host and not args.no_pcie:
            # the boundary hands over host buffers (commit(record, traces)): the same step with the traces re-uploaded from page-locked
            # host memory every time (slabbed DMA + transpose on their own streams). Reported beside `value`, never as `value`.
            hpj, pkj, chj, trj, outj = lanes[0]
            host = []
            for t in trj:      # the resident traces go back to (page-locked) host memory, row-major as the reference holds them
                h = hpj.ctx.host_alloc((t.height, t.width))
                h[...] = t.to_host()
                host.append(h)
                t.free()

            def from_host_step():
                dev = [hpj.ctx.upload_async(h) for h in host]
                hpj.prove_shard(pkj, shard.public_values, dev, chj.copy(), out=outj)
                for d in dev:
                    d.free()
            from_host_step()
            hpj.ctx.synchronize()
            t0 = time.perf_counter()
            n_host = min(3, steps)
            for _ in range(n_host):
                from_host_step()
            hpj.ctx.synchronize()
            dt = (time.perf_counter() - t0) / n_host
            pcie = {"value": round(1.0 / dt, 4), "unit": "shard-proofs/s", "ms_per_step": round(dt * 1e3, 3), "steps": n_host,
                    "host_bytes_per_step": int(sum(h.nbytes for h in host)),
                    "note": "traces in page-locked host memory handed over every step (zkm_matrix_upload_async), upload overlapped with commit"}
            for h in host:
                hpj.ctx.host_free(h)
        fib = None
        if world == 1 and M == 1 and not args.from_host and args.fib:
            for t in lanes[0][3]:          # the SYN traces are not needed any more
                try:
                    t.free()
                except Exception:           # noqa: BLE001  (already handed back by the PCIe leg)
                    pass
            fib = {f"FIB-{kk}": fib_leg(local_rank, fri, int(kk), 3, not args.interpreter) for kk in args.fib.split(",")}
        line = {"metric": "shard-proofs/sec", "value": round(value, 4), "unit": "shard-proofs/s", "n_gpus": world,
                "steps": steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u32",
                "data": "synthetic" + (" (traces re-uploaded from pinned host memory every step)" if args.from_host else ""),
                "config": {"workload": f"SYN-{k}: full shard proof (commit+open), Cpu-like chip 2^{k} rows x 67 main "
                                       f"cols + 7 smaller chips, blowup 2, 84 queries, 16 PoW bits",
                           "log_rows": k, "parallelism": f"{world} GPU(s) x {M} shard(s) in flight, independent shards, no collective"},
                "phases_ms": {n: round(v / steps, 3) for n, v in phase_acc.items()},
                "kernels_ms": {n: {"ms": round(v[0] / table_steps, 3), "launches": v[1] // table_steps,
                                   "GBps": round(v[2] / max(v[0], 1e-9) / 1e6, 1)} for n, v in
                               sorted(table.items(), key=lambda kv: -kv[1][0])},
                "kernels_ms_source": (f"a pass of {inst_steps} steps after the timed region with every launch >= 256 KiB timed ({inst_ms:.3f} ms per step: a timed "
                                      f"launch costs a few microseconds of dispatch latency, ~250 launches per proof); inside the timed region only "
                                      f"{dominant} is timed, and the roofline is computed from those launches") if inst_acc else "the timed region",
                "roofline": roofline, "valu": valu, "lde": lde, "two_in_flight": two, "pcie_inclusive": pcie, "cpu_baseline": cpu, "fib": fib}
        print(json.dumps(line))
    farm.close()


if __name__ == "__main__":
    main()
