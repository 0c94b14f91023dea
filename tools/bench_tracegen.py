#!/usr/bin/env python3
"""Device trace generation benchmark: `python tools/bench_tracegen.py [--log-rows 21] [--steps 5] [--warmup 1]`
(also reachable as `python bench.py --tracegen`)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from ziren_amd import lib, prover

HBM_PEAK_GBPS = 8000.0


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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-rows", type=int, default=21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    tracegen_bench(ap.parse_args())
