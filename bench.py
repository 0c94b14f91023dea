#!/usr/bin/env python3
"""Benchmark of the shard-prover hot path (BASELINE.json metric: shard-proofs/sec, fibonacci 2^22-row trace).

A "step" is one full shard proof per GPU — MachineProver::commit + open (crates/stark/src/prover.rs:258-653), core FRI parameters
(blowup 2, 84 queries, 16 PoW bits). The workload is BASELINE.json's own: middle shards of the fibonacci guest as the reference cuts and
shapes them (the executor's shape check closes the shard, `CoreShapeConfig::fix_shape` pads it: ziren_amd/shape.py) at SHARD_SIZE = 2^21 —
1 569 808 cycles in a Cpu trace of 2^22 rows, 16 core chips + Byte + Program.

  python bench.py [--gpus N] [--steps K] [--warmup W]

ONE experiment for every N (the 1 -> N curve compares like with like): one process per GPU (launched by the driver through
torch.distributed.run, or by this script itself when WORLD_SIZE is not set: `python bench.py --gpus 8` starts 8 ranks); K N distinct
shards are dealt from one claim queue to whichever lane is free (crates/core/machine/src/utils/prove.rs:484-497), two lanes (a context +
host thread each) per GPU; per shard: the executor's events in page-locked host memory -> device traces (one zkm_tracegen_shard call) ->
proof, the next shard's events crossing PCIe under the current proof; the proof streams are gathered to rank 0 inside the timed region (a
no-op at N = 1); there is no data-path collective. Per-GPU work is fixed: scaling is weak, value = shards / max-over-ranks time. Every
lane starts the timed region with its first shard's events resident in HBM (claimed and prefetched before the barrier).

After the timed region (outside `value`) rank 0 runs the resident one-lane leg on the same shard — traces resident in HBM, one shard in
flight — for the per-kernel figures: `roofline` (dominant kernel, HIP events), `valu`, `lde`, `kernels_ms` (a pass with the side-stream
overlap off, so that the durations add up), `resident_one_lane`; and, at N = 1, the continuity legs and `cpu_baseline`.

Every number this file prints belongs to proofs the restated verifier (oracle verify_shard, outside the timed region) accepted: a leg
whose proof is rejected raises instead of printing. `--resident` prints the old resident-only line (profiling scripts use it).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # before anything initialises the HIP runtime (torch.distributed for N > 1 would): ziren_amd/lib.py

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VECTOR_TFLOPS = 78.6  # MI355X FP64 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz (MI355X_MICROARCH.md)
PROFILE_ROUND = "r06"
HASHING_KERNELS = ("compress_layer", "compress_layer_rowdig", "hash_leaves", "hash_rows", "hash_leaves_tree", "hash_fri_leaves", "hash_fri_leaves_lanes", "hash_fri_leaves_tree", "compress_small", "compress_tail")
LDE_KERNELS = ("lde_rows", "lde_cols_forward", "lde_cols_inverse")
ROCPROF_NAMES = {"compress_layer": "merkle::compress_layer", "hash_leaves": "merkle::hash_leaves", "hash_rows": "merkle::hash_rows", "hash_leaves_tree": "merkle::hash_leaves_tree",
                 "lde_rows": "lde::lde_rows_big", "lde_cols_forward": "lde::lde_cols<true>", "lde_cols_inverse": "lde::lde_cols<false>"}

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


def newest_profile(suffix):
    """profiles/<round>_<suffix> of the newest round that has one (this round's first), relative to the repo; None when no round has."""
    first = int(PROFILE_ROUND[1:])
    for n in range(first, 0, -1):
        rel = os.path.join("profiles", f"r{n:02d}_{suffix}")
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    return None


def poseidon2_isa():
    """Dynamic VALU instructions per permutation as the hardware counts them (SQ_INSTS_VALU over tools/ubench_p2's kernels, written by
    tools/profile_r06.sh -> tools/pmc_poseidon2.py); the newest profile kept; None when there is none."""
    rel = newest_profile("poseidon2_isa.json")
    if rel is not None:
        return json.load(open(os.path.join(ROOT, rel))), rel
    return None, None


def usable_cores():
    """(cores this process may actually use, logical CPUs the host shows, why they differ). A container sees every logical CPU of the host
    (os.cpu_count) but is scheduled under a cgroup CPU quota: on the GPU boxes of this pool cpu.max is 16 CPUs' worth of time on a
    256-thread host, and running more threads than the quota makes every OpenMP loop slower (profiles/r03_cpu_scaling.json), so the quota
    is the core count."""
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


def oracle():
    """The CPU restatement (tests/oracle_lib.py over oracle/liboracle.so): the checker of every leg's last proof and the `cpu_baseline`
    leg — never inside a timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    # OpenMP would start one thread per logical CPU of the host in every rank: the cores this process may use, shared among the node's ranks
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    O.lib().orc_set_num_threads(max(1, usable_cores()[0] // max(1, local_world)))
    return O


# ---- self-launch ---------------------------------------------------------------------------------------------------------------------------

def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set as torch.distributed.run would), rank 0 inheriting stdout — its one JSON line is the output — and fail
    if any rank fails."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ZKM_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = []
    deadline = time.time() + float(os.environ.get("ZKM_BENCH_LAUNCH_TIMEOUT", "3000"))
    for p in procs:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()                 # exactly the processes this function started
            rcs.append(-9)
    if any(rcs):
        sys.stderr.write(f"bench.py --gpus {n}: rank exit codes {rcs}\n")
        return 1
    return 0


# ---- workloads ---------------------------------------------------------------------------------------------------------------------------

class FibWorkload:
    """A middle shard of the fibonacci guest (examples/fibonacci; the loop's events in closed form, ziren_amd/fibfast.py — event for event
    what the executor restatement gives), the real chips with their recorded AIRs. kind "shaped": cut and padded as the reference does by
    default (fibfast.shaped_shard + shape.fix_shape); kind "tight": 2^k cycles, every chip padded to its next power of two."""

    def __init__(self, kind, log_size, shard_no=2):
        from ziren_amd import fibfast, shape as SH
        self.kind, self.log_size = kind, log_size
        t0 = time.perf_counter()
        if kind in ("shaped", "cut"):
            cycles, why = SH.executor_shard_cycles(1 << log_size, fibfast.loop_event_estimate)
            n = ((shard_no + 1) * cycles) // fibfast.LOOP + 8
            self.machine = fibfast.fib_shard(n, cycles, shard_no)
            # "cut": the same record without the shape step — chips with events only, every trace padded to its next power of two
            self.ds = fibfast.DeviceShard(self.machine, shape="fix" if kind == "shaped" else None)
            self.cut = why
            self.tag = f"fibs{log_size}" if kind == "shaped" else f"fibc{log_size}"
        else:
            assert shard_no == 2
            self.machine = fibfast.full_shard(log_size)
            self.ds = fibfast.DeviceShard(self.machine)
            self.cut = None
            self.tag = f"fib{log_size}"
        self.event_generation_s = time.perf_counter() - t0
        self.chips, self.public_values = self.ds.chips, self.ds.public_values
        self.cycles = len(self.machine.shards[0].record.cpu)
        self.data = "synthetic (the fibonacci guest's events in closed form, ziren_amd/fibfast.py; no executor or ELF in this environment)"

    @property
    def label(self):
        heights = {c.name: c.log_height for c in self.chips}
        if self.kind == "shaped":
            return (f"FIB-S{self.log_size}: a middle shard of examples/fibonacci as the reference cuts and shapes it at SHARD_SIZE = 2^{self.log_size} "
                    f"({self.cycles} cycles: the executor's shape check closes the shard; fix_shape pads it to the covering shape of least area — Cpu 2^{heights['Cpu']} rows, "
                    f"{len(self.chips)} chips), full shard proof (commit+open), blowup 2, 84 queries, 16 PoW bits")
        if self.kind == "cut":
            return (f"FIB-C{self.log_size}: the same record as FIB-S{self.log_size} ({self.cycles} cycles) without the shape step: the {len(self.chips)} chips that have events, "
                    f"tight heights (Cpu 2^{heights['Cpu']}), full shard proof (commit+open), blowup 2, 84 queries, 16 PoW bits")
        return (f"FIB-{self.log_size}: a middle shard of examples/fibonacci, {self.cycles} cycles, tight heights (no shape), {len(self.chips)} chips, "
                f"full shard proof (commit+open), blowup 2, 84 queries, 16 PoW bits")

    def setup(self, ctx, fri, specialize=True):
        from ziren_amd import chips as CH, field as F, prover, synth
        hp = prover.HipProver(self.chips, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=specialize)
        self.zero_digest = F.to_monty(np.array(CH.SEPTIC_START_X + CH.SEPTIC_START_Y, dtype=np.uint64)).astype(np.uint32)
        self.pc_start = F.to_monty(self.machine.pc_base)
        prep = self.ds.preprocessed(ctx)
        self.prep_host = [p.to_host() for p in prep]
        pk = hp.setup(prep, [0, 0], self.pc_start, self.zero_digest)
        ch0 = prover.new_challenger()
        pk.observe_into(ch0)
        return hp, pk, ch0

    def resident_traces(self, ctx):
        return self.ds.traces(ctx)

    def oracle_pk(self, O, fri):
        return O.Pk(self.prep_host, [0, 0], self.pc_start, self.zero_digest, fri.log_blowup)


class SynWorkload:
    """SYN-k: a Cpu-like chip at 2^k rows plus seven smaller chips, synthetic constraints; round 1-3's headline, kept for
    continuity."""

    def __init__(self, k, seed=0x5A4B4D00):
        from ziren_amd import synth
        self.kind, self.log_size, self.tag = "syn", k, f"syn{k}"
        self.shard = synth.syn_shard(k, seed=seed)
        self.chips, self.public_values = self.shard.chips, self.shard.public_values
        self.label = (f"SYN-{k}: full shard proof (commit+open), Cpu-like chip 2^{k} rows x 67 main cols + 7 smaller chips, blowup 2, 84 queries, 16 PoW bits")
        self.data = "synthetic"

    def setup(self, ctx, fri, specialize=True):
        from ziren_amd import prover, synth
        hp = prover.HipProver(self.chips, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=specialize)
        pk = hp.setup([], [], self.shard.pc_start, self.shard.initial_global_cumulative_sum)
        ch0 = prover.new_challenger()
        pk.observe_into(ch0)
        return hp, pk, ch0

    def resident_traces(self, ctx):
        out = [ctx.upload(c.trace) for c in self.chips]
        return out

    def host_traces(self):
        return [c.trace for c in self.chips]

    def oracle_pk(self, O, fri):
        return O.Pk([], [], self.shard.pc_start, self.shard.initial_global_cumulative_sum, fri.log_blowup)


def verify_or_die(wl, fri, ch0, proof, what):
    """The restated verifier (oracle verify_shard: constraints at zeta against the quotient, FRI queries, Merkle paths, proof of work) on a
    leg's last proof, outside every timed region. A rejected proof ends the run: no number is printed for it."""
    from ziren_amd import synth
    O = oracle()
    if getattr(wl, "_opk", None) is None:
        wl._opk = wl.oracle_pk(O, fri)          # the oracle's own commit of the preprocessed traces: once per workload
    rc = O.verify_shard(wl._opk, wl.chips, fri, synth.NUM_PV_ELTS, ch0.copy(), np.ascontiguousarray(proof, dtype=np.uint32).copy())
    if rc != 0:
        raise SystemExit(f"bench.py: the verifier REJECTED the {what} proof (code {rc}); nothing is reported for it")
    return True


# ---- legs --------------------------------------------------------------------------------------------------------------------------------

def accumulate(acc, ctx):
    for name, ms, calls, nbytes in ctx.kernel_timings():
        a = acc.setdefault(name, [0.0, 0, 0.0])
        a[0] += ms
        a[1] += calls
        a[2] += nbytes


def resident_leg(farm, wl, hp, pk, ch0, traces, steps, warmup, kernel_timing):
    """W warm-up proofs, then exactly K timed proofs between barriers, traces resident in HBM. Inside the timed region only the dominant
    kernel (the one the roofline is quoted on: the kernel with the most HIP-event time in a warm-up proof that runs every kernel alone) is timed; the per-kernel
    table comes from a pass after it (timing every launch >= 256 KiB costs ~2.5 % of a step)."""
    from ziren_amd import lib
    L = lib.load()
    ctx = hp.ctx
    out = np.zeros(1 << 22, dtype=np.uint32)
    L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2 if kernel_timing == 3 else kernel_timing))
    state = {}

    def step():
        state["proof"] = hp.prove_shard(pk, wl.public_values, traces, ch0.copy(), out=out)     # challenger cloned per shard (prove.rs:496)

    for _ in range(max(1, warmup)):
        step()
    dominant = None
    if kernel_timing == 3:
        # ranked on a proof with the side-stream LDE overlap off: with it on, a side-stream launch stretches under the leaf hashing it runs
        # beside (lde_cols_inverse: 12.7 ms of HIP-event time for 1.9 ms of work) and would be picked for a duration that is not its own
        if hasattr(L, "zkm_ctx_set_lde_overlap"):
            env = os.environ.get("ZKM_LDE_OVERLAP")
            was_on = not (env is not None and env.strip().lstrip("+-").isdigit() and int(env) == 0)     # the context's default (host_ctx.hpp)
            L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(0))
            step()
            L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(1 if was_on else 0))
        last = ctx.kernel_timings()
        if last:
            dominant = max(last, key=lambda t: t[1])[0]
            L.zkm_ctx_set_kernel_timing_only(ctx.h, dominant.encode())
    phase_acc, kern_acc = {}, {}

    def timed_step():
        step()
        for name, ms in ctx.last_timings():
            phase_acc[name] = phase_acc.get(name, 0.0) + ms
        accumulate(kern_acc, ctx)

    elapsed = farm.timed(timed_step, steps=steps, warmup=0)
    table, table_steps, table_ms, table_mode = kern_acc, steps, elapsed / steps * 1e3, "in the timed region (side-stream overlap on: concurrent kernels' durations overlap)"
    if dominant is not None:
        # the per-kernel table: every launch >= 256 KiB timed, the side-stream LDE overlap OFF for this pass — each kernel then runs alone
        # on the main stream and the HIP-event durations add up to (at most) the step
        L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2))
        can_serialise = hasattr(L, "zkm_ctx_set_lde_overlap")          # not in an older build under A/B comparison (ZKM_HIP_LIB)
        if can_serialise:
            L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(0))
        step()                                   # the first serialised proof re-sizes the pool's scratch
        table, table_steps = {}, min(steps, 5)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(table_steps):
            step()
            accumulate(table, ctx)
        table_ms = (time.perf_counter() - t0) / table_steps * 1e3
        table_mode = ("serialised: side-stream overlap off (zkm_ctx_set_lde_overlap 0)" if can_serialise else "side-stream overlap ON (library without the switch)") + ", every launch >= 256 KiB timed"
        if can_serialise:
            L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(1))
    return {"elapsed": elapsed, "phases": {n: v / steps for n, v in phase_acc.items()}, "kern_timed": kern_acc, "table": table,
            "table_steps": table_steps, "table_ms": table_ms, "table_mode": table_mode, "dominant": dominant, "proof": state["proof"].copy(), "out": out}


def traffic_profile(tag):
    """The rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this workload kept under profiles/ (tools/profile_r04.sh), only if it was
    taken on these very kernel sources (it records their digest); PMC counters cannot be read from inside this process."""
    rel = newest_profile(f"{tag}_hbm_traffic.json")
    if rel is None:
        return None, {"file": f"profiles/{PROFILE_ROUND}_{tag}_hbm_traffic.json", "missing": True}
    path = os.path.join(ROOT, rel)
    tj = json.load(open(path))
    if tj.get("csrc_digest") != csrc_digest():
        return None, {"file": rel, "stale": True, "note": "taken on other kernel sources (csrc digest differs): not quoted"}
    return tj, {"file": rel, "commit": tj.get("commit"), "csrc_digest": tj.get("csrc_digest")}


def roofline_objects(wl, fri, leg, steps):
    """`roofline` (dominant kernel: algorithmic bytes per launch from the launch site / its average HIP-event duration, measured inside the
    timed region), `valu` (the hashing kernels against the FP64 vector issue rate) and `lde` (the three LDE kernels as a group)."""
    from ziren_amd import synth
    ms_per_step = leg["elapsed"] / steps * 1e3
    alg_bytes = synth.shard_algorithmic_bytes(wl)
    tj, tsrc = traffic_profile(wl.tag)
    roofline = None
    if leg["kern_timed"]:
        name, (ms, calls, nbytes) = max(leg["kern_timed"].items(), key=lambda kv: kv[1][0])
        per_launch_ms = ms / max(calls, 1)
        kbytes = nbytes / max(calls, 1)
        achieved = kbytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        traffic = None
        if tj is not None:
            tk = tj["kernels"].get(ROCPROF_NAMES.get(name, name))
            traffic = int(tk["hbm_bytes_per_launch"]) if tk else None
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": tsrc,
                    "measured_in": "the resident one-lane leg (traces in HBM, one shard in flight), its timed steps, HIP events on the launch stream",
                    "launches_per_step": calls // steps, "avg_launch_ms": round(per_launch_ms, 4), "algorithmic_bytes_per_launch": int(kbytes),
                    "kernel_share_of_step": round(ms / steps / ms_per_step, 3),
                    "whole_shard": {"algorithmic_bytes": alg_bytes, "achieved": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                    "frac": round(alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}
    table, table_steps, table_ms = leg["table"], leg["table_steps"], leg["table_ms"]
    valu = None
    hashing = [n for n in HASHING_KERNELS if n in table]
    if hashing:
        perms = synth.shard_poseidon2_permutations(wl, fri.log_blowup)
        hms = sum(table[n][0] for n in hashing) / table_steps
        isa, isa_src = poseidon2_isa()
        per_perm = isa["fp64"]["valu_instr_per_permutation"] if isa else None
        # one wave64 FP64 instruction = 64 lanes; the vector peak counts an FMA as 2 flop: lane-instructions/s = TFLOPS / 2
        peak = FP64_VECTOR_TFLOPS * 1e12 / 2 / per_perm / 1e9 if per_perm else None
        valu = {"bound": "fp64-vector-issue", "kernels": hashing, "poseidon2_permutations": perms, "ms": round(hms, 3),
                "achieved": round(perms / hms / 1e6, 3), "peak": round(peak, 2) if peak else None, "unit": "Gperm/s",
                "frac": round(perms / hms / 1e6 / peak, 3) if peak else None, "valu_instr_per_permutation": per_perm,
                "valu_instr_source": (isa_src + " (SQ_INSTS_VALU / permutations, tools/ubench_p2)") if isa else None,
                "share_of_step": round(hms / table_ms, 3)}
        srel = newest_profile(f"{wl.tag}_sq_counters.csv")
        spath = os.path.join(ROOT, srel) if srel else None
        if spath:
            import csv
            rows = {r["Name"]: r for r in csv.DictReader(open(spath))}
            valu["valu_pipe_busy_pct_measured"] = {n: float(rows[n]["ValuPipeBusyPct(of SIMD time)"]) for n in ("merkle::compress_layer", "merkle::compress_layer_rowdig", "merkle::hash_leaves", "merkle::hash_rows")
                                                   if n in rows and rows[n].get("ValuPipeBusyPct(of SIMD time)")}
            valu["valu_pipe_busy_source"] = os.path.relpath(spath, ROOT)
    lde = None
    lde_names = [n for n in LDE_KERNELS if n in table]
    if lde_names:
        lde_ms = sum(table[n][0] for n in lde_names) / table_steps
        cells = sum((1 << c.log_height) * (c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in wl.chips)
        alg = 12.0 * cells
        tr = None
        if tj is not None:
            tr = sum(tj["kernels"][ROCPROF_NAMES[n]]["hbm_bytes_total"] for n in LDE_KERNELS if ROCPROF_NAMES[n] in tj["kernels"]) / max(tj.get("steps", 1), 1)
        lde = {"ms": round(lde_ms, 3), "alg_GB": round(alg / 1e9, 3), "alg_GBps": round(alg / lde_ms / 1e6, 1), "frac_of_hbm_peak": round(alg / lde_ms / 1e6 / HBM_PEAK_GBPS, 4),
               "traffic_GB": round(tr / 1e9, 3) if tr else None, "ratio": round(tr / alg, 2) if tr else None,
               "time_is": "serialised (side-stream overlap off): the three kernels' HIP-event durations, each running alone",
               "structural_floor": "36 n w: a two-level split reads and writes the column three times (strided inverse, rows, strided forward)"}
    kernels_ms = {n: {"ms": round(v[0] / table_steps, 3), "launches": v[1] // table_steps, "GBps": round(v[2] / max(v[0], 1e-9) / 1e6, 1)}
                  for n, v in sorted(table.items(), key=lambda kv: -kv[1][0])}
    ksum = sum(v[0] for v in table.values()) / table_steps
    if leg["dominant"] and "serialised" in leg["table_mode"] and ksum > table_ms * 1.001:
        raise SystemExit(f"bench.py: the serialised per-kernel durations add up to {ksum:.3f} ms, more than the {table_ms:.3f} ms step they were measured in")
    source = {"pass": f"{table_steps} proofs after the timed region, {leg['table_mode']}", "ms_per_step_of_that_pass": round(table_ms, 3),
              "kernels_ms_sum": round(ksum, 3), "untimed_remainder_ms": round(table_ms - ksum, 3),
              "note": "kernels_ms_sum <= the pass's step time is asserted; the remainder is launches below 256 KiB, copies, host round trips and idle gaps. "
                      f"Inside the resident leg's timed region only {leg['dominant']} is timed, and the roofline is computed from those launches (overlap on: in situ)"}
    return roofline, valu, lde, kernels_ms, source


def events_leg(wl, hp, pk, ch0, out, steps):
    """events -> device traces -> proof, the reference's prove-a-record step (crates/core/machine/src/utils/prove.rs:484-497), fib only.
    `serial`: one shard at a time, every generator uploading its own events (page-locked). `pipelined`: one context, one host thread —
    the next shard's events are queued on the DMA stream (zkm_events_upload_async) right before the current shard's proof, so they cross
    PCIe under its kernels; trace generation then finds them in HBM."""
    ctx, ds = hp.ctx, wl.ds
    ds.pin(ctx)
    event_bytes = ds.event_bytes()

    def one(pre=None):
        born = ds.traces(ctx, pre)
        proof = hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
        for t in born:
            t.free()
        return proof

    one()
    ctx.synchronize()
    t0 = time.perf_counter()
    tg = 0.0
    for _ in range(steps):
        t1 = time.perf_counter()
        born = ds.traces(ctx)
        ctx.synchronize()
        tg += time.perf_counter() - t1
        hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
        for t in born:
            t.free()
    ctx.synchronize()
    serial_ms = (time.perf_counter() - t0) / steps * 1e3
    # upload rate on its own: the whole shard's events, page-locked host -> HBM, nothing else running
    t1 = time.perf_counter()
    pre = ds.prefetch(ctx)
    for d in pre.values():
        d.free()                   # waits for the copy
    h2d_s = time.perf_counter() - t1
    pre = ds.prefetch(ctx)
    one(pre)                       # warm the pipelined path
    ctx.synchronize()
    pre = ds.prefetch(ctx)
    ctx.synchronize()              # steady state: when a proof starts, its own events landed under the previous one (a 6-shard loop would
                                   # otherwise charge the first, unhidden upload to every shard: +1.7 ms)
    parts = {"prefetch_calls": 0.0, "trace_generation": 0.0, "proof": 0.0}
    t0 = time.perf_counter()
    for i in range(steps):
        ta = time.perf_counter()
        nxt = ds.prefetch(ctx) if i + 1 < steps else None      # shard i + 1's events start crossing PCIe now
        tb = time.perf_counter()
        born = ds.traces(ctx, pre)
        tc = time.perf_counter()
        proof = hp.prove_shard(pk, wl.public_values, born, ch0.copy(), out=out)
        td = time.perf_counter()
        for t in born:
            t.free()
        pre = nxt
        parts["prefetch_calls"] += tb - ta
        parts["trace_generation"] += tc - tb
        parts["proof"] += td - tc
    ctx.synchronize()
    pipe_ms = (time.perf_counter() - t0) / steps * 1e3
    return {"event_bytes": event_bytes, "h2d_GBps_alone": round(event_bytes / h2d_s / 1e9, 2),
            "serial": {"ms_per_shard": round(serial_ms, 3), "value": round(1e3 / serial_ms, 4), "tracegen_ms": round(tg / steps * 1e3, 3),
                       "note": "one shard at a time: every generator uploads its events (page-locked) and runs, then the proof; nothing overlapped"},
            "pipelined": {"ms_per_shard": round(pipe_ms, 3), "value": round(1e3 / pipe_ms, 4), "steps": steps,
                          "host_ms": {k: round(v / steps * 1e3, 3) for k, v in parts.items()},
                          "note": "one context, one host thread: shard i+1's events are queued on the DMA stream right before shard i's proof"},
            "unit": "shard-proofs/s"}, proof.copy()


def two_in_flight_leg(wl, fri, device, lane0, specialize, steps):
    """The same GPU with two shards in flight (a second context + host thread, its own copy of the traces): the launch gaps and host round
    trips of one proof are filled by the other's kernels. Throughput beside `value`; latency per proof doubles."""
    import threading
    from ziren_amd import prover
    hp0, pk0, ch0, tr0, out0 = lane0
    ctx1 = prover.Context(device)
    hp1, pk1, ch1 = wl.setup(ctx1, fri, specialize)
    tr1 = wl.resident_traces(ctx1)
    out1 = np.zeros(1 << 22, dtype=np.uint32)
    both = ((hp0, pk0, ch0, tr0, out0), (hp1, pk1, ch1, tr1, out1))

    def loop(lane, n):
        hpj, pkj, chj, trj, outj = lane
        for _ in range(n):
            hpj.prove_shard(pkj, wl.public_values, trj, chj.copy(), out=outj)

    loop(both[1], 1)
    n2 = max(2, min(steps, 8))
    hp0.ctx.synchronize(); hp1.ctx.synchronize()
    t0 = time.perf_counter()
    th = threading.Thread(target=loop, args=(both[1], n2))
    th.start()
    loop(both[0], n2)
    th.join()
    hp0.ctx.synchronize(); hp1.ctx.synchronize()
    dt = time.perf_counter() - t0
    for t in tr1:
        t.free()
    return {"value": round(2 * n2 / dt, 4), "unit": "shard-proofs/s", "shards": 2 * n2, "ms_per_shard": round(dt / (2 * n2) * 1e3, 3),
            "note": "two contexts on the one GPU, each proving back to back; the default line keeps one shard in flight (lowest latency)"}


def pcie_leg(wl, hp, pk, ch0, out, steps):
    """SYN only (for fib the events leg is the boundary's PCIe-inclusive figure): the traces handed over as page-locked host buffers every
    step (zkm_matrix_upload_async: slabbed DMA + transpose on their own streams, overlapped with commit). Never `value`."""
    ctx = hp.ctx
    host = []
    for t in wl.host_traces():
        h = ctx.host_alloc(t.shape)
        h[...] = t
        host.append(h)

    def step():
        dev = [ctx.upload_async(h) for h in host]
        hp.prove_shard(pk, wl.public_values, dev, ch0.copy(), out=out)
        for d in dev:
            d.free()
    step()
    ctx.synchronize()
    n = min(3, steps)
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    res = {"value": round(1.0 / dt, 4), "unit": "shard-proofs/s", "ms_per_step": round(dt * 1e3, 3), "steps": n, "host_bytes_per_step": int(sum(h.nbytes for h in host)),
           "note": "traces in page-locked host memory handed over every step (zkm_matrix_upload_async), upload overlapped with commit"}
    for h in host:
        ctx.host_free(h)
    return res


def cpu_baseline_leg(wl, fri, full, sample_log):
    """The CPU restatement (oracle, kind "port": canonical `% p` arithmetic, OpenMP) proving a bounded sample of the same workload on the
    cores this process may use — rank 0, N = 1 only, after the timed region. fib: the same guest's shard cut and shaped at a smaller
    SHARD_SIZE (traces from the oracle's row builders, not timed); syn: SYN-k' . Scaled to the full size: the linear phases by the ratio
    of committed cells, the LDEs by an extra (k + 1) / (k' + 1)."""
    from ziren_amd import synth
    O = oracle()
    L = O.lib()
    L.orc_lde_seconds.restype = C.c_double
    threads, shown, why = usable_cores()
    L.orc_set_num_threads(threads)

    def cells(chips):
        return sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in chips)

    if wl.kind == "syn":
        ks = wl.log_size if full else min(sample_log, wl.log_size)
        sh = synth.syn_shard(ks)
        opk = O.Pk([], [], sh.pc_start, sh.initial_global_cumulative_sum, fri.log_blowup)
        chips, traces, pv = sh.chips, [c.trace for c in sh.chips], sh.public_values
        what = f"SYN-{ks}"
        k_full, k_s = wl.log_size, ks
    else:
        import machine_lib as ML
        from ziren_amd import fibfast
        sample = wl if full else FibWorkload(wl.kind, min(sample_log, wl.log_size))
        chips = ML.build_shard(ML.Oracle(O), sample.machine, 0, shape=sample.ds.shape)
        chips[-2].prep_trace = O.tracegen_byte_table()
        chips[-1].prep_trace = O.tracegen_program(0, sample.machine.shards[0].record.cpu, sample.machine.program, sample.machine.pc_base, sample.ds.plh)
        from ziren_amd import chips as CH, field as F
        zero = F.to_monty(np.array(CH.SEPTIC_START_X + CH.SEPTIC_START_Y, dtype=np.uint64)).astype(np.uint32)
        opk = O.Pk([chips[-2].prep_trace, chips[-1].prep_trace], [0, 0], F.to_monty(sample.machine.pc_base), zero, fri.log_blowup)
        traces, pv = [c.trace for c in chips], sample.ds.public_values
        what = f"FIB-S{sample.log_size}" if wl.kind == "shaped" else f"FIB-{sample.log_size}"
        what += f" ({sample.cycles} cycles, Cpu 2^{chips[0].log_height} rows)"
        k_full, k_s = wl.chips[0].log_height, chips[0].log_height
    ch = O.new_challenger()
    opk.observe_into(ch)
    L.orc_lde_seconds(C.c_int(1))
    t0 = time.time()
    O.prove_shard(opk, chips, traces, pv, fri, synth.NUM_PV_ELTS, ch)
    wall = time.time() - t0
    lde_s = float(L.orc_lde_seconds(C.c_int(0)))
    ratio = cells(wl.chips) / cells(chips)
    est = (wall - lde_s) * ratio + lde_s * ratio * (k_full + 1) / (k_s + 1)
    return {"value": round(1.0 / est, 6), "unit": "shard-proofs/s", "cores": threads, "host_logical_cpus": shown, "cores_note": why, "kind": "port",
            "sample": (f"oracle (CPU restatement, canonical `% p` arithmetic, OpenMP, {threads} threads) proving one {what} shard in {wall:.2f} s ({lde_s:.2f} s of it coset LDEs)"
                       + ("" if full else f"; scaled to the benchmarked shard by committed cells (x{ratio:.2f}), the LDEs by a further ({k_full} + 1) / ({k_s} + 1)")),
            "measured_at_full_size": bool(full), "sample_seconds": round(wall, 3), "estimated_seconds_full_size": round(est, 2),
            "full_size_measurement": ("this run: the benchmarked shard itself on the CPU; kept per round as " if full else "") + (newest_profile("cpu_baseline_full.json") or "none kept under profiles/") + " (bench.py's default since round 6; --cpu-sample scales a smaller shard instead)"}


# ---- the farm line (every N): the claim queue over events -> traces -> proof -----------------------------------------------------------------

class StubLane:
    """tests/test_bench_cli.py only (ZKM_BENCH_STUB_PROVER=1, CPU, gloo): stands in for the GPU lane so that the launch / queue / gather /
    line logic of `--gpus N` runs without a GPU. Its "proofs" are labelled words; the line says "stub": true and carries no rate claim."""

    def __init__(self, rank):
        self.rank = rank
        self.event_bytes = 1 << 20
        self.last = None

    def prefetch(self, i):
        return ("events", i)

    def prove(self, i, handle):
        assert handle == ("events", i)
        time.sleep(0.01)
        self.last = i
        return np.array([0x5AFE, i, self.rank] + [i] * (8 + i % 3), dtype=np.uint32)

    def verify_last(self):
        return True

    def verify(self, i, proof):
        return True

    def sync(self):
        pass


class FibQueueLane:
    """One rank's side of the farm: a pool of distinct middle shards of the guest (their events page-locked), one context, the key."""

    def __init__(self, device, rank, fri, log_size, specialize, pool=4, share=None):
        from ziren_amd import lib, prover
        self.fri = fri
        # a second lane of the same rank proves from the same page-locked pool
        self.wls = share.wls if share is not None else [FibWorkload("shaped", log_size, shard_no=2 + (pool * rank + j) % 6) for j in range(pool)]
        self.ctx = prover.Context(device)
        self.hp, self.pk, self.ch0 = self.wls[0].setup(self.ctx, fri, specialize)       # one program, one shape: one key for every shard
        lib.load().zkm_ctx_set_kernel_timing(self.ctx.h, C.c_int(0))
        if hasattr(lib.load(), "zkm_ctx_set_host_wait") and os.environ.get("ZKM_BENCH_LANE_WAIT", "blocking") == "blocking":
            # a farm lane sleeps while it waits for its GPU work: with two lanes per GPU the throughput is that of spinning lanes
            # (profiles/r05_ab_host_wait.txt) at 1.2 host cores per rank instead of 3 — 8 ranks fit a 16-CPU host
            lib.load().zkm_ctx_set_host_wait(self.ctx.h, C.c_int(1))
        if share is None:
            for w in self.wls:
                w.prep_host, w.pc_start, w.zero_digest = self.wls[0].prep_host, self.wls[0].pc_start, self.wls[0].zero_digest     # one program: one key
                assert [c.name for c in w.chips] == [c.name for c in self.wls[0].chips] and [c.log_height for c in w.chips] == [c.log_height for c in self.wls[0].chips]
                w.ds.pin(self.ctx)
        self.event_bytes = self.wls[0].ds.event_bytes()
        self.out = np.zeros(1 << 22, dtype=np.uint32)
        self.last = None

    def challenger(self, i):
        """Shard i's transcript: the key, then its index in the batch — what makes every queue entry its own proof even where two entries
        share a pool shard's events."""
        from ziren_amd import abi, lib
        ch = self.ch0.copy()
        idx = np.array([i + 1], dtype=np.uint32)
        lib.load().zkm_challenger_observe(C.byref(ch), abi.as_u32p(idx), C.c_size_t(1))
        return ch

    def prefetch(self, i):
        return self.wls[i % len(self.wls)].ds.prefetch(self.ctx)

    def prove(self, i, handle):
        w = self.wls[i % len(self.wls)]
        born = w.ds.traces(self.ctx, handle)          # one zkm_tracegen_shard call: every generator queued behind the events' copy
        proof = self.hp.prove_shard(self.pk, w.public_values, born, self.challenger(i), out=self.out)
        for t in born:
            t.free()
        self.last = (i, w, proof.copy())
        return proof

    def verify(self, i, proof):
        """Queue entry i's proof through the restated verifier (its pool shard's chips and public values, its own transcript)."""
        w = self.wls[i % len(self.wls)]
        for other in self.wls:
            if getattr(other, "_opk", None) is not None:
                w._opk = other._opk               # one program, one key: the oracle commits the preprocessed traces once per rank
        return verify_or_die(w, self.fri, self.challenger(i), proof, f"queue shard {i}")

    def verify_last(self):
        i, w, proof = self.last
        return self.verify(i, proof)

    def sync(self):
        self.ctx.synchronize()


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "shards", "warmup", "ms_per_step", "ms_per_shard", "higher_is_better", "scaling", "vs_baseline", "dtype",
             "verified", "verified_proofs", "stub", "data", "config", "roofline", "valu", "lde", "kernels_ms", "kernels_ms_source", "phases_ms", "resident_one_lane",
             "two_in_flight", "events_to_proof", "other_workloads", "micro", "cpu_baseline", "primed", "host_ms_per_shard", "host_cpu_s_per_shard", "event_bytes_per_shard", "h2d_GBps_per_rank",
             "shards_proved", "shards_proved_by_rank0", "fewest_shards_on_a_rank", "lib_digest", "wall_s")


def build_and_warm_lanes(args, fri, rank, local_rank):
    """A rank's lanes (contexts, key, page-locked event pool) and their warm-up proofs. bench.py runs this BEFORE the process group exists:
    HIP deals a process's streams onto a few hardware queues in the order they are made (GPU_MAX_HW_QUEUES, four by default), and which
    streams end up sharing one is worth up to 3 ms per shard (profiles/r05_ab_hw_queues.txt: 43.2 ms with the prover's streams made first,
    44.4-45.0 when torch.cuda and RCCL made theirs first, 46-52 ms for the unlucky counts) — so every rank of every N makes the prover's
    streams first, in the order the N = 1 line makes them, and torch / RCCL take what is left for streams that are idle while shards are
    proven."""
    t_start = time.perf_counter()
    stub = os.environ.get("ZKM_BENCH_STUB_PROVER") == "1"
    specialize = not args.interpreter
    M = max(1, args.inflight)
    lib_digest = None
    if not stub:
        from ziren_amd import lib
        lib_digest = lib.check_build_identity()          # a library built from other sources than the tree's is refused here
    lane = StubLane(rank) if stub else FibQueueLane(local_rank, rank, fri, args.shard_size_log, specialize)
    lanes = [lane] + [StubLane(rank) if stub else FibQueueLane(local_rank, rank, fri, args.shard_size_log, specialize, share=lane) for _ in range(M - 1)]
    warm = max(1, args.warmup)
    t_setup = time.perf_counter()
    if len(lanes) > 1 and not stub:
        import threading
        ths = [threading.Thread(target=lambda l=l: [l.prove(0, l.prefetch(0)) for _ in range(warm)]) for l in lanes]   # warm the lanes the way they run: together
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    else:
        for l in lanes:
            for _ in range(warm):
                l.prove(0, l.prefetch(0))
    return {"lanes": lanes, "lib_digest": lib_digest, "t_start": t_start, "t_setup": t_setup, "stub": stub, "warm": warm}


def farm_main(args, farm, fri, built):
    """The one experiment behind every `--gpus N` (module docstring): K N shards through the claim queue, M lanes per GPU."""
    rank, local_rank, world = farm.rank, farm.local_rank, farm.world
    lanes, lib_digest, t_start, t_setup, stub, warm = (built[k] for k in ("lanes", "lib_digest", "t_start", "t_setup", "stub", "warm"))
    M = len(lanes)
    lane = lanes[0]

    def sync_all():
        for l in lanes:
            l.sync()
    farm.device_sync = sync_all
    # default: exactly --steps K shards per GPU, K N in the queue (weak scaling: per-GPU work fixed); --queue S: S shards in all
    per_gpu = max(1, args.steps)
    n_shards = args.queue if args.queue > 0 else per_gpu * world
    # the gather's first use (RCCL loads its kernels and sets its channels up on the first call of a collective: ~20 ms, a millisecond per
    # shard of a 20-shard timed region) belongs to the warm-up as well: streams of the benchmarked length, dealt round-robin, thrown away
    if farm.dist is not None:
        last = lanes[0].last[2] if isinstance(getattr(lanes[0], "last", None), tuple) else np.zeros(1 << 15, dtype=np.uint32)
        mine_w = [i for i in range(n_shards) if i % world == rank]
        farm.gather_proofs(mine_w, [last] * len(mine_w), n_shards)
    t_warm = time.perf_counter()
    pairs = [(l.prove, l.prefetch) for l in lanes]
    farm.prime_queue(n_shards, pairs)         # every lane's first shard: claimed, its events queued for HBM — resident when the clock starts
    farm.barrier()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    ids, proofs = farm.run_queue(n_shards, lanes=pairs)
    t_g = time.perf_counter()
    gathered = farm.gather_proofs(ids, proofs, n_shards, copy=False)
    t_b = time.perf_counter()
    farm.barrier()
    gather_s = (t_b - t_g, time.perf_counter() - t_b)      # the gather (with this rank's wait for the slowest rank's last proof), the closing barrier
    elapsed = farm.max_over_ranks(time.perf_counter() - t0)
    cpu_s = time.process_time() - cpu0
    wall_local = time.perf_counter() - t0
    t_timed = time.perf_counter()
    mine = float(np.mean(farm.host_ms)) if farm.host_ms else 0.0
    slowest = farm.max_over_ranks(mine)
    cpu_per_shard = cpu_s / max(1, len(ids))
    cpu_per_shard_max = farm.max_over_ranks(cpu_per_shard)
    proved = farm.sum_over_ranks(float(len(ids)))
    fewest = -farm.max_over_ranks(-float(len(ids)))
    # verification, outside the timed region. N = 1: EVERY gathered proof goes through the restated verifier (a fraction of a second each on
    # the box's cores). N > 1: every lane's last proof on every rank — each context's transcript is checked, bounded at M proofs per rank
    # (seconds on a rank's share of the host cores) — and `verified_proofs` says exactly how many of how many that was.
    ok, checked = 1.0, 0
    if world == 1 and gathered is not None:
        for i, p in enumerate(gathered):
            ok = min(ok, 1.0 if lane.verify(i, p) else 0.0)
            checked += 1
    else:
        for l in lanes:
            if l.last is not None:
                ok = min(ok, 1.0 if l.verify_last() else 0.0)
                checked += 1
    all_ok = -farm.max_over_ranks(-ok)
    verified_proofs = int(farm.sum_over_ranks(float(checked)))
    t_verify = time.perf_counter()
    ranks_in_group = farm.dist.get_world_size() if farm.dist is not None else 1
    if all_ok != 1.0 or int(proved) != n_shards:
        raise SystemExit("bench.py: a rank's proof was rejected or a shard was proven by nobody")
    line = dict.fromkeys(LINE_KEYS)
    pool_n = len(getattr(lane, "wls", [None]))
    if rank == 0:
        assert len(gathered) == n_shards and len({p.tobytes() for p in gathered}) == n_shards, "the gathered proofs are not distinct"
        ms = elapsed / n_shards * 1e3
        backend = "none (one process)" if farm.dist is None else "gloo" if (stub or os.environ.get("ZKM_BENCH_ONE_DEVICE") == "1") else "nccl (RCCL)"
        line.update({
            "metric": "shard-proofs/sec", "value": round(n_shards / elapsed, 4), "unit": "shard-proofs/s", "n_gpus": world,
            "steps": n_shards if args.queue > 0 else per_gpu, "shards": n_shards,
            "warmup": warm, "ms_per_step": round(ms if args.queue > 0 else elapsed / per_gpu * 1e3, 3), "ms_per_shard": round(ms, 3),
            "higher_is_better": True, "scaling": "strong" if args.queue > 0 else "weak",
            "vs_baseline": None, "dtype": "u32", "verified": True, "verified_proofs": {"checked_by_the_verifier": verified_proofs, "of": n_shards,
                                                                                    "which": "every gathered proof" if world == 1 else "the last proof of every lane of every rank",
                                                                                    "all_gathered_streams_distinct": True},
            "stub": stub or None,
            "data": f"synthetic (the fibonacci guest's events in closed form, ziren_amd/fibfast.py; no executor or ELF in this environment; {pool_n} distinct middle shards of the guest per rank, their events page-locked)",
            "config": {"workload": (f"FIB-S{args.shard_size_log}: {n_shards} shard proofs ({'as given' if args.queue > 0 else str(per_gpu) + ' per GPU'}), each its own transcript, over {pool_n} distinct event sets per rank "
                                    f"(middle shards of examples/fibonacci, cut and shaped as the reference does at SHARD_SIZE = 2^{args.shard_size_log}), dealt from one claim queue; per shard: executor events in "
                                    f"page-locked host memory -> device traces -> full shard proof (commit+open, blowup 2, 84 queries, 16 PoW bits); proof streams gathered to rank 0 inside the timed region"),
                       "distinct_event_sets_per_rank": pool_n,
                       "parallelism": f"{world} GPU(s), one process each, {M} shard(s) in flight per GPU (a context + host thread each), claim queue (every lane claims one shard "
                                      f"ahead: its events cross PCIe under the current proof), {backend} gather of {sum(len(p) for p in gathered) * 4} proof bytes to rank 0, "
                                      f"no data-path collective",
                       "shards_in_flight_per_gpu": M, "ranks_in_process_group": ranks_in_group, "backend": backend},
            "host_ms_per_shard": {"rank0_mean": round(mine, 3), "max_over_ranks_of_means": round(slowest, 3)},
            "host_cpu_s_per_shard": {"rank0": round(cpu_per_shard, 4), "max_over_ranks": round(cpu_per_shard_max, 4),
                                     "cores_busy_rank0": round(cpu_s / max(wall_local, 1e-9), 2),
                                     "lane_threads_cores_rank0": [round(c / max(wall_local, 1e-9), 2) for c in getattr(farm, "lane_cpu_s", [])],
                                     "other_threads_cores_rank0": round((cpu_s - sum(getattr(farm, "lane_cpu_s", []))) / max(wall_local, 1e-9), 2),
                                     "note": "process CPU seconds (all threads: the lanes' Python + ctypes, root polling, the HIP runtime's own threads) per shard this rank proved, timed region only"},
            "primed": {"shards_claimed_and_uploaded_before_the_clock": M, "event_bytes_each": lane.event_bytes,
                       "note": "every lane claims its first shard and queues its events' copy before the barrier (inputs resident in HBM when the clock starts, as the bench contract asks); "
                               "the copies of the other shards cross PCIe inside the timed region, under proofs. `events_to_proof.unprimed_estimate` prices the first copies in"},
            "event_bytes_per_shard": lane.event_bytes,
            "h2d_GBps_per_rank": round(lane.event_bytes * len(ids) / elapsed / 1e9, 2),      # rank 0's events over the timed region: what its PCIe link carried
            "shards_proved": int(proved), "shards_proved_by_rank0": len(ids), "fewest_shards_on_a_rank": int(fewest), "lib_digest": lib_digest})
    if rank == 0 and not stub:
        farm_extras(args, fri, lane, lanes, line, world, elapsed, n_shards)
    t_post = time.perf_counter()
    if rank == 0:
        line["wall_s"] = {"setup": round(t_setup - t_start, 2), "warmup": round(t_warm - t_setup, 2), "timed_region": round(elapsed, 3),
                          "gather_inside_it": round(gather_s[0], 4), "closing_barrier_inside_it": round(gather_s[1], 4), "verification": round(t_verify - t_timed, 2), "resident_leg_and_extras": round(t_post - t_verify, 2)}
        print(json.dumps(line), flush=True)
    farm.barrier()          # the other ranks leave with rank 0
    farm.close()


def farm_extras(args, fri, lane, lanes, line, world, elapsed, n_shards):
    """Rank 0, after the timed region and outside `value`: the resident one-lane leg on the lane's first pool shard — `roofline`, `valu`,
    `lde`, `kernels_ms`, `resident_one_lane` — and, at N = 1, the continuity legs and the CPU baseline."""
    from ziren_amd import lib, prover, synth
    L = lib.load()
    wl = lane.wls[0]
    hp, pk, ch0, ctx = lane.hp, lane.pk, lane.ch0, lane.ctx
    for l in lanes[1:]:
        l.ctx.trim()                    # the other lanes' pools go back to the driver: the legs below run on the first lane's context
    alg = synth.shard_algorithmic_bytes(wl)
    steps = max(3, min(args.steps, 10))
    if hasattr(L, "zkm_ctx_set_host_wait"):
        L.zkm_ctx_set_host_wait(ctx.h, C.c_int(0))          # one lane alone: the spinning wait (lowest latency), as in rounds 1-4
    traces = wl.resident_traces(ctx)            # inputs resident in HBM before timing
    leg = resident_leg(_LocalTimer(ctx), wl, hp, pk, ch0, traces, steps, 2, args.kernel_timing)
    verify_or_die(wl, fri, ch0, leg["proof"], wl.tag + " resident")
    roofline, valu, lde, kernels_ms, ksource = roofline_objects(wl, fri, leg, steps)
    if roofline is not None:
        roofline["whole_shard_in_the_timed_region"] = {
            "algorithmic_bytes_per_shard": alg, "achieved": round(alg * n_shards / elapsed / 1e9, 1), "peak": HBM_PEAK_GBPS * world,
            "frac": round(alg * n_shards / elapsed / 1e9 / (HBM_PEAK_GBPS * world), 4),
            "note": "SURVEY 8(d)'s algorithmic bytes of one shard x shards / time of the timed region against N x 8 TB/s"}
    rms = leg["elapsed"] / steps * 1e3
    line.update({"roofline": roofline, "valu": valu, "lde": lde, "kernels_ms": kernels_ms, "kernels_ms_source": ksource,
                 "phases_ms": {n: round(v, 3) for n, v in leg["phases"].items()},
                 "resident_one_lane": {"value": round(steps / leg["elapsed"], 4), "unit": "shard-proofs/s", "ms_per_step": round(rms, 3), "steps": steps, "verified": True,
                                       "note": "rounds 1-4's headline: traces resident in HBM, one shard in flight, one context; after the timed region, not `value`"}})
    line["config"].update({"chips": {c.name: c.log_height for c in wl.chips},
                           "committed_cells": int(sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in wl.chips)),
                           "proof_words": int(len(leg["proof"])), "cycles": wl.cycles,
                           "shape": {"registered_under_log2_shard_size": wl.ds.shape_key[0], "cluster": wl.ds.shape_key[1], "shard_closed_by": wl.cut,
                                     "reference": "crates/core/machine/src/shape/mod.rs:139-191, crates/core/executor/src/executor.rs:2429-2516"}})
    if world > 1 or args.no_extra:
        for t in traces:
            t.free()
        return
    L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
    line["two_in_flight"] = two_in_flight_leg(wl, fri, lane.ctx.device, (hp, pk, ch0, traces, leg["out"]), not args.interpreter, steps)
    for t in traces:
        t.free()
    ev, proof = events_leg(wl, hp, pk, ch0, leg["out"], min(steps, 6))
    verify_or_die(wl, fri, ch0, proof, wl.tag + " events->proof")
    ev["verified"] = True
    # what `value` would be with the lanes' first uploads inside the clock: both lanes' copies share the link, so one lane's worth of copy
    # time (at the rate measured alone, above) is exposed at the start of the region
    exposed_s = lane.event_bytes * len(lanes) / (ev["h2d_GBps_alone"] * 1e9)
    ev["unprimed_estimate"] = {"value": round(n_shards / (elapsed + exposed_s), 4), "unit": "shard-proofs/s", "exposed_copy_ms": round(exposed_s * 1e3, 2),
                               "note": "value with the first shards' event copies (primed before the barrier) charged to the timed region at the H2D rate measured alone"}
    ev["proof_identical_to_resident_leg"] = bool(np.array_equal(proof, leg["proof"]))     # same events, same transcript: the same words
    if not ev["proof_identical_to_resident_leg"]:
        raise SystemExit("bench.py: the proof made from prefetched events differs from the one made from resident traces")
    line["events_to_proof"] = ev
    also = [a for a in args.also.split(",") if a] + [f"fibc{wl.log_size}"] + ([] if args.no_syn else ["syn22"])
    ctx.trim()
    line["other_workloads"] = other_workload_legs(also, fri, lane.ctx.device, not args.interpreter, steps) or None
    if not args.no_reduce:
        # the recursion-tree reduce over this run's shards (ziren_amd/reduce.py; stand-in programs at the reference's compress shapes): per-shape
        # ms, an 8-leaf tree through the farm's queue, and its share of core + reduce time at this line's ms per core shard
        ctx.trim()
        line["other_workloads"] = dict(line["other_workloads"] or {}, REDUCE=reduce_leg(lane.ctx.device, (8,), 3, elapsed / n_shards * 1e3))
    if not args.no_micro:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_micro
        ctx.trim()
        line["micro"] = bench_micro.micro_bench(ctx, (20, 21, 22), 64, 3)
    if not args.no_cpu_baseline:
        sample = args.cpu_sample_log if args.cpu_sample_log is not None else 18
        line["cpu_baseline"] = cpu_baseline_leg(wl, fri, not args.cpu_sample, sample)


def reduce_leg(device, leaves, steps, core_ms_per_shard):
    """tools/bench_reduce_tree.py's measurement, trimmed to what a line carries."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_reduce_tree
    r = bench_reduce_tree.reduce_bench(leaves=leaves, steps=steps, core_ms_per_shard=core_ms_per_shard, device=device)
    for leg in r["per_shape"].values():
        leg["kernels_ms"] = {k: v["ms"] for k, v in list(leg["kernels_ms"].items())[:6]}
        leg.pop("shape", None), leg.pop("fill", None)
    r["full_measurement"] = f"profiles/{PROFILE_ROUND}_reduce_tree.json (tools/bench_reduce_tree.py --leaves 8,16,32)"
    return r


def other_workload_legs(also, fri, device, specialize, steps):
    """Resident one-lane legs of other workloads, for continuity: fibc21 (the same record without the shape step), syn22 (rounds 1-3's
    headline), fib21 / fibs20 / syn20 ... on request (--also)."""
    from ziren_amd import prover
    extra = {}
    for name in also:
        if name.startswith("syn"):
            w2 = SynWorkload(int(name[3:]))
        elif name.startswith("fibs"):
            w2 = FibWorkload("shaped", int(name[4:]))
        elif name.startswith("fibc"):
            w2 = FibWorkload("cut", int(name[4:]))
        else:
            w2 = FibWorkload("tight", int(name[3:]))
        ctx2 = prover.Context(device)
        hp2, pk2, ch2 = w2.setup(ctx2, fri, specialize)
        tr2 = w2.resident_traces(ctx2)
        n2 = max(3, min(steps, 5))
        leg2 = resident_leg(_LocalTimer(ctx2), w2, hp2, pk2, ch2, tr2, n2, 1, 2)
        verify_or_die(w2, fri, ch2, leg2["proof"], w2.tag)
        t = leg2["table"]
        extra[name.upper()] = {
            "workload": w2.label, "ms_per_step": round(leg2["elapsed"] / n2 * 1e3, 3), "value": round(n2 / leg2["elapsed"], 4), "unit": "shard-proofs/s", "steps": n2,
            "verified": True, "chips": {c.name: c.log_height for c in w2.chips},
            "committed_cells": int(sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in w2.chips)),
            "kernels_ms": {k: round(v[0] / leg2["table_steps"], 3) for k, v in sorted(t.items(), key=lambda kv: -kv[1][0])[:8]}}
        for x in tr2:
            x.free()
        pk2.free()                     # before its context goes
        del hp2, pk2, tr2, leg2, w2
        ctx2.close()
    return extra


# ---- main ----------------------------------------------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracegen", action="store_true", help="benchmark device trace generation of the core chips instead of the shard proof (tools/bench_tracegen.py)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="shards per GPU in the timed region (K N in the queue)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up proofs per lane")
    ap.add_argument("--resident", action="store_true", help="N = 1 only: print the resident-only line of rounds 1-4 (traces in HBM, one shard in flight, no queue): "
                    "what the profiling scripts run; takes --workload / --log-rows")
    ap.add_argument("--no-reduce", action="store_true", help="skip the recursion-tree reduce leg (other_workloads.REDUCE)")
    ap.add_argument("--leaves", type=str, default="8,16,32", help="--workload reduce-tree: core proofs under the trees that are timed")
    ap.add_argument("--workload", choices=["fib", "fib-tight", "syn", "reduce-tree"], default="fib",
                    help="--resident only. fib (default): the fibonacci shard as the reference cuts and shapes it at SHARD_SIZE = 2^--shard-size-log; fib-tight: 2^--log-rows cycles, "
                         "tight heights; syn: SYN---log-rows")
    ap.add_argument("--shard-size-log", type=int, default=21, help="log2 of the reference's SHARD_SIZE (MAX_SHARD_SIZE = 2^21, crates/stark/src/opts.rs:6; "
                    "at 2^22 this guest's shard has no covering shape: fix_shape fails in the reference too)")
    ap.add_argument("--log-rows", type=int, default=None, help="fib-tight: log2 cycles (default 21, the most a shard's 24-bit clock holds); syn: log2 rows of the tallest chip (default 22)")
    ap.add_argument("--cpu-sample-log", type=int, default=None, help="size of the cpu_baseline sample (fib: log2 SHARD_SIZE, default 18; syn: log rows, default 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="(the default since round 6) time the CPU baseline on the benchmarked shard itself: ~100 s on 16 cores")
    ap.add_argument("--cpu-sample", action="store_true", help="time the CPU baseline on a smaller shard of the same guest (--cpu-sample-log) and scale it, instead of the ~100 s full-size run")
    ap.add_argument("--no-micro", action="store_true", help="skip the single-matrix LDE / Merkle micro-benchmarks (`micro`)")
    ap.add_argument("--no-extra", action="store_true", help="only the timed region, its verification and the resident leg's roofline objects")
    ap.add_argument("--no-syn", action="store_true", help="skip the SYN-22 continuity leg")
    ap.add_argument("--also", type=str, default="", help="comma-separated extra resident legs: fib21 (2^21 cycles, tight), fibs20 (shaped at SHARD_SIZE 2^20), fibc20 (the same record, tight), syn20 ...")
    ap.add_argument("--kernel-timing", type=int, default=3,
                    help="resident leg: 0 off, 1 every launch, 2 launches >= 256 KiB, 3 (default): inside its timed steps only the dominant kernel is timed")
    ap.add_argument("--interpreter", action="store_true", help="no generated kernels: the bytecode interpreter for the quotient, the generic kernel for the permutation traces")
    ap.add_argument("--inflight", type=int, default=2, help="shards in flight per GPU (a context + host thread each); 2 fills one proof's transcript "
                    "round trips and launch gaps with the other's kernels")
    ap.add_argument("--queue", type=int, default=0, metavar="SHARDS", help="deal SHARDS distinct shards in all through the claim queue (strong scaling) instead of --steps per GPU")
    args = ap.parse_args()
    if args.log_rows is None:
        args.log_rows = 21 if (args.workload == "fib-tight" or args.tracegen) else 22
    if args.tracegen:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_tracegen
        return bench_tracegen.tracegen_bench(args)

    if args.workload == "reduce-tree" and args.gpus == 1 and "WORLD_SIZE" not in os.environ:
        # the recursion-tree reduce on its own, one GPU, with everything measured: per-shape legs + trees under both schedules; one JSON line,
        # `value` = recursion shards per second of the largest tree
        r = reduce_leg(0, tuple(int(x) for x in args.leaves.split(",") if x), max(3, min(args.steps, 10)), None)
        t = r["trees"][-1]
        print(json.dumps({"metric": "recursion-shard-proofs/sec (reduce tree)", "value": round(t["recursion_shards"] / t["wall_ms"] * 1e3, 3), "unit": "shard-proofs/s", "n_gpus": 1,
                          "steps": t["recursion_shards"], "warmup": 1, "ms_per_step": t["ms_per_recursion_shard"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u32", "data": "synthetic (stand-in recursion programs, ziren_amd/reduce.py)", "verified": True,
                          "config": {"workload": f"reduce tree over {t['leaves']} core proofs: first layer + {len(t['layers']) - 2} reduce layers + shrink, compress-machine shards at the reference's shapes, pipelined"},
                          "reduce": r}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.workload == "reduce-tree":
        return reduce_tree_main(args)

    from ziren_amd import abi, farm as farm_mod
    # tests only: ZKM_BENCH_STUB_PROVER=1 (no GPU at all: gloo, stub lanes) and ZKM_BENCH_ONE_DEVICE=1 (a 1-GPU box standing in for N: every
    # rank proves on device 0 and the process group runs over gloo — RCCL refuses two ranks on one device)
    one_device = os.environ.get("ZKM_BENCH_ONE_DEVICE") == "1"
    fri = abi.FriConfig(1, 84, 16)  # core config, crates/stark/src/kb31_poseidon2.rs:203-213
    built = None
    if not args.resident:          # the lanes and their warm-up before torch.cuda / RCCL exist in this process: see build_and_warm_lanes
        rank_env, local_env, _ = farm_mod.env_rank_world()
        built = build_and_warm_lanes(args, fri, rank_env, 0 if one_device else local_env)
    farm = farm_mod.Farm(backend="gloo" if (os.environ.get("ZKM_BENCH_STUB_PROVER") == "1" or one_device) else None)
    if one_device:
        farm.local_rank = 0
    if farm.world != args.gpus:
        farm.close()
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {farm.world} rank(s) (WORLD_SIZE): refusing to print a line for another N")
    if args.resident:
        if farm.world != 1:
            raise SystemExit("bench.py: --resident is the one-GPU resident-trace line")
        return resident_main(args, farm, fri)
    return farm_main(args, farm, fri, built)


def reduce_tree_main(args):
    """`--workload reduce-tree --gpus N` (N > 1, or under a launcher): the reduce tree over N ranks — one process per GPU, two lanes each, all
    nodes of the tree one claim queue in the process group's store, children's words through the store, ONE gather of the proof streams to
    rank 0 at the end (RCCL on GPUs: `north_star`'s "RCCL/xGMI only for the final recursion-tree reduce"). `--leaves K` (the last of the list):
    K N leaves — K first-layer proofs per GPU, so the work per GPU is fixed as N grows (weak scaling; the tree gains log2 N reduce layers).
    W untimed trees (they build the keys), then `--steps` timed trees between barriers, max over ranks; rank 0 verifies the shrink proof and
    the root of the reduce layers of the last tree and prints one line: `value` = recursion shards per second of the whole job."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_reduce_tree as BRT
    from ziren_amd import farm as farm_mod, field as F, lib, prover, reduce as RD
    one_device = os.environ.get("ZKM_BENCH_ONE_DEVICE") == "1"
    rank_env, local_env, world_env = farm_mod.env_rank_world()
    device = 0 if one_device else local_env
    lib_digest = lib.check_build_identity()
    ctxs = [prover.Context(device) for _ in range(max(1, args.inflight))]        # the prover's streams before torch / RCCL make theirs
    lanes = [RD.ReduceLane(c) for c in ctxs]
    for c in ctxs:
        lib.load().zkm_ctx_set_host_wait(c.h, C.c_int(1))
    tree = RD.ReduceTree(RD.TreePlan(1, 0, 0), BRT.device_permute(ctxs[0]))
    per_gpu = [int(x) for x in args.leaves.split(",") if x][-1]
    n_leaves = per_gpu * world_env
    for nc in (1, 2):                    # every rank generates the same programs (seeded) before the clock
        tree.program(0, nc)
    tree.program(1, 1)
    farm = farm_mod.Farm(backend="gloo" if one_device else None)
    if one_device:
        farm.local_rank = 0
    if farm.world != args.gpus:
        farm.close()
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {farm.world} rank(s)")
    farm.device_sync = lambda: [c.synchronize() for c in ctxs]
    core = np.random.default_rng(7).integers(0, F.P, (n_leaves, RD.CHILD_WORDS), dtype=np.uint64)      # the same on every rank: child_words of the gathered core proofs
    for _ in range(max(1, args.warmup)):
        tree.run(farm, lanes, core)
    steps = max(1, min(args.steps, 5))
    farm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        streams, words = tree.run(farm, lanes, core)
    farm.barrier()
    elapsed = farm.max_over_ranks(time.perf_counter() - t0)
    proved = farm.sum_over_ranks(float(sum(l.host_s["nodes"] for l in lanes)))
    fewest = -farm.max_over_ranks(-float(sum(l.host_s["nodes"] for l in lanes)))
    if farm.rank == 0:
        layers = tree.layers(n_leaves)
        n_nodes = sum(len(nodes) for *_, nodes in layers)
        O = oracle()
        below = [core] + [w.astype(np.uint64) for w in words]
        salt, ok = 1, True
        for li, (nm, si, fri_cfg, nodes) in enumerate(layers):
            if li >= len(layers) - 2:
                i = len(nodes) - 1
                ok = ok and BRT.verify(O, tree.program(si, len(nodes[i])), fri_cfg, np.concatenate([below[li][c] for c in nodes[i]]), streams[li][i], salt=salt + i)
            salt += len(nodes)
        if not ok:
            raise SystemExit("bench.py: the verifier REJECTED a proof of the tree; nothing is reported for it")
        backend = "none (one process)" if farm.dist is None else "gloo" if one_device else "nccl (RCCL)"
        print(json.dumps({"metric": "recursion-shard-proofs/sec (reduce tree)", "value": round(n_nodes * steps / elapsed, 3), "unit": "shard-proofs/s", "n_gpus": farm.world,
                          "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u32", "data": "synthetic (stand-in recursion programs, ziren_amd/reduce.py)", "verified": True,
                          "config": {"workload": f"reduce tree over {n_leaves} core proofs ({per_gpu} per GPU): first layer (shape 1) + {len(layers) - 2} reduce layers + shrink (shape 0), "
                                                 f"{n_nodes} compress-machine shards at the reference's shapes per tree, pipelined; a step = one tree",
                                     "parallelism": f"{farm.world} GPU(s), one process each, {len(lanes)} lane(s) per GPU, one claim queue over all nodes, children's words through the store, "
                                                    f"{backend} gather of {4 * sum(len(p) for s in streams for p in s)} proof bytes to rank 0 at the end of every tree",
                                     "backend": backend, "ranks_in_process_group": farm.dist.get_world_size() if farm.dist is not None else 1},
                          "tree_ms": round(elapsed / steps * 1e3, 2), "recursion_shards_per_tree": n_nodes, "nodes_proved_in_all_warmup_included": int(proved), "fewest_nodes_on_a_rank_warmup_included": int(fewest),
                          "verified_proofs": "the root of the reduce layers and the shrink proof of the last tree", "lib_digest": lib_digest}), flush=True)
    farm.barrier()
    farm.close()


def resident_main(args, farm, fri):
    """`--resident`: rounds 1-4's N = 1 line — W warm-up + K timed proofs of one shard whose traces are resident in HBM, one in flight."""
    from ziren_amd import lib, prover
    wl = {"fib": lambda: FibWorkload("shaped", args.shard_size_log), "fib-tight": lambda: FibWorkload("tight", args.log_rows),
          "syn": lambda: SynWorkload(args.log_rows)}[args.workload]()
    ctx = prover.Context(farm.local_rank)
    farm.device_sync = ctx.synchronize
    hp, pk, ch0 = wl.setup(ctx, fri, not args.interpreter)
    traces = wl.resident_traces(ctx)            # inputs resident in HBM before timing
    leg = resident_leg(farm, wl, hp, pk, ch0, traces, args.steps, args.warmup, args.kernel_timing)
    verify_or_die(wl, fri, ch0, leg["proof"], wl.tag)
    steps = args.steps
    ms_per_step = leg["elapsed"] / steps * 1e3
    roofline, valu, lde, kernels_ms, ksource = roofline_objects(wl, fri, leg, steps)
    line = {"metric": "shard-proofs/sec", "value": round(steps / leg["elapsed"], 4), "unit": "shard-proofs/s", "n_gpus": 1, "steps": steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": wl.data, "verified": True,
            "config": {"workload": wl.label, "chips": {c.name: c.log_height for c in wl.chips},
                       "committed_cells": int(sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in wl.chips)),
                       "proof_words": int(len(leg["proof"])), "parallelism": "1 GPU x 1 shard in flight, traces resident (--resident)"},
            "phases_ms": {n: round(v, 3) for n, v in leg["phases"].items()}, "kernels_ms": kernels_ms, "kernels_ms_source": ksource,
            "roofline": roofline, "valu": valu, "lde": lde, "lib_digest": lib.check_build_identity()}
    if wl.kind != "syn":
        line["config"]["cycles"] = wl.cycles
    if wl.kind == "syn" and not args.no_extra:
        line["pcie_inclusive"] = pcie_leg(wl, hp, pk, ch0, leg["out"], steps)
    line["cpu_baseline"] = None if (args.no_cpu_baseline or args.no_extra) else cpu_baseline_leg(wl, fri, args.cpu_full and not args.cpu_sample, args.cpu_sample_log if args.cpu_sample_log is not None else (20 if wl.kind == "syn" else 18))
    print(json.dumps(line), flush=True)
    farm.close()


class _LocalTimer:
    """farm.timed for a leg outside the main timed region (one process, no barrier)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def timed(self, step, steps, warmup):
        for _ in range(warmup):
            step()
        self.ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.ctx.synchronize()
        return time.perf_counter() - t0


if __name__ == "__main__":
    main()
