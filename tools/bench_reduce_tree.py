#!/usr/bin/env python3
"""The recursion-tree reduce at the reference's own shapes, measured (ziren_amd/reduce.py).

  python tools/bench_reduce_tree.py [--leaves 8,16,32] [--leaf-shape 1] [--reduce-shape 0] [--steps 5] [--out profiles/r06_reduce_tree.json]

Per allowed shape of `RecursionShapeConfig::default()` (crates/recursion/core/src/shape.rs:134-171): one compress-machine shard of the
stand-in program, one lane, events in page-locked host memory -> device traces -> commit + open under (1, 84) — ms per shard, the
serialised per-kernel table, the fraction of the HBM roofline by SURVEY 8(d)'s algorithmic bytes, `setup_ms` (the proving key: preprocessed
traces + their commitment, what the reference redoes per proof at lib.rs:816 and this prover keeps per program); the shrink configuration
(2, 42) at the fastest shape. Then whole trees — first layer, reduce layers, shrink — for K leaves through `Farm.run_queue` with two lanes
on this GPU, layer by layer, wall-clock. Every figure belongs to proofs the restated verifier accepted (outside the timed loops).
"""
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

HBM_PEAK_GBPS = 8000.0


def device_permute(ctx):
    from ziren_amd import field as F, prover
    return lambda v: F.from_monty(prover.poseidon2_permute_batch(ctx, F.to_monty(np.asarray(v, dtype=np.uint64))))


def verify(O, prog, fri, inputs, proof, salt=0):
    """The restated verifier on one recursion shard proof (the oracle commits the preprocessed traces itself)."""
    from test_reduce import host_shard, oracle_key
    from ziren_amd import abi, synth
    recs = host_shard(prog, O, inputs)
    opk, och = oracle_key(O, recs, fri[0])
    if salt:
        O.challenger_observe(och, np.array([salt], dtype=np.uint32))
    return O.verify_shard(opk, recs, abi.FriConfig(*fri), synth.NUM_PV_ELTS, och, np.ascontiguousarray(proof, dtype=np.uint32).copy()) == 0


def shard_leg(lane, prog, shape_idx, fri, steps, O=None):
    """One lane, one shard in flight: W = 2 warm-up proofs, `steps` timed ones; then the serialised per-kernel pass."""
    from ziren_amd import lib, reduce as RD, synth
    L = lib.load()
    ctx = lane.ctx
    inputs = (np.arange(prog.n_inputs, dtype=np.uint64) * 2654435761 + 17 * shape_idx + fri[0]) % np.uint64(0x7F000001)
    pid = ("leg", shape_idx, prog.n_inputs)
    L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
    for _ in range(2):
        proof = lane.prove(pid, prog, shape_idx, fri, inputs).copy()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lane.prove(pid, prog, shape_idx, fri, inputs)
    ctx.synchronize()
    serial_ms = (time.perf_counter() - t0) / steps * 1e3
    # pipelined, as the tree's lanes run: the next node's events are queued on the DMA stream right before the current node's proof
    h = lane.prefetch(prog, inputs)
    lane.prove(pid, prog, shape_idx, fri, inputs, handle=h)
    h = lane.prefetch(prog, inputs)
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        nxt = lane.prefetch(prog, inputs) if i + 1 < steps else None
        lane.prove(pid, prog, shape_idx, fri, inputs, handle=h)
        h = nxt
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the same with the traces resident: what the proof alone takes
    hp, recs, pk, ch0 = lane.key_for(pid, prog, shape_idx, fri)
    w = prog.witness(inputs)
    born = lane.traces(prog, recs, w)
    pv = prog.public_values(w["digest"])
    hp.prove_shard(pk, pv, born, ch0.copy(), out=lane.out)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hp.prove_shard(pk, pv, born, ch0.copy(), out=lane.out)
    ctx.synchronize()
    resident_ms = (time.perf_counter() - t0) / steps * 1e3
    phases = dict(ctx.last_timings())
    # per-kernel table: side-stream overlap off, every launch >= 256 KiB timed
    L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(2))
    L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(0))
    hp.prove_shard(pk, pv, born, ch0.copy(), out=lane.out)
    table, n = {}, min(steps, 3)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        hp.prove_shard(pk, pv, born, ch0.copy(), out=lane.out)
        for name, kms, calls, nbytes in ctx.kernel_timings():
            a = table.setdefault(name, [0.0, 0, 0.0])
            a[0] += kms; a[1] += calls; a[2] += nbytes
    table_ms = (time.perf_counter() - t0) / n * 1e3
    L.zkm_ctx_set_lde_overlap(ctx.h, C.c_int(1))
    L.zkm_ctx_set_kernel_timing(ctx.h, C.c_int(0))
    for t in born:
        t.free()
    wl = type("W", (), {"chips": recs})
    alg = synth.shard_algorithmic_bytes(wl)
    cells = int(sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in recs))
    event_bytes = int(sum(4 * len(prog.streams[spec[1]]) for name, spec in zip(RD.CHIP_ORDER, RD.chip_specs()) if spec[1] is not None and name != "ExpReverseBitsLen")
                      + 4 * (len(prog.streams["exp_bases"]) + len(prog.streams["exp_bits"]) + len(prog.streams["exp_offsets"])))
    dom = max(table.items(), key=lambda kv: kv[1][0])
    out = {"shape": {c: prog.shape[c] for c in RD.CHIP_ORDER}, "fri": {"log_blowup": fri[0], "queries": fri[1], "pow_bits": fri[2]},
           "fill": prog.fill(), "committed_cells": cells, "proof_words": int(len(proof)), "event_bytes": event_bytes,
           "ms_per_shard_from_events": round(ms, 3), "ms_per_shard_from_events_unpipelined": round(serial_ms, 3), "ms_per_shard_traces_resident": round(resident_ms, 3),
           "setup_ms_once_per_program": round(lane.setup_ms[(pid, tuple(fri))], 3),
           "phases_ms": {k: round(v, 3) for k, v in phases.items()},
           "kernels_ms": {k: {"ms": round(v[0] / n, 3), "launches": v[1] // n, "GBps": round(v[2] / max(v[0], 1e-9) / 1e6, 1)} for k, v in sorted(table.items(), key=lambda kv: -kv[1][0])},
           "kernels_ms_sum": round(sum(v[0] for v in table.values()) / n, 3), "ms_of_the_serialised_pass": round(table_ms, 3),
           "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "achieved": round(alg / (resident_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(alg / (resident_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "of": "the whole shard (SURVEY 8d: sum over chips of n (36 m + 36 p + 24 r + 12 q + 672)), traces resident",
                        "dominant_kernel": {"name": dom[0], "ms_per_launch": round(dom[1][0] / max(dom[1][1], 1), 4), "algorithmic_bytes_per_launch": int(dom[1][2] / max(dom[1][1], 1)),
                                            "frac": round(dom[1][2] / max(dom[1][0], 1e-9) / 1e6 / HBM_PEAK_GBPS, 4)}}}
    if O is not None:
        out["verified"] = bool(verify(O, prog, fri, inputs, proof))
        if not out["verified"]:
            raise SystemExit(f"bench_reduce_tree: the verifier REJECTED the shape-{shape_idx} proof")
    return out


def two_lane_leg(lanes, prog, shape_idx, fri, steps):
    """Both lanes of the GPU proving the same program back to back (a host thread each): ms per shard from events, and with the traces
    resident — what the tree's wide layers can reach, and how much of it the events' path costs."""
    import threading
    inputs = (np.arange(prog.n_inputs, dtype=np.uint64) * 977 + 5) % np.uint64(0x7F000001)
    pid = ("two", shape_idx, prog.n_inputs)
    for l in lanes:
        l.prove(pid, prog, shape_idx, fri, inputs)
    res = {}

    def timed(fn):
        for l in lanes:
            l.ctx.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=fn, args=(l,)) for l in lanes]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for l in lanes:
            l.ctx.synchronize()
        return (time.perf_counter() - t0) / (steps * len(lanes)) * 1e3

    def from_events(l):
        h = l.prefetch(prog, inputs)
        for i in range(steps):
            nxt = l.prefetch(prog, inputs) if i + 1 < steps else None
            l.prove(pid, prog, shape_idx, fri, inputs, handle=h)
            h = nxt

    for l in lanes:
        l.host_s = {k: 0.0 for k in l.host_s}
    res["from_events_ms_per_shard"] = round(timed(from_events), 3)
    n = max(1, sum(l.host_s["nodes"] for l in lanes))
    res["from_events_host_ms_per_node"] = {k: round(1e3 * sum(l.host_s[k] for l in lanes) / n, 3) for k in ("prefetch", "traces", "prove_shard", "free")}
    state = {}
    for l in lanes:
        hp, recs, pk, ch0 = l.key_for(pid, prog, shape_idx, fri)
        w = prog.witness(inputs)
        state[id(l)] = (hp, pk, ch0, l.traces(prog, recs, w), prog.public_values(w["digest"]))

    def resident(l):
        hp, pk, ch0, born, pv = state[id(l)]
        for _ in range(steps):
            hp.prove_shard(pk, pv, born, ch0.copy(), out=l.out)

    resident(lanes[0]); resident(lanes[1])
    res["traces_resident_ms_per_shard"] = round(timed(resident), 3)
    for hp, pk, ch0, born, pv in state.values():
        for t in born:
            t.free()
    return res


def tree_leg(tree, farm, lanes, n_leaves, O=None, repeat=2):
    """K leaves -> one shrink proof, layer by layer through the farm's queue; the best of `repeat` runs (the first builds the keys)."""
    from ziren_amd import field as F, reduce as RD
    rng = np.random.default_rng(n_leaves)
    core = rng.integers(0, F.P, (n_leaves, RD.CHILD_WORDS), dtype=np.uint64)
    best, best_layered = None, None
    for _ in range(repeat + 1):
        for l in lanes:
            l.ctx.synchronize()
        t0, c0 = time.perf_counter(), time.process_time()
        streams, words = tree.run(farm, lanes, core)
        dt = time.perf_counter() - t0
        cores = (time.process_time() - c0) / dt
        if best is None or dt < best[0]:
            best = (dt, list(tree.layer_seconds), streams, words, cores)
        for l in lanes:
            l.ctx.synchronize()
        t0 = time.perf_counter()
        s2, w2 = tree.run(farm, lanes, core, pipelined=False)
        dt2 = time.perf_counter() - t0
        if best_layered is None or dt2 < best_layered[0]:
            best_layered = (dt2, list(tree.layer_seconds))
        assert all(np.array_equal(a, b) for a, b in zip(words, w2)) and all(np.array_equal(p, q) for x, y in zip(streams, s2) for p, q in zip(x, y)), "the two schedules made different proofs"
    dt, layers, streams, words, cores = best
    n_proofs = sum(len(s) for s in streams)
    res = {"cores_busy": round(cores, 2), "leaves": n_leaves, "recursion_shards": n_proofs, "wall_ms": round(dt * 1e3, 2), "ms_per_recursion_shard": round(dt * 1e3 / n_proofs, 3),
           "schedule": "pipelined: one queue over all nodes in layer order, a node waits only for its own children (as the reference's channels do, lib.rs:655-915); one gather at the end",
           "layers": [{"layer": nm, "nodes": n, "last_node_finished_at_ms": round(s * 1e3, 2)} for nm, n, s in layers],
           "layer_by_layer": {"wall_ms": round(best_layered[0] * 1e3, 2), "layers": [{"layer": nm, "nodes": n, "ms": round(s * 1e3, 2)} for nm, n, s in best_layered[1]],
                              "note": "the same nodes with a barrier, an all-reduce of the witnessed words and a gather per layer (ReduceTree.run_layers): same proofs, word for word"},
           "proof_bytes_gathered": int(4 * sum(len(p) for s in streams for p in s))}
    if O is not None:      # the shrink proof (the tree's output) and the root of the reduce layers through the verifier
        specs = tree.layers(n_leaves)
        below = [core] + [w.astype(np.uint64) for w in words]
        salt = 1
        ok = True
        for li, (nm, si, fri, nodes) in enumerate(specs):
            if li >= len(specs) - 2:
                i = len(nodes) - 1
                prog = tree.program(si, len(nodes[i]))
                ok = ok and verify(O, prog, fri, np.concatenate([below[li][c] for c in nodes[i]]), streams[li][i], salt=salt + i)
            salt += len(nodes)
        if not ok:
            raise SystemExit("bench_reduce_tree: the verifier REJECTED a proof of the tree")
        res["verified"] = "the root of the reduce layers and the shrink proof"
    return res


def reduce_bench(leaves=(8, 16, 32), leaf_shape=1, reduce_shape=0, shrink_shape=0, steps=5, core_ms_per_shard=None, device=0, check=True, shapes_to_time=(0, 1, 2), n_lanes=2):
    from ziren_amd import farm as farm_mod, lib, prover, reduce as RD
    lib.check_build_identity()
    O = None
    if check:
        import bench
        O = bench.oracle()
    ctxs = [prover.Context(device) for _ in range(max(2, n_lanes))]
    lanes = [RD.ReduceLane(c) for c in ctxs]
    shapes = RD.load_shapes()
    tree = RD.ReduceTree(RD.TreePlan(leaf_shape, reduce_shape, shrink_shape), device_permute(ctxs[0]))
    t0 = time.perf_counter()
    per_shape = {}
    for si in shapes_to_time:
        prog = tree.program(si, 2 if si == reduce_shape else 1)
        per_shape[f"shape{si}_compress_1_84"] = shard_leg(lanes[0], prog, si, RD.COMPRESS_FRI, steps, O)
    per_shape[f"shape{shrink_shape}_shrink_2_42"] = shard_leg(lanes[0], tree.program(shrink_shape, 1), shrink_shape, RD.SHRINK_FRI, steps, O)
    two = {f"shape{si}": two_lane_leg(lanes, tree.program(si, 2 if si == reduce_shape else 1), si, RD.COMPRESS_FRI, steps) for si in shapes_to_time}
    for l in lanes:
        l.close()
        l.ctx.trim()
    gen_s = time.perf_counter() - t0
    # the tree's lanes sleep between stream queries, as the core farm's do (zkm_ctx_set_host_wait 1): 0.05 CPU-seconds per shard instead of a
    # spinning core per lane, same wall time (alternated on one box: 104.3 / 368.8 ms spinning, 103.8 / 367.8 sleeping for 8 / 32 leaves)
    for l in lanes:
        lib.load().zkm_ctx_set_host_wait(l.ctx.h, C.c_int(1))
    f = farm_mod.Farm()
    trees = [tree_leg(tree, f, lanes, k, O) for k in leaves]
    host = {"lanes_wait": "sleeping between stream queries (zkm_ctx_set_host_wait 1)", "cores_busy_while_a_tree_runs": [t.pop("cores_busy") for t in trees]}
    out = {"what": "the recursion-tree reduce (crates/prover/src/lib.rs:617-957) at the reference's compress shapes (crates/recursion/core/src/shape.rs:134-171), one MI355X",
           "program": "STAND-IN (ziren_amd/reduce.py): the reference's nine chips, heights and widths, events filling 3/4 of every height, balanced memory lookups, "
                      "the children's commitments + digests witnessed and absorbed into the committed digest; not a verifier of its children (that program comes out of the Rust recursion compiler)",
           "plan": {"first_layer_shape": leaf_shape, "reduce_layers_shape": reduce_shape, "shrink_shape": shrink_shape,
                    "note": "which allowed shape a first-layer / reduce / shrink program lands in is the recursion compiler's output: an assumption here; ms per shard is given for all three"},
           "per_shape": per_shape, "two_lanes": two, "trees": trees, "host": host,
           "lanes": len(lanes), "tree_mode": "lanes (a context + host thread each) on the one GPU claim nodes from the tree's queue; children's words through the board (a dictionary in one process, the process group's store across ranks); one gather_proofs at the end; proving keys kept per program",
           "lib_digest": lib.check_build_identity(), "seconds_of_program_generation_and_shape_legs": round(gen_s, 1)}
    if core_ms_per_shard:
        out["share_of_a_fibonacci_run"] = [{"core_shards": t["leaves"], "core_ms": round(core_ms_per_shard * t["leaves"], 1), "reduce_tree_ms": t["wall_ms"],
                                            "reduce_share_of_core_plus_reduce": round(t["wall_ms"] / (t["wall_ms"] + core_ms_per_shard * t["leaves"]), 4)} for t in trees]
        out["core_ms_per_shard_used"] = core_ms_per_shard
    for l in lanes:
        l.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves", type=str, default="8,16,32")
    ap.add_argument("--leaf-shape", type=int, default=1)
    ap.add_argument("--reduce-shape", type=int, default=0)
    ap.add_argument("--shrink-shape", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--core-ms", type=float, default=None, help="ms per core shard (bench.py's ms_per_shard on this box) for the share figure")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--lanes", type=int, default=2, help="contexts (+ host threads) on the GPU claiming nodes")
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    res = reduce_bench([int(x) for x in args.leaves.split(",") if x], args.leaf_shape, args.reduce_shape, args.shrink_shape, args.steps, args.core_ms, check=not args.no_check, n_lanes=args.lanes)
    txt = json.dumps(res, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
