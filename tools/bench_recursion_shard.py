#!/usr/bin/env python3
"""A recursion-machine shard on the device: the nine chips of the compress machine (BaseAlu, ExtAlu, MemoryConst, MemoryVar, Select, Poseidon2Wide,
ExpReverseBitsLen, BatchFRI, PublicValues) over one balanced synthetic program (ziren_amd/recursion.py), traces built on the device, commit + open under the FRI
configuration of the reference's compress prover (`InnerSC::default()`: log_blowup 1, 84 queries; crates/prover/src/lib.rs:192), `--shrink`: the shrink prover's
(`InnerSC::compressed()`: 2 / 42; :196), `--ultra`: the ultra-compressed KoalaBear configuration (3 / 28; kb31_poseidon2.rs:229-241), with generated quotient kernels.
`--wrap-chips` swaps in the wrap machine's chip set (Poseidon2Skinny, BatchFRI at DEGREE 9) under the ultra-compressed configuration — the chips, not the reference's wrap
prover, which commits with a BN254 hasher (lib.rs:199).

  python tools/bench_recursion_shard.py [--log-hashes 16] [--steps 3]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from ziren_amd import abi, field as F, lib, prover, recursion as R, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-hashes", type=int, default=16, help="log2 of the number of Poseidon2 permutations in the program")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--shrink", action="store_true", help="shrink prover's configuration: log_blowup 2, 42 queries")
    ap.add_argument("--ultra", action="store_true", help="ultra-compressed KoalaBear configuration: log_blowup 3, 28 queries")
    ap.add_argument("--wrap-chips", dest="wrap", action="store_true",
                    help="the wrap machine's chips (Poseidon2Skinny and BatchFRI at DEGREE 9, no ExpReverseBitsLen) under the ultra-compressed KoalaBear configuration")
    args = ap.parse_args()
    ctx = prover.Context(0)
    n_hash = 1 << args.log_hashes

    def permute(vals):   # canonical (n, 16) -> canonical, on the device
        return F.from_monty(prover.poseidon2_permute_batch(ctx, F.to_monty(vals)))

    t0 = time.perf_counter()
    prog = R.balanced_program(n_hash // 2, n_hash // 4, 512, seed=1, n_var=4096, n_select=n_hash // 8, n_poseidon2=n_hash, permute_batch=permute,
                              n_exp=0 if args.wrap else n_hash // 64, n_batch_fri=n_hash // 32, commit_public_values=True)
    if args.wrap:
        prog["skinny_prep"] = F.to_monty(R.poseidon2_skinny_prep(prog["poseidon2_instrs"])).reshape(-1)
    gen_s = time.perf_counter() - t0
    specs = [("base_instrs", "base_events", R.ENTRIES_PER_ROW * R.ACCESS_COLS, R.ENTRIES_PER_ROW * R.BASE_VALUE_COLS, R.ENTRIES_PER_ROW, lambda lh, i: R.record_chip(False, lh, i)),
             ("ext_instrs", "ext_events", R.ENTRIES_PER_ROW * R.ACCESS_COLS, R.ENTRIES_PER_ROW * R.EXT_VALUE_COLS, R.ENTRIES_PER_ROW, lambda lh, i: R.record_chip(True, lh, i)),
             ("mem_entries", None, R.CONST_MEM_ENTRIES_PER_ROW * R.CONST_MEM_ENTRY_COLS, 1, R.CONST_MEM_ENTRIES_PER_ROW, R.record_mem_const),
             ("var_prep", "var_values", 2 * R.VAR_MEM_ENTRIES_PER_ROW, 4 * R.VAR_MEM_ENTRIES_PER_ROW, R.VAR_MEM_ENTRIES_PER_ROW, R.record_mem_var),
             ("select_prep", "select_events", R.SELECT_PREP_COLS, R.SELECT_COLS, 1, R.record_select),
             ("poseidon2_prep", "poseidon2_events", R.POSEIDON2_WIDE_PREP_WIDTH, R.POSEIDON2_WIDE_WIDTH, 1, R.record_poseidon2_wide),
             ("exp_prep", "exp_main", R.EXP_REVERSE_BITS_PREP_COLS, R.EXP_REVERSE_BITS_COLS, 1, R.record_exp_reverse_bits),
             ("batch_fri_prep", "batch_fri_main", R.BATCH_FRI_PREP_COLS, R.BATCH_FRI_COLS, 1, R.record_batch_fri),
             ("pv_prep", "pv_main", R.PUBLIC_VALUES_PREP_COLS, 1, 1, lambda lh, i: R.record_public_values(i))]
    if args.wrap:
        specs = [sp for sp in specs if sp[0] not in ("exp_prep", "poseidon2_prep", "batch_fri_prep")]
        specs.insert(5, ("skinny_prep", "poseidon2_events", R.SKINNY_PREP_WIDTH, R.SKINNY_WIDTH, 1, lambda lh, i: R.record_poseidon2_skinny(lh, i, degree=9)))
        specs.insert(6, ("batch_fri_prep", "batch_fri_main", R.BATCH_FRI_PREP_COLS, R.BATCH_FRI_COLS, 1, lambda lh, i: R.record_batch_fri(lh, i, degree=9)))
    recs, preps, mains = [], [], []
    for idx, (pk_key, ev_key, pw, mw, per_row, record) in enumerate(specs):
        n_rec = len(prog[pk_key]) // (pw // per_row)
        lh = R.PUBLIC_VALUES_LOG_HEIGHT if pk_key == "pv_prep" else (R.padded_rows(n_rec, -1, per_row)).bit_length() - 1
        recs.append(record(lh, idx))
        preps.append(ctx.tracegen_flat(prog[pk_key], pw, lh))
        mains.append((ev_key, mw, lh))
    fri = abi.FriConfig(3, 28, 16) if args.ultra or args.wrap else abi.FriConfig(2, 42, 16) if args.shrink else abi.FriConfig(1, 84, 16)
    hp = prover.HipProver(recs, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=True)
    pk = hp.setup(preps, [int(r.local_only) for r in recs], F.to_monty(0), F.to_monty(np.zeros(14, dtype=np.uint64)))
    ch0 = prover.new_challenger()
    pk.observe_into(ch0)
    pvv = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint64)
    pvv[R.PV_DIGEST_POS:R.PV_DIGEST_POS + 8] = prog["pv_digest"]
    pv = F.to_monty(pvv)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    out = np.zeros(1 << 23, dtype=np.uint32)
    res = []
    for it in range(args.steps + 1):
        t0 = time.perf_counter()
        born = []
        for (ev_key, mw, lh), r in zip(mains, recs):
            if r.name == "Poseidon2WideDeg3":
                born.append(ctx.tracegen_poseidon2_wide(prog[ev_key], lh))
            elif r.name.startswith("Poseidon2Skinny"):
                born.append(ctx.tracegen_poseidon2_skinny(prog[ev_key], lh))
            elif r.name == "ExpReverseBitsLen":
                born.append(ctx.tracegen_exp_reverse_bits(prog["exp_bases"], prog["exp_bits"], prog["exp_offsets"], lh))
            elif ev_key is None:
                born.append(ctx.tracegen_flat(np.zeros(0, dtype=np.uint32), mw, lh))
            else:
                born.append(ctx.tracegen_flat(prog[ev_key], mw, lh))
        t1 = time.perf_counter()
        proof = hp.prove_shard(pk, pv, born, ch0.copy(), out=out)
        t2 = time.perf_counter()
        phases = dict(ctx.last_timings())
        kern = {nm: (round(ms, 3), calls) for nm, ms, calls, _ in ctx.kernel_timings()}
        for m in born:
            m.free()
        if it:
            res.append({"tracegen_ms": (t1 - t0) * 1e3, "prove_ms": (t2 - t1) * 1e3, "phases": phases, "kernels": kern})
    r = res[-1]
    label = ("the wrap machine's chips, ultra-compressed KoalaBear" if args.wrap else "ultra-compressed KoalaBear" if args.ultra else
             "shrink prover's" if args.shrink else "compress prover's")
    cells = sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in recs)
    print(json.dumps({"workload": f"REC-{args.log_hashes}: balanced recursion program, 2^{args.log_hashes} Poseidon2 permutations + ALU / select / memory "
                                  f"instructions; {label} FRI configuration (blow-up {1 << fri.log_blowup}, {fri.num_queries} queries)",
                      "program_generation_seconds_python": round(gen_s, 1),
                      "tracegen_ms": round(float(np.mean([x["tracegen_ms"] for x in res])), 3),
                      "prove_ms": round(float(np.mean([x["prove_ms"] for x in res])), 3), "committed_cells": cells, "proof_words": int(len(proof)),
                      "chips": {c.name: {"rows": 1 << c.log_height, "prep": c.prep_width, "main": c.main_width, "perm_ext": c.perm_ext_width,
                                         "constraints": c.num_constraints} for c in recs},
                      "phases_ms": {nm: round(v, 3) for nm, v in r["phases"].items()}, "kernels_ms": r["kernels"]}, indent=1))


if __name__ == "__main__":
    main()
