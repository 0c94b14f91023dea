#!/usr/bin/env python3
"""Device trace generation of the precompile tables: events in host memory -> column-major Montgomery matrix in HBM, per chip.

  python tools/bench_precompile_tracegen.py [--log-events 12]

Events are synthetic but valid (the generators refuse events whose results do not follow from their inputs): SHA-256 blocks, Ed25519
additions and decompressions of multiples of the base point, Keccak calls. Reported: kernel time (HIP events), rows/s, bytes of trace per
second against the 8 TB/s HBM peak — and, for the Ed25519 chips, the modular inversions / square roots per second the rows contain."""
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

from ziren_amd import events as E, lib, prover


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-events", type=int, default=12)
    args = ap.parse_args()
    n = 1 << args.log_events
    import test_ed25519 as TE
    import test_keccak as TK
    import test_sha256 as TS
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    NPTS = min(n, 8192)      # distinct points: the witness bytes of distinct rows are what crowds the lookup tables
    pts = [TE.BASE]
    for _ in range(NPTS - 1):
        pts.append(E.ed25519_add(pts[-1], TE.BASE))
    ed_add = np.array([TE.ed_event(pts[int(rng.integers(0, NPTS))], pts[int(rng.integers(0, NPTS))], clk=100 + 3 * i, seed=i)[0] for i in range(n)])
    ed_dec = np.array([TE.dec_event(pts[i % NPTS][1], pts[i % NPTS][0] & 1, clk=100 + 3 * i, seed=i)[0] for i in range(n)])
    sha = [TS.sha_events([int(x) for x in rng.integers(0, 1 << 32, 16)], E.SHA256_IV, clk=100 + 60 * i, seed=i) for i in range(n // 16)]
    sha_ext, sha_cmp = np.array([x[0] for x in sha]), np.array([x[1] for x in sha])
    kec = np.concatenate([TK.sponge_blocks(E.keccak256_words(bytes(rng.integers(0, 256, 100, dtype=np.uint8))), clk=100 + 10 * i, seed=i)[0] for i in range(n // 16)])
    # short-Weierstrass and field-tower tables: 1024 distinct events each (Python big integers are slow), repeated to the row count — the
    # generators check every row's result, and 1024 rows of random bytes already meet every range-check counter
    import functools
    import test_fptower as TF
    import test_weierstrass as TW
    tile = lambda a: np.tile(a, n // len(a) + 1)[:n]      # noqa: E731
    distinct = min(n, 1024)
    curve_tables = []
    for curve in ("Secp256k1", "Bls12381"):
        pts_w = TW.multiples(curve, 64)
        adds = np.array([TW.wevent(curve, False, pts_w[int(rng.integers(1, 64))], pts_w[0], clk=300 + 10 * i, seed=i)[0] for i in range(distinct)])
        dbls = np.array([TW.wevent(curve, True, pts_w[int(rng.integers(0, 64))], clk=300 + 10 * i, seed=i)[0] for i in range(distinct)])
        curve_tables.append((curve + "AddAssign", tile(adds), functools.partial(lambda c, ev, h, blu: ctx.tracegen_weierstrass(c, False, ev, h, blu), curve), 1))
        curve_tables.append((curve + "DoubleAssign", tile(dbls), functools.partial(lambda c, ev, h, blu: ctx.tracegen_weierstrass(c, True, ev, h, blu), curve), 1))
    for field in ("Bn254", "Bls12381"):
        ev = np.concatenate([TF.some_events(field, "fp2_mul", n=119, seed=k)[0] for k in range(distinct // 128 + 1)])[:distinct]
        curve_tables.append((field + "Fp2MulAssign", tile(ev), functools.partial(lambda f, ev, h, blu: ctx.tracegen_fp_tower(f, "fp2_mul", ev, h, blu), field), 1))
    # the U256Field chips, the decompressions, the garbled-circuit checks and the Linux syscalls: distinct events repeated to the row count
    import random
    import test_garble as TG
    import test_sys_linux as TL
    import test_uint256 as TU
    rnd = random.Random(11)
    moduli = [rnd.choice([0, TU.BIG, rnd.randrange(1 << 255, 1 << 256)]) for _ in range(distinct)]      # y below the modulus: the quotient fits 256 bits
    u256 = np.array([TU.uevent(rnd.randrange(1 << 256), rnd.randrange(m or 1 << 256), m, clk=100 + 7 * i, seed=i)[0] for i, m in enumerate(moduli)])
    u2048 = np.array([TU.xevent(rnd.randrange(1 << 256), rnd.randrange(1 << 2048), clk=100 + 7 * i, seed=i)[0] for i in range(min(distinct, 256))])
    curve_tables.append(("Uint256MulMod", tile(u256), lambda ev, h, blu: ctx.tracegen_uint256_mul(ev, h, blu), 1))
    curve_tables.append(("U256XU2048Mul", tile(u2048), lambda ev, h, blu: ctx.tracegen_u256x2048_mul(ev, h, blu), 1))
    for curve in ("Secp256k1", "Bls12381"):
        pts_w = TW.multiples(curve, 64)
        dec = np.array([TW.devent(curve, pts_w[int(rng.integers(0, 64))][0], int(rng.integers(0, 2)), clk=300 + 10 * i, seed=i)[0] for i in range(distinct)])
        curve_tables.append((curve + "Decompress", tile(dec), functools.partial(lambda c, ev, h, blu: ctx.tracegen_weierstrass_decompress(c, ev, h, blu), curve), 1))
    calls, _ = TG.some_calls(seed=5, sizes=tuple([31] * (n // 32)))      # calls of 31 gates: 32 rows each
    curve_tables.append(("BooleanCircuitGarble", calls, lambda ev, h, blu: ctx.tracegen_boolean_circuit_garble(ev, h, blu), 1))
    codes = [E.SYS_BRK, E.SYS_MMAP, E.SYS_MMAP2, E.SYS_CLONE, E.SYS_FCNTL, E.SYS_READ, E.SYS_WRITE_LINUX, E.SYS_OPEN]
    linux = np.array([TL.levent(codes[i % 8], int(rng.integers(0, 4)), int(rng.integers(0, 1 << 24)), brk=int(rng.integers(0, 1 << 32)), heap=int(rng.integers(0, 1 << 31)),
                                clk=100 + 7 * i, seed=i) for i in range(distinct)])
    curve_tables.append(("SysLinux", tile(linux), lambda ev, h, blu: ctx.tracegen_sys_linux(ev, h, blu), 1))
    gen_s = time.perf_counter() - t0
    ctx = prover.Context(0)
    lib.load().zkm_ctx_set_kernel_timing(ctx.h, C.c_int(1))
    out = {"events_generated_in_python_seconds": round(gen_s, 1), "tables": {}}
    for name, ev, fn, rows_per_event in (("EdAddAssign", ed_add, ctx.tracegen_ed_add, 1), ("EdDecompress", ed_dec, ctx.tracegen_ed_decompress, 1),
                                         ("ShaExtend", sha_ext, ctx.tracegen_sha_extend, 48), ("ShaCompress", sha_cmp, ctx.tracegen_sha_compress, 80),
                                         ("KeccakSponge", kec, ctx.tracegen_keccak_sponge, 24), *curve_tables):
        best, bare = None, None
        for rep in range(7):
            blu = ctx.byte_lookups() if rep < 4 else None      # the last repetitions: the rows alone, no byte lookups counted
            t0 = time.perf_counter()
            m = fn(ev, -1, blu)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            kern = sum(ms for nm, ms, _, _ in ctx.kernel_timings() if nm.startswith("tracegen"))
            shape = (m.height, m.width)
            m.free()
            if blu is not None:
                blu.free()
                if rep and (best is None or kern < best[0]):
                    best = (kern, wall)
            elif bare is None or kern < bare:
                bare = kern
        kern, wall = best
        nbytes = 4 * shape[0] * shape[1]
        rec = {"events": int(len(ev)), "rows": shape[0], "columns": shape[1], "kernel_ms": round(kern, 3), "wall_ms": round(wall, 3),
               "rows_per_s": round(shape[0] / kern * 1e3), "trace_GBps": round(nbytes / kern / 1e6, 1), "frac_of_hbm_peak": round(nbytes / kern / 1e6 / 8000, 4), "kernel_ms_without_lookups": round(bare, 3)}
        if name == "EdAddAssign":
            rec["modular_inversions_per_s"] = round(2 * len(ev) / kern * 1e3)
        if name == "EdDecompress":
            rec["inversions_plus_square_roots_per_s"] = round(2 * len(ev) / kern * 1e3)
        out["tables"][name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
