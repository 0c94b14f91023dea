"""The short-Weierstrass precompiles (crates/core/machine/src/syscall/precompiles/weierstrass/weierstrass_add.rs, weierstrass_double.rs): eight
chips — Secp256k1, Secp256r1, Bn254, Bls12381 x AddAssign, DoubleAssign — from one parameterised implementation on each side, as in the
reference. Nine / eleven FieldOpCols per row (the same gadget as the Ed25519 chips, over the curve's base field: 32 byte limbs, 48 for
Bls12381). Pinned by the reference's cost table (4013 / 4492, Bls12381 6045 / 6772: widths, lookup counts and degree all enter), by the
reference's MODULUS bytes and generators (events.WEIERSTRASS_CURVES is checked against crates/curves when the tree is there), and by
Python-integer curve arithmetic (every result is on the curve; G + 2G = 3G = 2G + G)."""
import json
import os

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine

CURVES = list(E.WEIERSTRASS_CURVES)


def on_curve(curve, pt):
    c = E.WEIERSTRASS_CURVES[curve]
    x, y = pt
    return (y * y - x * x * x - c["a"] * x - c["b"]) % c["p"] == 0


def wevent(curve, double, p, q=None, shard=2, clk=300, p_ptr=0x800000, q_ptr=0x800200, seed=0):
    """The flattened EllipticCurveAddEvent / EllipticCurveDoubleEvent of <CURVE>_ADD(p_ptr, q_ptr) / <CURVE>_DOUBLE(p_ptr)
    (create_ec_add_event / create_ec_double_event, events/precompiles/ec.rs:96-176)."""
    rng = np.random.default_rng(seed)
    c = E.WEIERSTRASS_CURVES[curve]
    W = c["n_limbs"] // 2
    add_dt, dbl_dt = E.weierstrass_event_dtypes(curve)
    e = np.zeros(1, dtype=dbl_dt if double else add_dt)[0]
    e["shard"], e["clk"], e["p_ptr"] = shard, clk, p_ptr
    r = E.weierstrass_double(curve, p) if double else E.weierstrass_add(curve, p, q)
    words = lambda pt: [(pt[k // (W // 2)] >> (32 * (k % (W // 2)))) & 0xffffffff for k in range(W)]      # noqa: E731
    pw, rw = words(p), words(r)
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    for k in range(W):
        e["p_memory_records"][k] = (rw[k], shard, clk + (0 if double else 1), pw[k]) + prev()
    if not double:
        e["q_ptr"] = q_ptr
        qw = words(q)
        for k in range(W):
            e["q_memory_records"][k] = (qw[k], shard, clk) + prev()
    return e, r


def devent(curve, x, sign_bit, shard=2, clk=300, ptr=0x810000, seed=0):
    """The flattened EllipticCurveDecompressEvent of <CURVE>_DECOMPRESS(ptr, sign_bit) (create_ec_decompress_event, events/precompiles/ec.rs:181-228):
    x read at ptr + N, y written at ptr, both at clk."""
    rng = np.random.default_rng(seed)
    W = E.WEIERSTRASS_CURVES[curve]["n_limbs"] // 4
    e = np.zeros(1, dtype=E.weierstrass_decompress_event_dtype(curve))[0]
    e["shard"], e["clk"], e["ptr"], e["sign_bit"] = shard, clk, ptr, sign_bit
    y = E.weierstrass_decompress(curve, x, sign_bit)
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    for k in range(W):
        e["x_memory_records"][k] = ((x >> (32 * k)) & 0xffffffff, shard, clk) + prev()
        e["y_memory_records"][k] = ((y >> (32 * k)) & 0xffffffff, shard, clk, int(rng.integers(0, 1 << 32))) + prev()
    return e, y


def some_decompressions(curve, n=6):
    pts = multiples(curve, n)
    evs = [devent(curve, pts[i][0], i & 1, clk=300 + 10 * i, seed=i) for i in range(n)]
    return np.array([e for e, _ in evs]), [(pts[i][0], y) for i, (_, y) in enumerate(evs)]


def multiples(curve, n):
    g = E.WEIERSTRASS_CURVES[curve]["generator"]
    pts = [g, E.weierstrass_double(curve, g)]
    while len(pts) < n:
        pts.append(E.weierstrass_add(curve, pts[-1], g))
    return pts


def some_events(curve):
    pts = multiples(curve, 7)
    adds = [wevent(curve, False, pts[i + 1], pts[i], clk=300 + 10 * i, seed=i) for i in range(5)]
    dbls = [wevent(curve, True, pts[i], clk=500 + 10 * i, seed=9 + i) for i in range(5)]
    return np.array([x[0] for x in adds]), [x[1] for x in adds], np.array([x[0] for x in dbls]), [x[1] for x in dbls]


def test_curve_parameters_and_arithmetic():
    for curve, c in E.WEIERSTRASS_CURVES.items():
        g = c["generator"]
        assert on_curve(curve, g)
        two = E.weierstrass_double(curve, g)
        three = E.weierstrass_add(curve, g, two)
        assert on_curve(curve, two) and on_curve(curve, three) and three == E.weierstrass_add(curve, two, g)
        assert E.weierstrass_double(curve, two) == E.weierstrass_add(curve, three, g)
    ref = "/root/reference/crates/curves/src/weierstrass"
    if os.path.isdir(ref):      # the reference tree is only there in the build container
        import re
        for curve, fname in (("Secp256k1", "secp256k1"), ("Secp256r1", "secp256r1"), ("Bn254", "bn254"), ("Bls12381", "bls12_381")):
            src = open(os.path.join(ref, fname + ".rs")).read()
            m = re.search(r"const MODULUS: &'static \[u8\] = &\[(.*?)\];", src, re.S)
            assert int.from_bytes(bytes(int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", m.group(1))), "little") == E.WEIERSTRASS_CURVES[curve]["p"]


@pytest.mark.parametrize("curve", CURVES)
def test_weierstrass_rows_satisfy_the_airs_and_cost_what_the_reference_says(oracle, curve):
    adds, add_sums, dbls, dbl_sums = some_events(curve)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    c = E.WEIERSTRASS_CURVES[curve]
    N, W = c["n_limbs"], c["n_limbs"] // 2
    per_gadget = 2 * (N // 2) + 2 * ((2 * N - 2) // 2)
    for double, evs, sums in ((False, adds, add_sums), (True, dbls, dbl_sums)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        t = oracle.tracegen_weierstrass(curve, double, evs, -1, counts)      # refuses events whose p records do not hold the result
        width = E.weierstrass_widths(curve)[1 if double else 0]
        assert t.shape == (16, width)
        assert counts.sum() == len(evs) * ((11 if double else 9) * per_gadget + 2 * W * (1 if double else 2))
        tc = F.from_monty(t)
        assert air.debug_constraints(chips.record_weierstrass_constraints(curve, double).b, tc) == []
        gadgets = (4 + 13 * W) if double else (5 + 22 * W)
        G = 6 * N - 4
        for i, pt in enumerate(sums):
            got = tuple(sum(int(tc[i, gadgets + G * k + j]) << (8 * j) for j in range(N)) for k in ((7, 9) if double else (5, 7)))
            assert got == pt and on_curve(curve, got)
        chip = chips.record_weierstrass_chip(curve, double, 4)
        assert chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name]
        forged = evs.copy()
        forged["p_memory_records"][1, 3]["value"] ^= 1
        with pytest.raises(RuntimeError, match="result point"):
            oracle.tracegen_weierstrass(curve, double, forged)
    # a doubling's padding row is the point (0, 1) with the dummy write record on the first word of y (weierstrass_double.rs:225-239)
    pad = F.from_monty(oracle.tracegen_weierstrass(curve, True, dbls))[len(dbls)]
    y0 = 4 + 13 * (W // 2)
    assert [int(pad[y0 + k]) for k in (0, 4, 10)] == [1, 1, 1] and not pad[:y0].any()


@pytest.mark.parametrize("curve", ["Secp256r1", "Bls12381"])
def test_every_weierstrass_column_is_bound(oracle, curve):
    adds, _, dbls, _ = some_events(curve)
    for double, evs in ((False, adds), (True, dbls)):
        t = F.from_monty(oracle.tracegen_weierstrass(curve, double, evs))
        holes = windowed_sweep(chips.record_weierstrass_constraints(curve, double), chips.record_weierstrass_chip(curve, double, 4), t, (1, 3))
        assert holes == [], (curve, double, holes)


@pytest.mark.parametrize("curve", list(E.WEIERSTRASS_DECOMPRESS))
def test_decompress_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle, curve):
    evs, pts = some_decompressions(curve)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    c = E.WEIERSTRASS_CURVES[curve]
    N, W = c["n_limbs"], c["n_limbs"] // 4
    lexicographic = E.WEIERSTRASS_DECOMPRESS[curve]["lexicographic"]
    for (x, y), sign in zip(pts, (0, 1) * 3):
        assert on_curve(curve, (x, y)) and ((y > c["p"] - y) if lexicographic else (y & 1)) == sign
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_weierstrass_decompress(curve, evs, -1, counts)
    assert t.shape == (16, E.weierstrass_decompress_width(curve))
    per_gadget = 2 * (N // 2) + 2 * ((2 * N - 2) // 2)
    # six gadgets, the root's own range checks and AND, range_x and the root's FieldLtCols, two lookups per memory record (+ two more comparisons)
    assert counts.sum() == len(evs) * (6 * per_gadget + N // 2 + 1 + 2 + 2 * 2 * W + (2 if lexicographic else 0))
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_weierstrass_decompress_constraints(curve).b, tc) == []
    chip = chips.record_weierstrass_decompress_chip(curve, 4)
    assert chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name]
    # the padding rows hold the generator's x (weierstrass_decompress.rs:254-271)
    assert sum(int(tc[15, 5 + 9 * (i // 4) + i % 4]) << (8 * i) for i in range(N)) == c["generator"][0] and not tc[15, :5].any()
    wrong_root = evs.copy()
    for k in range(W):      # the other root: a valid point, but not the one the sign bit names
        wrong_root["y_memory_records"][0, k]["value"] = ((c["p"] - pts[0][1]) >> (32 * k)) & 0xffffffff
    with pytest.raises(RuntimeError, match="sign bit asks for"):
        oracle.tracegen_weierstrass_decompress(curve, wrong_root)
    forged = evs.copy()
    forged["y_memory_records"][1, 0]["value"] ^= 2
    with pytest.raises(RuntimeError, match="does not write a root"):
        oracle.tracegen_weierstrass_decompress(curve, forged)
    off_curve = evs.copy()
    x_bad = next(x for x in range(2, 50) if pow((x ** 3 + c["a"] * x + c["b"]) % c["p"], (c["p"] - 1) // 2, c["p"]) != 1)
    for k in range(W):
        off_curve["x_memory_records"][0, k]["value"] = (x_bad >> (32 * k)) & 0xffffffff
    with pytest.raises(RuntimeError, match="not on the curve"):
        oracle.tracegen_weierstrass_decompress(curve, off_curve)


@pytest.mark.parametrize("curve", ["Secp256r1", "Bls12381"])
def test_every_decompress_column_is_bound(oracle, curve):
    evs, _ = some_decompressions(curve)
    t = F.from_monty(oracle.tracegen_weierstrass_decompress(curve, evs))
    holes = windowed_sweep(chips.record_weierstrass_decompress_constraints(curve), chips.record_weierstrass_decompress_chip(curve, 4), t, (1, 2, 4))
    assert holes == [], (curve, holes)


def curve_machine():
    return M.run_machine(1200, seed=4, shard_cycles=1024, curve_calls={"Secp256k1": 3, "Bls12381": 2})


def test_machine_with_curve_calls_is_coherent(oracle):
    """A run that doubles and adds points of two curves with the precompiles: CPU shards, one precompile shard per syscall code
    (Secp256k1 add, Secp256k1 double, Bls12381 add, Bls12381 double), the memory shard. p ends at (2 k + 1) G."""
    m = curve_machine()
    kinds = [s.kind for s in m.shards]
    assert kinds[-5:] == ["precompile"] * 4 + ["memory"]
    recs = {s.record.weierstrass[0]: s.record.weierstrass[1] for s in m.shards if s.kind == "precompile"}
    assert {k: len(v) for k, v in recs.items()} == {"Secp256k1_add": 3, "Secp256k1_double": 1, "Bls12381_add": 2, "Bls12381_double": 1}
    for curve, calls in (("Secp256k1", 3), ("Bls12381", 2)):
        W = E.WEIERSTRASS_CURVES[curve]["n_limbs"] // 2
        last = [int(x) for x in recs[curve + "_add"][-1]["p_memory_records"]["value"]]
        want = multiples(curve, 2 * calls + 2)[2 * calls]       # (2 calls + 1) G
        assert last == [(want[k // (W // 2)] >> (32 * (k % (W // 2)))) & 0xffffffff for k in range(W)]
    shards = check_machine_airs(oracle, m)
    assert {c.name for cs in shards for c in cs} >= {"Secp256k1AddAssign", "Secp256k1DoubleAssign", "Bls12381AddAssign", "Bls12381DoubleAssign"}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


def decompress_machine():
    return M.run_machine(900, seed=6, shard_cycles=1024, decompress_calls={"Secp256k1": 3, "Secp256r1": 2, "Bls12381": 2})


def test_machine_with_decompress_calls_is_coherent(oracle):
    """A run that decompresses the generators with both sign bits: one precompile shard per curve; what ends up at ptr is the generator's y or
    its negative; every lookup balances and the global digests cancel."""
    m = decompress_machine()
    pre = [s.record.weierstrass_decompress for s in m.shards if s.kind == "precompile"]
    assert [(c, len(ev)) for c, ev in pre] == [("Secp256k1", 3), ("Secp256r1", 2), ("Bls12381", 2)]
    for curve, ev in pre:
        cv = E.WEIERSTRASS_CURVES[curve]
        ys = [sum(int(w) << (32 * k) for k, w in enumerate(e["y_memory_records"]["value"])) for e in ev]
        assert all(y in (cv["generator"][1], cv["p"] - cv["generator"][1]) for y in ys) and ys[0] != ys[1]
    shards = check_machine_airs(oracle, m)
    assert {c.name for cs in shards for c in cs} >= {"Secp256k1Decompress", "Secp256r1Decompress", "Bls12381Decompress"}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("curve", CURVES)
def test_gpu_weierstrass_tracegen_matches_oracle(hip_ctx, oracle, curve):
    """zkm_tracegen_weierstrass_add / _double against the restated generate_trace for every curve, bit for bit, with their byte lookups: the
    hand-made operations, one, none, 150 in a fixed table; a forged result and a coordinate that is not below the modulus are errors."""
    from ziren_amd import lib
    adds, _, dbls, _ = some_events(curve)
    pts = multiples(curve, 40)
    rng = np.random.default_rng(6)
    many_add = np.array([wevent(curve, False, pts[int(rng.integers(20, 40))], pts[int(rng.integers(0, 20))], clk=100 + 7 * i, seed=i)[0] for i in range(150)])
    many_dbl = np.array([wevent(curve, True, pts[int(rng.integers(0, 40))], clk=100 + 7 * i, seed=i)[0] for i in range(150)])
    for double, cases in ((False, ((adds, -1), (adds[:1], -1), (adds[:0], -1), (many_add, 8))), (True, ((dbls, -1), (dbls[:1], -1), (dbls[:0], -1), (many_dbl, 8)))):
        for ev, fixed in cases:
            counts = np.zeros((1 << 16, 10), dtype=np.uint32)
            want = oracle.tracegen_weierstrass(curve, double, ev, fixed, counts)
            blu = hip_ctx.byte_lookups()
            born = hip_ctx.tracegen_weierstrass(curve, double, ev, fixed, blu)
            mults = hip_ctx.tracegen_byte_mults(blu)
            assert (born.height, born.width) == want.shape
            got = born.to_host()
            assert np.array_equal(got, want), (curve, double, len(ev), np.argwhere(got != want)[:5])
            assert np.array_equal(F.from_monty(mults.to_host()), counts)
            born.free(); mults.free(); blu.free()
    forged = adds.copy()
    forged["p_memory_records"][1, 3]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="result point"):
        hip_ctx.tracegen_weierstrass(curve, False, forged)
    too_big = dbls[:1].copy()
    too_big["p_memory_records"]["prev_value"][0, :] = 0xffffffff
    with pytest.raises(lib.ZkmError, match="not below"):
        hip_ctx.tracegen_weierstrass(curve, True, too_big)


@pytest.mark.gpu
def test_gpu_machine_with_curve_calls_proves_and_verifies(hip_ctx, oracle):
    m = curve_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None


@pytest.mark.gpu
@pytest.mark.parametrize("curve", list(E.WEIERSTRASS_DECOMPRESS))
def test_gpu_decompress_tracegen_matches_oracle(hip_ctx, oracle, curve):
    """zkm_tracegen_weierstrass_decompress against the restated generate_trace, bit for bit, with the byte lookups: the hand-made calls, one,
    none, 150 in a fixed table; the other root, a forged word, an x off the curve and an x that is not below the modulus are errors."""
    from ziren_amd import lib
    evs, pts = some_decompressions(curve)
    mult = multiples(curve, 40)
    rng = np.random.default_rng(8)
    many = np.array([devent(curve, mult[int(rng.integers(0, 40))][0], int(rng.integers(0, 2)), clk=100 + 7 * i, seed=i)[0] for i in range(150)])
    for ev, fixed in ((evs, -1), (evs[:1], -1), (evs[:0], -1), (many, 8)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_weierstrass_decompress(curve, ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_weierstrass_decompress(curve, ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (curve, len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    c = E.WEIERSTRASS_CURVES[curve]
    W = c["n_limbs"] // 4
    wrong_root = evs.copy()
    for k in range(W):
        wrong_root["y_memory_records"][0, k]["value"] = ((c["p"] - pts[0][1]) >> (32 * k)) & 0xffffffff
    with pytest.raises(lib.ZkmError, match="sign bit asks for"):
        hip_ctx.tracegen_weierstrass_decompress(curve, wrong_root)
    forged = evs.copy()
    forged["y_memory_records"][1, 0]["value"] ^= 2
    with pytest.raises(lib.ZkmError, match="do not write a root"):
        hip_ctx.tracegen_weierstrass_decompress(curve, forged)
    off_curve = evs.copy()
    x_bad = next(x for x in range(2, 50) if pow((x ** 3 + c["a"] * x + c["b"]) % c["p"], (c["p"] - 1) // 2, c["p"]) != 1)
    for k in range(W):
        off_curve["x_memory_records"][0, k]["value"] = (x_bad >> (32 * k)) & 0xffffffff
    with pytest.raises(lib.ZkmError, match="not on the curve"):
        hip_ctx.tracegen_weierstrass_decompress(curve, off_curve)
    too_big = evs[:1].copy()
    too_big["x_memory_records"]["value"][0, :] = 0xffffffff
    with pytest.raises(lib.ZkmError, match="not below"):
        hip_ctx.tracegen_weierstrass_decompress(curve, too_big)


@pytest.mark.gpu
def test_gpu_machine_with_decompress_calls_proves_and_verifies(hip_ctx, oracle):
    m = decompress_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
