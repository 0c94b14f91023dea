"""The two chips over U256Field. U256XU2048Mul (crates/core/machine/src/syscall/precompiles/u256x2048_mul/air.rs): a 256-bit by 2048-bit product as
eight gadgets chained through their carries (cost 5849 = the reference's). Uint256MulMod (crates/core/machine/src/syscall/precompiles/uint256/air.rs): x <- x * y mod m over 256-bit integers, a zero modulus standing for
2^256 — one FieldOpCols over U256Field whose modulus polynomial is read from memory (63 witness limbs, because 2^256 has 33). Pinned by the
reference's cost table (880: width 480, 98 lookups, degree 3), by Python-integer arithmetic and by the completeness sweep."""
import json
import os
import random

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine

BIG = (1 << 256) - 189


def uevent(x, y, m, shard=2, clk=300, x_ptr=0xa00000, y_ptr=0xa00100, seed=0):
    """The flattened Uint256MulEvent of UINT256_MUL(x_ptr, y_ptr) (syscalls/precompiles/uint256.rs:14-97): y and the modulus read at clk, the
    product written over x at clk + 1."""
    rng = np.random.default_rng(seed)
    e = np.zeros(1, dtype=E.UINT256_MUL_EVENT)[0]
    e["shard"], e["clk"], e["x_ptr"], e["y_ptr"] = shard, clk, x_ptr, y_ptr
    r = E.uint256_mulmod(x, y, m)
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    w = lambda v, k: (v >> (32 * k)) & 0xffffffff      # noqa: E731
    for k in range(8):
        e["x_memory_records"][k] = (w(r, k), shard, clk + 1, w(x, k)) + prev()
        e["y_memory_records"][k] = (w(y, k), shard, clk) + prev()
        e["modulus_memory_records"][k] = (w(m, k), shard, clk) + prev()
    return e, r


def some_events(seed=1):
    """Random operands under a large modulus and under none, and the corners: tiny values, zeros, the largest operands, a power of two."""
    rnd = random.Random(seed)
    cases = [(rnd.randrange(BIG), rnd.randrange(BIG), BIG), (rnd.randrange(1 << 256), rnd.randrange(1 << 256), 0), (2, 2, 3), (0, 0, 0), (BIG - 1, BIG - 1, BIG),
             ((1 << 256) - 1, (1 << 256) - 1, 0), (rnd.randrange(1 << 128), rnd.randrange(1 << 128), 1 << 255), (rnd.randrange(1 << 200), 0, 12345), (7, 1, 7)]
    evs = [uevent(*c, clk=300 + 10 * i, seed=i) for i, c in enumerate(cases)]
    return np.array([e for e, _ in evs]), [r for _, r in evs], cases


def test_uint256_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    evs, results, cases = some_events()
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_uint256_mul(evs, -1, counts)
    assert t.shape == (16, E.UINT256_MUL_WIDTH)
    with_modulus = sum(1 for c in cases if c[2])
    # result, carry (16 pairs each), two witness vectors of 63 (32 lookups each), two lookups per memory record, one comparison where there is a modulus
    assert counts.sum() == len(evs) * (16 + 16 + 32 + 32 + 2 * 24) + with_modulus
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_uint256_mul_constraints().b, tc) == []
    for i, r in enumerate(results):
        assert sum(int(tc[i, 255 + j]) << (8 * j) for j in range(32)) == r == cases[i][0] * cases[i][1] % (cases[i][2] or 1 << 256)
        assert int(tc[i, 253]) == (cases[i][2] == 0) and int(tc[i, 254]) == (cases[i][2] != 0)
    chip = chips.record_uint256_mul_chip(4)
    assert chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name] == 880
    forged = evs.copy()
    forged["x_memory_records"][0, 2]["value"] ^= 4
    with pytest.raises(RuntimeError, match="does not write"):
        oracle.tracegen_uint256_mul(forged)
    overflow = uevent((1 << 255) + 5, (1 << 255) + 9, 3)[0]      # x * y / 3 has 509 bits
    with pytest.raises(RuntimeError, match="does not fit"):
        oracle.tracegen_uint256_mul(np.array([overflow]))


def test_every_uint256_column_is_bound(oracle):
    evs, _, _ = some_events()
    t = F.from_monty(oracle.tracegen_uint256_mul(evs))
    holes = windowed_sweep(chips.record_uint256_mul_constraints(), chips.record_uint256_mul_chip(4), t, (1, 4, 6))      # with, without a modulus, a power of two
    assert holes == [], holes


def uint256_machine():
    return M.run_machine(700, seed=9, shard_cycles=1024, uint256_calls=5)


def test_machine_with_uint256_calls_is_coherent(oracle):
    m = uint256_machine()
    pre = [s.record.uint256_mul for s in m.shards if s.kind == "precompile"]
    assert len(pre) == 1 and len(pre[0]) == 5
    as_int = lambda words: sum(int(w) << (32 * k) for k, w in enumerate(words))      # noqa: E731
    x = as_int(pre[0][0]["x_memory_records"]["prev_value"])
    for j, e in enumerate(pre[0]):      # every call continues from the last one's result; the modulus alternates
        modulus = as_int(e["modulus_memory_records"]["value"])
        assert modulus == (0 if j & 1 else BIG) and as_int(e["x_memory_records"]["prev_value"]) == x
        x = x * as_int(e["y_memory_records"]["value"]) % (modulus or 1 << 256)
        assert as_int(e["x_memory_records"]["value"]) == x
    shards = check_machine_airs(oracle, m)
    assert "Uint256MulMod" in {c.name for cs in shards for c in cs}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


def xevent(a, b, shard=2, clk=300, a_ptr=0xb00000, b_ptr=0xb00100, lo_ptr=0xb00400, hi_ptr=0xb00800, seed=0):
    """The flattened U256xU2048MulEvent of U256XU2048_MUL(a_ptr, b_ptr) with $a2 = lo_ptr, $a3 = hi_ptr (syscalls/precompiles/u256x2048_mul.rs:20-93)."""
    rng = np.random.default_rng(seed)
    e = np.zeros(1, dtype=E.U256X2048_MUL_EVENT)[0]
    e["shard"], e["clk"], e["a_ptr"], e["b_ptr"], e["lo_ptr"], e["hi_ptr"] = shard, clk, a_ptr, b_ptr, lo_ptr, hi_ptr
    prod = a * b
    lo, hi = prod % (1 << 2048), prod >> 2048
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    w = lambda v, k: (v >> (32 * k)) & 0xffffffff      # noqa: E731
    e["lo_ptr_memory"] = (lo_ptr, shard, clk) + prev()
    e["hi_ptr_memory"] = (hi_ptr, shard, clk) + prev()
    for k in range(8):
        e["a_memory_records"][k] = (w(a, k), shard, clk) + prev()
        e["hi_memory_records"][k] = (w(hi, k), shard, clk + 1, int(rng.integers(0, 1 << 32))) + prev()
    for k in range(64):
        e["b_memory_records"][k] = (w(b, k), shard, clk) + prev()
        e["lo_memory_records"][k] = (w(lo, k), shard, clk + 1, int(rng.integers(0, 1 << 32))) + prev()
    return e, (lo, hi)


def some_products(seed=2):
    rnd = random.Random(seed)
    cases = [(rnd.randrange(1 << 256), rnd.randrange(1 << 2048)), ((1 << 256) - 1, (1 << 2048) - 1), (0, rnd.randrange(1 << 2048)), (1, 1),
             (rnd.randrange(1 << 256), 1 << 2047), (rnd.randrange(1 << 64), rnd.randrange(1 << 300))]
    return np.array([xevent(*c, clk=300 + 10 * i, seed=i)[0] for i, c in enumerate(cases)]), cases


def test_u256x2048_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    evs, cases = some_products()
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_u256x2048_mul(evs, -1, counts)
    assert t.shape == (16, E.U256X2048_MUL_WIDTH)
    assert counts.sum() == len(evs) * (8 * 96 + 2 * (2 + 8 + 64 + 64 + 8))      # 96 range checks per gadget, two lookups per memory record
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_u256x2048_mul_constraints().b, tc) == []
    for i, (a, b) in enumerate(cases):      # the results are the low 2048 bits, the last carry the high 256
        lo = sum(int(tc[i, 1608 + 190 * (j // 32) + j % 32]) << (8 * j) for j in range(256))
        hi = sum(int(tc[i, 1608 + 190 * 7 + 32 + j]) << (8 * j) for j in range(32))
        assert (hi << 2048) + lo == a * b
    chip = chips.record_u256x2048_mul_chip(4)
    assert not chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name] == 5849
    forged = evs.copy()
    forged["lo_memory_records"][0, 40]["value"] ^= 1
    with pytest.raises(RuntimeError, match="does not write the product"):
        oracle.tracegen_u256x2048_mul(forged)
    moved = evs.copy()
    moved["hi_ptr"][1] += 4
    with pytest.raises(RuntimeError, match="registers hold"):
        oracle.tracegen_u256x2048_mul(moved)


def test_every_u256x2048_column_is_bound(oracle):
    evs, _ = some_products()
    t = F.from_monty(oracle.tracegen_u256x2048_mul(evs))
    holes = windowed_sweep(chips.record_u256x2048_mul_constraints(), chips.record_u256x2048_mul_chip(4), t, (1, 4))
    assert holes == [], holes


def u2048_machine():
    return M.run_machine(900, seed=10, shard_cycles=1024, u2048_calls=2, uint256_calls=2)


def test_machine_with_u256x2048_calls_is_coherent(oracle):
    m = u2048_machine()
    pre = {("uint256" if len(s.record.uint256_mul) else "u2048"): s.record for s in m.shards if s.kind == "precompile"}
    assert len(pre["uint256"].uint256_mul) == 2 and len(pre["u2048"].u256x2048_mul) == 2
    as_int = lambda words: sum(int(w) << (32 * k) for k, w in enumerate(words))      # noqa: E731
    for e in pre["u2048"].u256x2048_mul:
        prod = as_int(e["a_memory_records"]["value"]) * as_int(e["b_memory_records"]["value"])
        assert (as_int(e["hi_memory_records"]["value"]) << 2048) + as_int(e["lo_memory_records"]["value"]) == prod
        assert (int(e["lo_ptr_memory"]["value"]), int(e["hi_ptr_memory"]["value"])) == (int(e["lo_ptr"]), int(e["hi_ptr"])) == (0x006d0400, 0x006d0800)
    shards = check_machine_airs(oracle, m)
    assert {c.name for cs in shards for c in cs} >= {"Uint256MulMod", "U256XU2048Mul"}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_uint256_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_uint256_mul against the restated generate_trace, bit for bit, with the byte lookups: the hand-made calls, one, none, 300 random
    ones (moduli of every size) in a fixed table; a forged result and a quotient that does not fit are errors."""
    from ziren_amd import lib
    evs, _, _ = some_events()
    rnd = random.Random(5)
    many = []
    for i in range(300):
        bits = rnd.choice([1, 8, 64, 200, 255, 256])
        m = rnd.choice([0, rnd.randrange(1, 1 << bits)])
        y = rnd.randrange(m) if m else rnd.randrange(1 << 256)
        many.append(uevent(rnd.randrange(1 << 256), y, m, clk=100 + 7 * i, seed=i)[0])
    many = np.array(many)
    for ev, fixed in ((evs, -1), (evs[:1], -1), (evs[:0], -1), (many, 9)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_uint256_mul(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_uint256_mul(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    forged = evs.copy()
    forged["x_memory_records"][0, 2]["value"] ^= 4
    with pytest.raises(lib.ZkmError, match=r"are not x \* y mod modulus"):
        hip_ctx.tracegen_uint256_mul(forged)
    overflow = uevent((1 << 255) + 5, (1 << 255) + 9, 3)[0]
    with pytest.raises(lib.ZkmError, match="does not fit 256 bits"):
        hip_ctx.tracegen_uint256_mul(np.array([overflow]))


@pytest.mark.gpu
def test_gpu_machine_with_uint256_calls_proves_and_verifies(hip_ctx, oracle):
    m = uint256_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None


@pytest.mark.gpu
def test_gpu_u256x2048_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_u256x2048_mul against the restated generate_trace, bit for bit, with the byte lookups: the hand-made products, one, none, 100
    random ones in a fixed table; a forged word and a pointer that is not the register's value are errors."""
    from ziren_amd import lib
    evs, _ = some_products()
    rnd = random.Random(7)
    many = np.array([xevent(rnd.randrange(1 << 256), rnd.randrange(1 << 2048), clk=100 + 7 * i, seed=i)[0] for i in range(100)])
    for ev, fixed in ((evs, -1), (evs[:1], -1), (evs[:0], -1), (many, 7)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_u256x2048_mul(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_u256x2048_mul(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    forged = evs.copy()
    forged["lo_memory_records"][0, 40]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match=r"are not a \* b"):
        hip_ctx.tracegen_u256x2048_mul(forged)
    moved = evs.copy()
    moved["hi_ptr"][1] += 4
    with pytest.raises(lib.ZkmError, match="register records hold"):
        hip_ctx.tracegen_u256x2048_mul(moved)


@pytest.mark.gpu
def test_gpu_machine_with_u256x2048_calls_proves_and_verifies(hip_ctx, oracle):
    m = u2048_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None


@pytest.mark.gpu
def test_gpu_u256x2048_range_check_histogram_at_scale(hip_ctx):
    """The histogram pass with strided, odd-length segments (eight gadgets 190 columns apart; 63-limb witnesses whose last limb is checked next to
    a zero) over several slabs with a ragged end: its U8Range counters equal a numpy recount of the downloaded trace."""
    n = (1 << 12) + 37
    rnd = random.Random(13)
    distinct = np.array([xevent(rnd.randrange(1 << 256), rnd.randrange(1 << 2048), clk=100 + 7 * i, seed=i)[0] for i in range(64)])
    evs = np.tile(distinct, n // 64 + 1)[:n]
    blu = hip_ctx.byte_lookups()
    born = hip_ctx.tracegen_u256x2048_mul(evs, 13, blu)
    mults = F.from_monty(hip_ctx.tracegen_byte_mults(blu).to_host())
    t = F.from_monty(born.to_host())[:n].astype(np.int64)
    want = np.zeros(1 << 16, dtype=np.int64)
    for g in range(8):
        base = 1608 + 190 * g
        both = t[:, base:base + 64]                                                  # result and carry: 32 + 32 limbs
        np.add.at(want, (both[:, 0::2] << 8 | both[:, 1::2]).ravel(), 1)
        for w0 in (base + 64, base + 127):                                           # witness_low, witness_high: 63 limbs each
            w = t[:, w0:w0 + 63]
            np.add.at(want, (w[:, 0:62:2] << 8 | w[:, 1:62:2]).ravel(), 1)
            np.add.at(want, (w[:, 62] << 8).ravel(), 1)
    for base, count, stride, limb in ((6, 2, 9, 8), (24, 8, 9, 8), (96, 64, 9, 8), (672, 64, 13, 12), (1504, 8, 13, 12)):      # the memory records' 8-bit limbs
        for k in range(count):
            np.add.at(want, t[:, base + stride * k + limb], 1)
    got = mults[:, 4].astype(np.int64)
    assert got.sum() == n * (8 * 96 + 146) and np.array_equal(got, want)
    born.free(); blu.free()
