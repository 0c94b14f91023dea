"""The Ed25519 precompiles (crates/core/machine/src/syscall/precompiles/edwards/): EdAddAssign — point addition, the operation a signature
verification (BASELINE config 5: the tendermint light client) spends its precompile rows on; one row per addition, 1861 columns: eight
big-field gadgets (operations/field/: two FieldInnerProductCols, four FieldOpCols, two FieldDenCols) over 32 byte limbs — and EdDecompress
— x from y and a sign bit, once per public key and per signature; 1566 columns: a less-than gadget, six field operations including a
division, a square root. Everything is in the reference tree; pinned by its cost table (3637 and 3062 per row: column counts, lookup
counts, degree 3) and by three independent computations of the same numbers: Python integers (events.ed25519_add / _decompress), the
oracle's byte-vector arithmetic with binary long division and the binary extended Euclid (oracle/bigfield.hpp), and the device's 32-bit
limbs with Barrett reduction, Fermat inversion and the (p + 3) / 8 power for the root (csrc/bigfield.cuh)."""
import json
import os

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine

P, D = E.ED25519_P, E.ED25519_D
BASE = (15112221349535400772501151409588531511454012693041857206046113283949847762202, 4 * pow(5, P - 2, P) % P)      # RFC 8032 section 5.1
GADGETS = 5 + 16 * 13 + 16 * 9


def on_curve(pt):
    x, y = pt
    return (-x * x + y * y - 1 - D * x * x * y * y) % P == 0


def words(v):
    return [(v >> (32 * i)) & 0xffffffff for i in range(8)]


def ed_event(p, q, shard=2, clk=300, p_ptr=0x600000, q_ptr=0x600100, seed=0):
    """The EllipticCurveAddEvent of ED_ADD(p_ptr, q_ptr) (create_ec_add_event, events/precompiles/ec.rs:96-139): q read at clk, p + q
    written over p at clk + 1."""
    rng = np.random.default_rng(seed)
    e = np.zeros(1, dtype=E.ED_ADD_EVENT)[0]
    e["shard"], e["clk"], e["p_ptr"], e["q_ptr"] = shard, clk, p_ptr, q_ptr
    r = E.ed25519_add(p, q)
    pw, qw, rw = words(p[0]) + words(p[1]), words(q[0]) + words(q[1]), words(r[0]) + words(r[1])
    for k in range(16):
        for records, rec in (("q_memory_records", (qw[k], shard, clk)), ("p_memory_records", (rw[k], shard, clk + 1, pw[k]))):
            prev = (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))
            e[records][k] = rec + prev
    return e, r


def some_additions():
    """A chain B, 2B, 3B + ..., an addition of the neutral element, a doubling, and a point added to its negative."""
    evs, sums = [], []
    pt, two = BASE, E.ed25519_add(BASE, BASE)
    for i in range(5):
        e, r = ed_event(pt, two if i % 2 else pt, clk=300 + 10 * i, seed=i)
        evs.append(e); sums.append(r)
        pt = r
    for i, (p, q) in enumerate((((0, 1), BASE), (two, two), (BASE, ((P - BASE[0]) % P, BASE[1])))):
        e, r = ed_event(p, q, clk=900 + 10 * i, seed=10 + i)
        evs.append(e); sums.append(r)
    return np.array(evs), sums


def test_ed25519_reference_arithmetic():
    assert on_curve(BASE) and BASE[1] == 46316835694926478169428394003475163141307993866256225615783033603165251855960
    two = E.ed25519_add(BASE, BASE)
    assert on_curve(two) and E.ed25519_add(two, BASE) == E.ed25519_add(BASE, two)
    assert E.ed25519_add(BASE, (0, 1)) == BASE and E.ed25519_add(BASE, ((P - BASE[0]) % P, BASE[1])) == (0, 1)
    # the group order annihilates the base point: l * B = neutral, by double-and-add through the same formula
    order = (1 << 252) + 27742317777372353535851937790883648493
    acc, addend = (0, 1), BASE
    for k in range(order.bit_length()):
        if (order >> k) & 1:
            acc = E.ed25519_add(acc, addend)
        addend = E.ed25519_add(addend, addend)
    assert acc == (0, 1)


def test_ed_add_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    evs, sums = some_additions()
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_ed_add(evs, -1, counts)          # the oracle refuses events whose p records do not hold p + q
    assert t.shape == (16, E.ED_ADD_WIDTH)
    assert counts.sum() == len(evs) * (8 * 94 + 32 * 2)   # per gadget 16 + 16 + 31 + 31 byte-pair range checks; two per memory access
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_ed_add_constraints().b, tc) == []
    for i, (x3, y3) in enumerate(sums):                   # the results the last two gadgets hold are Python's
        got = tuple(sum(int(tc[i, GADGETS + g * 188 + k]) << (8 * k) for k in range(32)) for g in (6, 7))
        assert got == (x3, y3) and on_curve(got), i
    # padding rows: the gadgets of the zero inputs — zero everywhere except witness_high = 2^14 >> 8
    pad = tc[len(evs)]
    assert not pad[:GADGETS].any() and set(int(x) for x in pad[GADGETS + 64 + 62:GADGETS + 188]) == {64} and not pad[GADGETS:GADGETS + 64 + 62].any()
    chip = chips.record_ed_add_chip(4)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    assert len(chip.sends) + len(chip.receives) == 881 and chip.local_only
    assert chip.main_width + 4 * chip.perm_ext_width + 8 == ref["EdAddAssign"] == 3637
    forged = evs.copy()
    forged["p_memory_records"][2, 9]["value"] ^= 1
    with pytest.raises(RuntimeError, match="p \\+ q"):
        oracle.tracegen_ed_add(forged)


ED_ADD_FREE = {}


def test_every_ed_add_column_is_bound(oracle):
    evs, _ = some_additions()
    t = F.from_monty(oracle.tracegen_ed_add(evs))
    holes = windowed_sweep(chips.record_ed_add_constraints(), chips.record_ed_add_chip(4), t, (1, 4, 6))
    assert [c for c in holes if c not in ED_ADD_FREE] == [], holes


def test_ed_decompress_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    evs, xs = some_decompressions()
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_ed_decompress(evs, -1, counts)      # refuses events whose x records do not hold the root their sign selects
    assert t.shape == (16, E.ED_DECOMPRESS_WIDTH)
    # per row: seven gadgets x 94, the root's 16 range checks + AND + LTU, y's LTU, sixteen memory accesses x 2
    assert counts.sum() == len(evs) * (7 * 94 + 18 + 1 + 32)
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_ed_decompress_constraints().b, tc) == []
    for i, x in enumerate(xs):
        root = sum(int(tc[i, 1155 + k]) << (8 * k) for k in range(32))
        neg = sum(int(tc[i, 1378 + k]) << (8 * k) for k in range(32))
        assert root % 2 == 0 and x in (root, neg) and (root + neg) % P == 0 and int(tc[i, 1377]) == 0
    chip = chips.record_ed_decompress_chip(4)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    assert len(chip.sends) + len(chip.receives) == 742 and chip.main_width + 4 * chip.perm_ext_width + 8 == ref["EdDecompress"] == 3062
    assert E.ed25519_decompress(2, 0) is None               # y = 2 is not on the curve
    forged = evs.copy()
    forged["x_memory_records"][1, 0]["value"] ^= 1
    with pytest.raises(RuntimeError, match="does not write x"):
        oracle.tracegen_ed_decompress(forged)
    off_curve = evs[:1].copy()
    off_curve["y_memory_records"][0]["value"] = words(2)
    with pytest.raises(RuntimeError, match="not a square"):
        oracle.tracegen_ed_decompress(off_curve)


ED_DECOMPRESS_FREE = {}


def test_every_ed_decompress_column_is_bound(oracle):
    evs, _ = some_decompressions()
    t = F.from_monty(oracle.tracegen_ed_decompress(evs))
    holes = windowed_sweep(chips.record_ed_decompress_constraints(), chips.record_ed_decompress_chip(4), t, (1, 2, 14))
    assert [c for c in holes if c not in ED_DECOMPRESS_FREE] == [], holes


def ed_machine():
    return M.run_machine(1500, seed=3, shard_cycles=1024, ed_calls=6)


def precompile_record(m, name):
    return [s.record for s in m.shards if s.kind == "precompile" and len(getattr(s.record, name))][0]


def dec_event(y, sign, shard=2, clk=300, ptr=0x700000, seed=0):
    """The EdDecompressEvent of ED_DECOMPRESS(ptr, sign) (syscalls/precompiles/edwards/decompress.rs:33-83)."""
    rng = np.random.default_rng(seed)
    e = np.zeros(1, dtype=E.ED_DECOMPRESS_EVENT)[0]
    e["shard"], e["clk"], e["ptr"], e["sign"] = shard, clk, ptr, sign
    x = E.ed25519_decompress(y, sign)
    for k in range(8):
        prev = (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))
        e["y_memory_records"][k] = (words(y)[k], shard, clk) + prev
        prev = (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))
        e["x_memory_records"][k] = (words(x)[k], shard, clk, int(rng.integers(0, 1 << 32))) + prev
    return e, x


def some_decompressions():
    """B .. 7B with the sign of their x, each also with the other sign, and the neutral element (y = 1, x = 0)."""
    pts, evs, xs = [BASE], [], []
    for _ in range(6):
        pts.append(E.ed25519_add(pts[-1], BASE))
    for i, (x, y) in enumerate(pts):
        for flip in (0, 1):
            e, got = dec_event(y, (x & 1) ^ flip, clk=300 + 20 * i + flip, seed=2 * i + flip)
            assert got == (x if not flip else P - x)
            evs.append(e); xs.append(got)
    e, got = dec_event(1, 0, clk=900)
    assert got == 0
    return np.array(evs + [e]), xs + [got]


def test_machine_with_ed_add_calls_is_coherent(oracle):
    """A run that adds 2B to B six times with the precompile: CPU shards, the EdAddAssign precompile shard, the memory shard; constraints,
    lookups and global digests as for the other precompiles, and the point left in memory is 13 B."""
    m = ed_machine()
    assert [s.kind for s in m.shards][-3:] == ["precompile", "precompile", "memory"]
    ev = precompile_record(m, "ed_add").ed_add
    dec = precompile_record(m, "ed_decompress").ed_decompress
    assert len(ev) == 6 and len(dec) == 2          # B and 2B are decompressed from their y and sign first, as a verifier gets A and R
    two = E.ed25519_add(BASE, BASE)
    assert [int(x) for x in dec["x_memory_records"]["value"][0]] == words(BASE[0]) and [int(x) for x in dec["x_memory_records"]["value"][1]] == words(two[0])
    pt = BASE
    for _ in range(6):
        pt = E.ed25519_add(pt, E.ed25519_add(BASE, BASE))
    last = [int(x) for x in ev[-1]["p_memory_records"]["value"]]
    assert last == words(pt[0]) + words(pt[1])
    thirteen = (0, 1)
    for _ in range(13):
        thirteen = E.ed25519_add(thirteen, BASE)
    assert pt == thirteen
    shards = check_machine_airs(oracle, m)
    names = [{c.name for c in cs} for cs in shards]
    assert {"SyscallPrecompile", "EdAddAssign", "MemoryLocal", "Global", "Byte", "Program"} in names
    assert {"SyscallPrecompile", "EdDecompress", "MemoryLocal", "Global", "Byte", "Program"} in names
    d = global_digests(shards)
    assert oracle.global_digest_sum(d + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_ed_add_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_ed_add against the restated generate_trace, bit for bit, with its byte lookups: the hand-made additions (neutral element,
    doubling, a point and its negative), a run's additions, one, none, 300 random multiples in a fixed table; a forged sum is an error."""
    from ziren_amd import lib
    evs, _ = some_additions()
    run = precompile_record(ed_machine(), "ed_add").ed_add
    rng = np.random.default_rng(5)
    pts = [BASE]
    for _ in range(40):
        pts.append(E.ed25519_add(pts[-1], pts[int(rng.integers(0, len(pts)))]))
    many = np.array([ed_event(pts[int(rng.integers(0, 41))], pts[int(rng.integers(0, 41))], clk=100 + 7 * i, seed=i)[0] for i in range(300)])
    for ev, fixed in ((evs, -1), (run, -1), (evs[:1], -1), (evs[:0], -1), (many, 9)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_ed_add(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_ed_add(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    forged = evs.copy()
    forged["p_memory_records"][2, 9]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="p \\+ q"):
        hip_ctx.tracegen_ed_add(forged)


@pytest.mark.gpu
def test_gpu_ed_decompress_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_ed_decompress against the restated generate_trace, bit for bit, with its byte lookups; a y that is no point's, a y not
    below p and a forged x are errors."""
    from ziren_amd import lib
    evs, _ = some_decompressions()
    run = precompile_record(ed_machine(), "ed_decompress").ed_decompress
    pts = [BASE]
    for _ in range(60):
        pts.append(E.ed25519_add(pts[-1], BASE))
    many = np.array([dec_event(y, x & 1, clk=100 + 3 * i, seed=i)[0] for i, (x, y) in enumerate(pts * 4)])
    for ev, fixed in ((evs, -1), (run, -1), (evs[:1], -1), (evs[:0], -1), (many, 8)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_ed_decompress(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_ed_decompress(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    forged = evs.copy()
    forged["x_memory_records"][1, 0]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="not the x"):
        hip_ctx.tracegen_ed_decompress(forged)
    off_curve = evs[:1].copy()
    off_curve["y_memory_records"][0]["value"] = words(2)
    with pytest.raises(lib.ZkmError, match="not a square"):
        hip_ctx.tracegen_ed_decompress(off_curve)
    too_big = evs[:1].copy()
    too_big["y_memory_records"][0]["value"] = words(P + 1)
    with pytest.raises(lib.ZkmError, match="not below"):
        hip_ctx.tracegen_ed_decompress(too_big)


@pytest.mark.gpu
def test_gpu_machine_with_ed_add_calls_proves_and_verifies(hip_ctx, oracle):
    m = ed_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None


@pytest.mark.gpu
def test_gpu_range_check_histogram_at_scale(hip_ctx):
    """The U8Range lookups of the big-field tables are counted from the finished columns by a pass of its own (tracegen::u8_pair_histogram, many
    slabs of rows, two halves of the key space, a reduction). At 2^14 + 100 rows — seventeen slabs, the last one ragged — its counters must
    equal a numpy recount over the downloaded trace: every gadget's columns in pairs, real rows only, plus the one kind of U8Range lookup the
    row kernel records itself (the 8-bit limb of each memory record's timestamp difference, checked next to a zero)."""
    n = (1 << 14) + 100
    pts = [BASE]
    for _ in range(63):
        pts.append(E.ed25519_add(pts[-1], BASE))
    rng = np.random.default_rng(4)
    distinct = np.array([ed_event(pts[int(rng.integers(0, 64))], pts[int(rng.integers(0, 64))], clk=100 + 3 * i, seed=i)[0] for i in range(512)])
    evs = np.tile(distinct, n // 512 + 1)[:n]
    blu = hip_ctx.byte_lookups()
    born = hip_ctx.tracegen_ed_add(evs, 15, blu)
    mults = F.from_monty(hip_ctx.tracegen_byte_mults(blu).to_host())
    t = F.from_monty(born.to_host())[:n]
    assert born.height == 1 << 15
    want = np.zeros(1 << 16, dtype=np.int64)
    g = t[:, GADGETS:GADGETS + 8 * 188].astype(np.int64)
    np.add.at(want, (g[:, 0::2] << 8 | g[:, 1::2]).ravel(), 1)                       # the eight gadgets: 94 pairs each, all of even length
    # the other U8Range lookups of a row: the 8-bit limb of every memory record's timestamp difference, checked next to a zero
    for k in range(16):
        np.add.at(want, t[:, 5 + 13 * k + 12].astype(np.int64), 1)                   # p: MemoryWriteCols, diff_8bit_limb (the key is 0 << 8 | limb)
        np.add.at(want, t[:, 5 + 16 * 13 + 9 * k + 8].astype(np.int64), 1)           # q: MemoryReadCols
    got = mults[:, 4].astype(np.int64)                                               # ByteOpcode::U8Range is column 4
    assert got.sum() == n * (8 * 94 + 32) and np.array_equal(got, want)
    born.free(); blu.free()
