"""The EdAddAssign precompile (crates/core/machine/src/syscall/precompiles/edwards/ed_add.rs): Ed25519 point addition, the operation a
signature verification (BASELINE config 5: the tendermint light client) spends its precompile rows on. One row per addition, 1861 columns:
eight big-field gadgets (operations/field/: two FieldInnerProductCols, four FieldOpCols, two FieldDenCols) over 32 byte limbs. Everything
is in the reference tree; pinned by its cost table (3637 per row: 1861 columns, 881 lookups, degree 3) and by three independent
computations of the same sums: Python integers (events.ed25519_add), the oracle's byte-vector arithmetic with binary long division and the
binary extended Euclid (oracle/bigfield.hpp), and the device's 32-bit limbs with Barrett reduction and Fermat inversion (csrc/bigfield.cuh)."""
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


def ed_machine():
    return M.run_machine(1500, seed=3, shard_cycles=1024, ed_calls=6)


def test_machine_with_ed_add_calls_is_coherent(oracle):
    """A run that adds 2B to B six times with the precompile: CPU shards, the EdAddAssign precompile shard, the memory shard; constraints,
    lookups and global digests as for the other precompiles, and the point left in memory is 13 B."""
    m = ed_machine()
    assert [s.kind for s in m.shards][-2:] == ["precompile", "memory"]
    ev = m.shards[-2].record.ed_add
    assert len(ev) == 6
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
    assert {c.name for c in shards[-2]} == {"SyscallPrecompile", "EdAddAssign", "MemoryLocal", "Global", "Byte", "Program"}
    d = global_digests(shards)
    assert oracle.global_digest_sum(d + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_ed_add_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_ed_add against the restated generate_trace, bit for bit, with its byte lookups: the hand-made additions (neutral element,
    doubling, a point and its negative), a run's additions, one, none, 300 random multiples in a fixed table; a forged sum is an error."""
    from ziren_amd import lib
    evs, _ = some_additions()
    run = ed_machine().shards[-2].record.ed_add
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
def test_gpu_machine_with_ed_add_calls_proves_and_verifies(hip_ctx, oracle):
    m = ed_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
