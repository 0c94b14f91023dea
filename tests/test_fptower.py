"""The field-tower precompiles (crates/core/machine/src/syscall/precompiles/fptower/fp.rs, fp2_addsub.rs, fp2_mul.rs): FpOpAssign, Fp2AddSubAssign and
Fp2MulAssign over the base fields of Bn254 and Bls12381 — six chips from one parameterised implementation per layer, on the same FieldOpCols
gadget as the curve chips; FpOp and Fp2AddSub choose their operation per row with flags (FieldOpCols::eval_variable). Pinned by the
reference's cost table (704 / 1382 / 2885 and 1048 / 2070 / 4341 — the reference spells the last two chips "Bls12831...") and by Python
integers for every result."""
import json
import os
import random

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine

CASES = [(field, kind) for field in ("Bn254", "Bls12381") for kind in ("fp", "fp2_addsub", "fp2_mul")]
OPS = {"fp": [E.FIELD_OP_ADD, E.FIELD_OP_MUL, E.FIELD_OP_SUB], "fp2_addsub": [E.FIELD_OP_ADD, E.FIELD_OP_SUB], "fp2_mul": [E.FIELD_OP_MUL]}


def fp_event(field, kind, op, x, y, shard=2, clk=300, x_ptr=0x900000, y_ptr=0x900200, seed=0):
    """The flattened FpOpEvent / Fp2AddSubEvent / Fp2MulEvent of the call (syscalls/precompiles/fptower/): y read at clk, the result written over
    x at clk + 1."""
    rng = np.random.default_rng(seed)
    per = E.WEIERSTRASS_CURVES[field]["n_limbs"] // 4
    e = np.zeros(1, dtype=E.fp_tower_event_dtype(field, kind))[0]
    e["shard"], e["clk"], e["x_ptr"], e["y_ptr"] = shard, clk, x_ptr, y_ptr
    if kind != "fp2_mul":
        e["op"] = op
    r = E.fp_tower_result(field, kind, op, x, y)
    flat = lambda v: [(c >> (32 * i)) & 0xffffffff for c in (v if isinstance(v, tuple) else (v,)) for i in range(per)]      # noqa: E731
    xw, yw, rw = flat(x), flat(y), flat(r)
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    for k in range(len(xw)):
        e["x_memory_records"][k] = (rw[k], shard, clk + 1, xw[k]) + prev()
        e["y_memory_records"][k] = (yw[k], shard, clk) + prev()
    return e, r


def some_events(field, kind, n=6, seed=1):
    """Random operands, plus the corners: p - 1 with p - 1, zero with anything, equal operands (a - a = 0)."""
    P = E.WEIERSTRASS_CURVES[field]["p"]
    rnd = random.Random(seed)
    el = (lambda: rnd.randrange(P)) if kind == "fp" else (lambda: (rnd.randrange(P), rnd.randrange(P)))
    corner = (lambda v: v) if kind == "fp" else (lambda v: (v, v))
    pairs = [(el(), el()) for _ in range(n)] + [(corner(P - 1), corner(P - 1)), (corner(0), el())]
    same = el()
    pairs.append((same, same))
    evs = [fp_event(field, kind, OPS[kind][i % len(OPS[kind])], x, y, clk=300 + 10 * i, seed=i) for i, (x, y) in enumerate(pairs)]
    return np.array([e for e, _ in evs]), [r for _, r in evs]


@pytest.mark.parametrize("field,kind", CASES)
def test_fp_tower_rows_satisfy_the_airs_and_cost_what_the_reference_says(oracle, field, kind):
    evs, results = some_events(field, kind)
    c = E.WEIERSTRASS_CURVES[field]
    N = c["n_limbs"]
    per, words, gadgets = N // 4, (N // 4 if kind == "fp" else N // 2), {"fp": 1, "fp2_addsub": 2, "fp2_mul": 6}[kind]
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_fp_tower(field, kind, evs, -1, counts)       # refuses events whose x records do not hold the result
    assert t.shape == (16, E.fp_tower_width(field, kind))
    assert counts.sum() == len(evs) * (gadgets * (N + 2 * (N - 1)) + 4 * words)
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_fp_tower_constraints(field, kind).b, tc) == []
    head = {"fp": 8, "fp2_addsub": 6, "fp2_mul": 5}[kind]
    G, g0 = 6 * N - 4, head + 22 * words
    outs = {"fp": [0], "fp2_addsub": [0, 1], "fp2_mul": [4, 5]}[kind]
    for i, r in enumerate(results):
        got = tuple(sum(int(tc[i, g0 + G * k + j]) << (8 * j) for j in range(N)) for k in outs)
        assert got == (r if isinstance(r, tuple) else (r,)), i
    chip = chips.record_fp_tower_chip(field, kind, 4)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    assert chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name]
    if kind != "fp2_mul":      # padding rows claim an addition of zeros
        assert int(tc[len(evs), 3]) == 1 and int(tc[len(evs), 0]) == 0
    forged = evs.copy()
    forged["x_memory_records"][1, 0]["value"] ^= 1
    with pytest.raises(RuntimeError, match="does not write the result"):
        oracle.tracegen_fp_tower(field, kind, forged)


@pytest.mark.parametrize("field,kind", [("Bn254", "fp"), ("Bn254", "fp2_addsub"), ("Bls12381", "fp2_mul")])
def test_every_fp_tower_column_is_bound(oracle, field, kind):
    evs, _ = some_events(field, kind)
    t = F.from_monty(oracle.tracegen_fp_tower(field, kind, evs))
    holes = windowed_sweep(chips.record_fp_tower_constraints(field, kind), chips.record_fp_tower_chip(field, kind, 4), t, (1, 2, 3, 6))
    assert holes == [], (field, kind, holes)


def fp_machine():
    return M.run_machine(1200, seed=6, shard_cycles=1024, fp_calls={"Bn254": 7, "Bls12381": 6})


def test_machine_with_field_tower_calls_is_coherent(oracle):
    """A run that calls all six syscalls of each field: the three Fp codes of a field land in one precompile shard (their events are filed under
    FP_ADD: syscalls/precompiles/fptower/fp.rs:83-120), so do FP2_ADD and FP2_SUB; FP2_MUL has its own."""
    m = fp_machine()
    kinds = [s.kind for s in m.shards]
    assert kinds[-7:] == ["precompile"] * 6 + ["memory"]
    recs = {s.record.fp_tower[0]: s.record.fp_tower[1] for s in m.shards if s.kind == "precompile"}
    assert {k: len(v) for k, v in recs.items()} == {"Bn254_fp": 4, "Bn254_fp2_addsub": 2, "Bn254_fp2_mul": 1, "Bls12381_fp": 3, "Bls12381_fp2_addsub": 2,
                                                    "Bls12381_fp2_mul": 1}
    assert sorted(int(x) for x in recs["Bn254_fp"]["op"]) == [0, 0, 1, 2]
    shards = check_machine_airs(oracle, m)
    assert {c.name for cs in shards for c in cs} >= {"Bn254FpOpAssign", "Bn254Fp2AddSubAssign", "Bn254Fp2MulAssign", "Bls12381FpOpAssign",
                                                     "Bls12831Fp2AddSubAssign", "Bls12831Fp2MulAssign"}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("field,kind", CASES)
def test_gpu_fp_tower_tracegen_matches_oracle(hip_ctx, oracle, field, kind):
    from ziren_amd import lib
    evs, _ = some_events(field, kind)
    many, _ = some_events(field, kind, n=180, seed=7)
    for ev, fixed in ((evs, -1), (evs[:1], -1), (evs[:0], -1), (many, 8)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_fp_tower(field, kind, ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_fp_tower(field, kind, ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (field, kind, len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    forged = evs.copy()
    forged["x_memory_records"][1, 0]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="not the result"):
        hip_ctx.tracegen_fp_tower(field, kind, forged)
    with pytest.raises(lib.ZkmError, match="ZKM_CURVE_BN254 or ZKM_CURVE_BLS12381"):
        h = __import__("ctypes").c_void_p()
        lib.check(lib.load().zkm_tracegen_fp_op(hip_ctx.h, 0, None, 0, -1, None, __import__("ctypes").byref(h)))


@pytest.mark.gpu
def test_gpu_machine_with_field_tower_calls_proves_and_verifies(hip_ctx, oracle):
    m = fp_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
