"""BooleanCircuitGarble (crates/core/machine/src/syscall/precompiles/boolean_circuit_garble/): the ciphertext checks of a garbled circuit's non-free
gates, 1 + num_gates rows per call with constraints between consecutive rows. Pinned by the reference's cost (588: width 292, 72 lookups)
and by the executor's own arithmetic (XORs and comparisons of words). Transcribed as written: a call with exactly one gate cannot satisfy the
reference's AIR (is_last_gate * is_first_gate = 0, air.rs:58), so the tests use two gates or more."""
import json
import os
import random

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine


def garble_rows(gates, delta, corrupt=(), shard=2, clk=300, input_ptr=0xc00000, output_ptr=0xc10000, seed=0):
    """The row records of BOOLEAN_CIRCUIT_GARBLE(input_ptr, output_ptr) (syscalls/precompiles/boolean_circuit/garble.rs:10-95): gates = (type, h0, h1,
    label_b) with four words each; `corrupt`: the gates whose expected ciphertext is wrong. Returns the records and the value written."""
    rng = np.random.default_rng(seed)
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    rows = np.zeros(1 + len(gates), dtype=E.GARBLE_ROW)
    head = rows[0]
    head["shard"], head["clk"], head["input_address"], head["output_address"], head["gates_num"], head["delta"] = shard, clk, input_ptr, output_ptr, len(gates), delta
    head["reads"][0] = (len(gates), shard, clk) + prev()
    for k in range(4):
        head["reads"][1 + k] = (delta[k], shard, clk) + prev()
    running = True
    for g, (t, h0, h1, lb) in enumerate(gates):
        want = [h0[i] ^ h1[i] ^ lb[i] ^ (delta[i] if t else 0) for i in range(4)]
        if g in corrupt:
            want[g % 4] ^= 0x100
        words = [t] + list(h0) + list(h1) + list(lb) + want
        r = rows[1 + g]
        r["shard"], r["clk"], r["input_address"], r["output_address"], r["gates_num"], r["delta"] = shard, clk, input_ptr + 20 + 68 * g, output_ptr, len(gates), delta
        r["is_gate"], r["gate_id"], r["pre_check"] = 1, g, int(running)
        for k in range(17):
            r["reads"][k] = (words[k], shard, clk) + prev()
        running = running and E.garble_gate_ok(words, delta)
        if g == len(gates) - 1:
            r["write"] = (int(running), shard, clk, int(rng.integers(0, 1 << 32))) + prev()
    return rows, int(running)


def some_calls(seed=3, sizes=(5, 3, 2)):
    rnd = random.Random(seed)
    w4 = lambda: [rnd.randrange(1 << 32) for _ in range(4)]      # noqa: E731
    gates = [(rnd.choice([0, E.GARBLE_OR_GATE]), w4(), w4(), w4()) for _ in range(max(sizes))]
    calls = [garble_rows(gates[:n], w4(), corrupt=(1,) if j == 1 else (), clk=300 + 10 * j, input_ptr=0xc00000 + 0x1000 * j, seed=j) for j, n in enumerate(sizes)]
    return np.concatenate([c[0] for c in calls]), [c[1] for c in calls]


def test_garble_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    rows, results = some_calls()
    assert results == [1, 0, 1]
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_boolean_circuit_garble(rows, -1, counts)
    assert t.shape == (16, E.GARBLE_WIDTH)
    n_gates, n_calls = 10, 3
    assert counts.sum() == n_gates * (48 + 2 * 17) + n_calls * (2 * 5 + 2)      # a gate: twelve XorOperations, seventeen reads; a call: five reads, one write
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_boolean_circuit_garble_constraints().b, tc) == []
    assert [int(tc[i, 291]) for i in (5, 9, 12)] == results      # checks[3] on each call's last gate
    chip = chips.record_boolean_circuit_garble_chip(4)
    assert not chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name] == 588
    for change, why in ((("write", "value", 5, 0), "does not write the result"), (("pre_check", None, 9, 1), "does not continue"), (("gate_id", None, 3, 3), "does not continue"),
                        (("input_address", None, 7, 0xc01000), "does not continue")):
        forged = rows.copy()
        field, sub, at, value = change
        if sub:
            forged[field][sub][at] = value
        else:
            forged[field][at] = value
        with pytest.raises(RuntimeError, match=why):
            oracle.tracegen_boolean_circuit_garble(forged)
    with pytest.raises(RuntimeError, match="cut short"):
        oracle.tracegen_boolean_circuit_garble(rows[:-1])


def test_every_garble_column_is_bound(oracle):
    rows, _ = some_calls()
    t = F.from_monty(oracle.tracegen_boolean_circuit_garble(rows))
    # a gate in the middle of a call, the gate whose ciphertext is wrong, a last gate
    holes = windowed_sweep(chips.record_boolean_circuit_garble_constraints(), chips.record_boolean_circuit_garble_chip(4), t, (3, 8, 9))
    # what the reference's AIR leaves free on a gate row: the IsZero inverses of a zero difference (the words are equal), as in every IsZeroOperation
    free = {c for c in range(244, 288) if (c - 244) % 11 in (0, 2, 4, 6)}
    assert [h for h in holes if h not in free] == [], holes


def garble_machine():
    return M.run_machine(1500, seed=12, shard_cycles=2048, garble_calls=(3, -4, 2))


def test_machine_with_garble_calls_is_coherent(oracle):
    m = garble_machine()
    pre = [s.record.garble for s in m.shards if s.kind == "precompile"]
    assert len(pre) == 1 and len(pre[0]) == 3 + 3 + 4 + 2
    assert [int(r["write"]["value"]) for r in pre[0] if r["is_gate"] and r["gate_id"] + 1 == r["gates_num"]] == [1, 0, 1]
    shards = check_machine_airs(oracle, m)
    assert "BooleanCircuitGarble" in {c.name for cs in shards for c in cs}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_garble_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_boolean_circuit_garble against the restated generate_trace, bit for bit, with the byte lookups: three calls, one, none, forty
    calls of up to thirty gates in a fixed table; broken chains, a wrong result and a cut call are errors."""
    from ziren_amd import lib
    rows, _ = some_calls()
    many, _ = some_calls(seed=5, sizes=tuple(2 + (7 * j) % 29 for j in range(40)))
    for ev, fixed in ((rows, -1), (rows[:6], -1), (rows[:0], -1), (many, 10)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_boolean_circuit_garble(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_boolean_circuit_garble(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    for field, sub, at, value, why in (("write", "value", 5, 0, "does not write the result"), ("pre_check", None, 9, 1, "does not continue"),
                                       ("gate_id", None, 3, 3, "does not continue")):
        forged = rows.copy()
        if sub:
            forged[field][sub][at] = value
        else:
            forged[field][at] = value
        with pytest.raises(lib.ZkmError, match=why):
            hip_ctx.tracegen_boolean_circuit_garble(forged)
    with pytest.raises(lib.ZkmError, match="cut short"):
        hip_ctx.tracegen_boolean_circuit_garble(rows[:-1])


@pytest.mark.gpu
def test_gpu_machine_with_garble_calls_proves_and_verifies(hip_ctx, oracle):
    m = garble_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
