"""Recursion-machine chips: BaseAlu and ExtAlu recorded from the reference's `eval`
(ziren_amd/recursion.py), their traces as padded record streams, proved under the recursion provers' FRI configurations
(compress = log_blowup 1 / 84 queries, shrink = 2 / 42: crates/prover/src/lib.rs:192-196)."""
import numpy as np
import pytest

from ziren_amd import abi, air, chips, field as F, recursion as R


def traces(ext, n, seed=1, fixed=-1):
    ins, ev = R.synthetic_program(ext, n, seed)
    vw = R.EXT_VALUE_COLS if ext else R.BASE_VALUE_COLS
    return ins, ev, R.flat_trace(ins, R.ENTRIES_PER_ROW * R.ACCESS_COLS, fixed), R.flat_trace(ev, R.ENTRIES_PER_ROW * vw, fixed)


@pytest.mark.parametrize("ext", [False, True])
def test_constraints_hold(ext):
    rec = R.record_constraints(ext)
    for n in (0, 1, 63, 64, 1000):
        _, _, prep, main = traces(ext, n, seed=n + 1)
        assert air.debug_constraints(rec.b, F.from_monty(main), prep=F.from_monty(prep)) == []
    _, _, prep, main = traces(ext, 200)
    main = F.from_monty(main).copy()
    main[3, 5] = (int(main[3, 5]) + 1) % F.P
    assert {row for _, row in air.debug_constraints(rec.b, main, prep=F.from_monty(prep))} == {3}
    # lookups: per entry two receives and one send of kind Memory, address + 4-word block
    assert len(rec.sends) == 4 and len(rec.receives) == 8
    assert all(lk.kind == air.KIND_MEMORY and len(lk.values) == 5 for lk in rec.sends + rec.receives)


def test_row_counts():
    assert R.padded_rows(0) == 16 and R.padded_rows(64) == 16 and R.padded_rows(65) == 32 and R.padded_rows(10, 9) == 512
    with pytest.raises(ValueError):
        R.padded_rows(1000, 5)


def balanced_shard(n_base, n_ext, n_const, seed, heights=(-1, -1, -1), n_var=0, n_select=0, n_poseidon2=0, oracle=None, n_exp=0,
                   n_batch_fri=0, commit_public_values=False, wrap=False, n_fri_fold=0, inputs=None):
    """BaseAlu + ExtAlu + MemoryConst (+ MemoryVar + Select) over one consistent program: (chips with host traces, flat
    record streams (preprocessed words, main words))."""
    prog = R.balanced_program(n_base, n_ext, n_const, seed, n_var=n_var, n_select=n_select, n_poseidon2=n_poseidon2, n_exp=n_exp,
                              n_batch_fri=n_batch_fri, commit_public_values=commit_public_values, n_fri_fold=n_fri_fold, inputs=inputs)
    n_var = prog["n_var"]
    n_poseidon2 = prog["n_poseidon2"]
    specs = (("base_instrs", "base_events", R.BASE_VALUE_COLS, False), ("ext_instrs", "ext_events", R.EXT_VALUE_COLS, True))
    recs, streams = [], []
    for idx, (ik, ek, vw, ext) in enumerate(specs):
        prep = R.flat_trace(prog[ik], R.ENTRIES_PER_ROW * R.ACCESS_COLS, heights[idx])
        main = R.flat_trace(prog[ek], R.ENTRIES_PER_ROW * vw, heights[idx])
        rc = R.record_chip(ext, prep.shape[0].bit_length() - 1, prep_index=idx)
        rc.trace, rc.prep_trace = main, prep
        recs.append(rc)
        streams.append((prog[ik], prog[ek]))
    prep = R.flat_trace(prog["mem_entries"], R.CONST_MEM_ENTRIES_PER_ROW * R.CONST_MEM_ENTRY_COLS, heights[2], R.CONST_MEM_ENTRIES_PER_ROW)
    rc = R.record_mem_const(prep.shape[0].bit_length() - 1, prep_index=2)
    rc.trace, rc.prep_trace = np.zeros((prep.shape[0], 1), dtype=np.uint32), prep
    recs.append(rc)
    streams.append((prog["mem_entries"], None))
    if n_var:
        prep = R.flat_trace(prog["var_prep"], 2 * R.VAR_MEM_ENTRIES_PER_ROW, -1, R.VAR_MEM_ENTRIES_PER_ROW)
        rc = R.record_mem_var(prep.shape[0].bit_length() - 1, prep_index=3)
        rc.trace, rc.prep_trace = R.flat_trace(prog["var_values"], 4 * R.VAR_MEM_ENTRIES_PER_ROW, -1, R.VAR_MEM_ENTRIES_PER_ROW), prep
        recs.append(rc)
        streams.append((prog["var_prep"], prog["var_values"]))
    if n_select:
        prep = R.flat_trace(prog["select_prep"], R.SELECT_PREP_COLS, -1, 1)
        rc = R.record_select(prep.shape[0].bit_length() - 1, prep_index=4)
        rc.trace, rc.prep_trace = R.flat_trace(prog["select_events"], R.SELECT_COLS, -1, 1), prep
        recs.append(rc)
        streams.append((prog["select_prep"], prog["select_events"]))
    if n_poseidon2 and wrap:      # the wrap machine hashes with the skinny chip: eleven rows per permutation, degree 9
        prep = R.flat_trace(F.to_monty(R.poseidon2_skinny_prep(prog["poseidon2_instrs"])).reshape(-1), R.SKINNY_PREP_WIDTH, -1, 1)
        rc = R.record_poseidon2_skinny(prep.shape[0].bit_length() - 1, prep_index=5, degree=9)
        rc.trace, rc.prep_trace = oracle.tracegen_poseidon2_skinny(prog["poseidon2_events"], rc.log_height), prep
        recs.append(rc)
        streams.append((prep[:len(prog["poseidon2_instrs"]) * R.SKINNY_ROWS].reshape(-1), prog["poseidon2_events"]))
    elif n_poseidon2:
        prep = R.flat_trace(prog["poseidon2_prep"], R.POSEIDON2_WIDE_PREP_WIDTH, -1, 1)
        rc = R.record_poseidon2_wide(prep.shape[0].bit_length() - 1, prep_index=5)
        rc.trace, rc.prep_trace = oracle.tracegen_poseidon2_wide(prog["poseidon2_events"], rc.log_height), prep
        recs.append(rc)
        streams.append((prog["poseidon2_prep"], prog["poseidon2_events"]))
    if n_exp:
        prep = R.flat_trace(prog["exp_prep"], R.EXP_REVERSE_BITS_PREP_COLS, -1, 1)
        rc = R.record_exp_reverse_bits(prep.shape[0].bit_length() - 1, prep_index=6)
        rc.trace, rc.prep_trace = oracle.tracegen_exp_reverse_bits(prog["exp_bases"], prog["exp_bits"], prog["exp_offsets"], rc.log_height), prep
        assert np.array_equal(rc.trace, R.flat_trace(prog["exp_main"], R.EXP_REVERSE_BITS_COLS, rc.log_height, 1))
        recs.append(rc)
        streams.append((prog["exp_prep"], (prog["exp_bases"], prog["exp_bits"], prog["exp_offsets"])))
    if n_batch_fri:
        prep = R.flat_trace(prog["batch_fri_prep"], R.BATCH_FRI_PREP_COLS, -1, 1)
        rc = R.record_batch_fri(prep.shape[0].bit_length() - 1, prep_index=7, degree=9 if wrap else 3)
        rc.trace, rc.prep_trace = R.flat_trace(prog["batch_fri_main"], R.BATCH_FRI_COLS, rc.log_height, 1), prep
        recs.append(rc)
        streams.append((prog["batch_fri_prep"], prog["batch_fri_main"]))
    if n_fri_fold:
        prep = R.flat_trace(prog["fri_fold_prep"], R.FRI_FOLD_PREP_COLS, -1, 1)
        rc = R.record_fri_fold(prep.shape[0].bit_length() - 1, prep_index=9)
        rc.trace, rc.prep_trace = R.flat_trace(prog["fri_fold_main"], R.FRI_FOLD_COLS, rc.log_height, 1), prep
        recs.append(rc)
        streams.append((prog["fri_fold_prep"], prog["fri_fold_main"]))
    if commit_public_values:
        rc = R.record_public_values(prep_index=8)
        rc.prep_trace = R.flat_trace(prog["pv_prep"], R.PUBLIC_VALUES_PREP_COLS, R.PUBLIC_VALUES_LOG_HEIGHT, 1)
        rc.trace = R.flat_trace(prog["pv_main"], 1, R.PUBLIC_VALUES_LOG_HEIGHT, 1)
        recs.append(rc)
        streams.append((prog["pv_prep"], prog["pv_main"]))
        streams.append(prog["pv_digest"])
    return recs, streams


def recursion_public_values(digest):
    from ziren_amd import synth
    pv = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint64)
    pv[R.PV_DIGEST_POS:R.PV_DIGEST_POS + 8] = digest
    return F.to_monty(pv)


def compress_machine_shard(oracle, scale=1, seed=31):
    """All nine chips of the compress / shrink machine (crates/recursion/core/src/machine.rs:112-132) over one balanced program."""
    recs, streams = balanced_shard(400 * scale, 250 * scale, 40, seed=seed, n_var=120 * scale, n_select=100 * scale, n_poseidon2=30 * scale,
                                   oracle=oracle, n_exp=25 * scale, n_batch_fri=30 * scale, commit_public_values=True)
    digest = streams.pop()
    return recs, streams, digest


def all_chips_shard(oracle, scale=1, seed=35):
    """`machine_wide_with_all_chips` (machine.rs:68-87): the compress machine's nine chips plus FriFold."""
    recs, streams = balanced_shard(300 * scale, 250 * scale, 40, seed=seed, n_var=100 * scale, n_select=80 * scale, n_poseidon2=20 * scale,
                                   oracle=oracle, n_exp=20 * scale, n_batch_fri=25 * scale, commit_public_values=True, n_fri_fold=30 * scale)
    digest = streams.pop()
    for i, r in enumerate(recs):
        r.prep_index = i
    return recs, streams, digest


def test_fri_fold_and_all_chips_machine(oracle):
    """FriFold (in the reference only part of its all-chips test machines): constraints across the rows of an instruction, corrupted
    cells caught, and the ten-chip machine's lookups cancel; the oracle's proof verifies."""
    from ziren_amd import synth
    recs, streams, digest = all_chips_shard(oracle)
    assert [r.name for r in recs] == ["BaseAlu", "ExtAlu", "MemoryConst", "MemoryVar", "Select", "Poseidon2WideDeg3", "ExpReverseBitsLen", "BatchFRI",
                                      "FriFold", "PublicValues"]
    ff = recs[8]
    rec = R.record_fri_fold(constraints_only=True)
    main, prep = F.from_monty(ff.trace), F.from_monty(ff.prep_trace)
    assert air.debug_constraints(rec.b, main, prep=prep) == [] and len(rec.sends) == 9
    multi = int(np.nonzero(prep[1:, 0] == 0)[0][0]) + 1       # a row that continues an instruction
    for row, col, hit in ((multi, 8, {multi - 1, multi}), (multi, 2, {multi - 1, multi}), (0, 27, {0}), (0, 30, {0}), (0, 10, {0})):
        bad = main.copy()
        bad[row, col] = (int(bad[row, col]) + 1) % F.P
        got = {x for _, x in air.debug_constraints(rec.b, bad, prep=prep)}
        assert got and got <= hit | {row + 1}, (row, col, got)
    t = tally_of(recs)
    assert t and not any(t.values())
    fri = abi.FriConfig(2, 42, 16)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    opk = oracle.Pk([r.prep_trace for r in recs], [int(r.local_only) for r in recs], F.to_monty(0), igcs, 2)
    start = oracle.new_challenger()
    opk.observe_into(start)
    proof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], recursion_public_values(digest), fri, synth.NUM_PV_ELTS, start.copy())
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0


def wrap_machine_shard(oracle, scale=1, seed=33):
    """The eight chips of the wrap machine (machine.rs:138-153): the skinny hash chip and BatchFRI at DEGREE 9, no ExpReverseBitsLen."""
    recs, streams = balanced_shard(300 * scale, 200 * scale, 40, seed=seed, n_var=100 * scale, n_select=80 * scale, n_poseidon2=25 * scale,
                                   oracle=oracle, n_batch_fri=30 * scale, commit_public_values=True, wrap=True)
    digest = streams.pop()
    for i, r in enumerate(recs):
        r.prep_index = i
    return recs, streams, digest


def test_wrap_machine(oracle):
    """Poseidon2Skinny<9> completes the wrap machine: eleven rows per permutation whose output row carries the reference's
    permutation, constraints of degree up to 9 (quotient degree 8 for it and for BatchFRI<9>, 2 for the others, in one shard), lookups
    that cancel, and an oracle proof under the ultra-compressed KoalaBear FRI configuration (blow-up 8; the reference's wrap prover itself
    commits with a BN254 hasher, crates/prover/src/lib.rs:199 — not this path) that verifies."""
    import json
    import os
    from ziren_amd import synth
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon2_kat.json")))["vectors"]
    ev = np.array([list(v["input"]) + list(v["output"]) for v in kat], dtype=np.uint64)
    rows = F.from_monty(oracle.tracegen_poseidon2_skinny(F.to_monty(ev)))
    assert rows.shape == (64, 28) and all(np.array_equal(rows[11 * k + 10, :16], ev[k, 16:]) and np.array_equal(rows[11 * k, :16], ev[k, :16]) for k in range(len(kat)))
    assert not rows[33:].any()
    recs, streams, digest = wrap_machine_shard(oracle)
    assert [r.name for r in recs] == ["BaseAlu", "ExtAlu", "MemoryConst", "MemoryVar", "Select", "Poseidon2SkinnyDeg9", "BatchFRI", "PublicValues"]
    assert [r.log_quotient_degree for r in recs] == [1, 1, 1, 1, 1, 3, 3, 1]
    sk = recs[5]
    rp = R.record_poseidon2_skinny(constraints_only=True)
    main, prep = F.from_monty(sk.trace), F.from_monty(sk.prep_trace)
    assert air.debug_constraints(rp.b, main, prep=prep) == []
    for row, col, hit in ((0, 3, {0}), (3, 7, {2, 3}), (5, 20, {5}), (10, 2, {9})):      # input, an external round, an s0, the output state
        bad = main.copy()
        bad[row, col] = (int(bad[row, col]) + 1) % F.P
        assert {x for _, x in air.debug_constraints(rp.b, bad, prep=prep)} == hit, (row, col)
    t = tally_of(recs)
    assert t and not any(t.values())
    fri = abi.FriConfig(3, 28, 16)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    opk = oracle.Pk([r.prep_trace for r in recs], [int(r.local_only) for r in recs], F.to_monty(0), igcs, 3)
    start = oracle.new_challenger()
    opk.observe_into(start)
    proof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], recursion_public_values(digest), fri, synth.NUM_PV_ELTS, start.copy())
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0


def test_compress_machine_is_complete(oracle):
    """ExpReverseBitsLen, BatchFRI and PublicValues complete the compress machine's chip set: their constraints hold (the first two
    across consecutive rows), corrupted cells are caught, the nine chips' memory lookups cancel exactly, and the oracle's proof —
    with the committed digest as public values — verifies; a different claimed digest does not."""
    from ziren_amd import synth
    recs, streams, digest = compress_machine_shard(oracle)
    assert [r.name for r in recs] == ["BaseAlu", "ExtAlu", "MemoryConst", "MemoryVar", "Select", "Poseidon2WideDeg3", "ExpReverseBitsLen", "BatchFRI",
                                      "PublicValues"]
    pv = recursion_public_values(digest)
    by = {r.name: r for r in recs}
    for name, rec, cols in (("ExpReverseBitsLen", R.record_exp_reverse_bits(constraints_only=True), (0, 4, 5, 6)),
                            ("BatchFRI", R.record_batch_fri(constraints_only=True), (0, 5, 9, 12)),
                            ("PublicValues", R.record_public_values(constraints_only=True), (0,))):
        main, prep = F.from_monty(by[name].trace), F.from_monty(by[name].prep_trace)
        assert air.debug_constraints(rec.b, main, prep=prep, public_values=F.from_monty(pv)) == [], name
        for col in cols:
            bad = main.copy()
            bad[3, col] = (int(bad[3, col]) + 1) % F.P
            rows = {row for _, row in air.debug_constraints(rec.b, bad, prep=prep, public_values=F.from_monty(pv))}
            assert rows and rows <= {2, 3}, (name, col)
    assert by["PublicValues"].trace.shape == (16, 1) and by["ExpReverseBitsLen"].local_only is False
    t = tally_of(recs)
    assert t and not any(t.values())
    fri = abi.FriConfig(2, 42, 16)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    opk = oracle.Pk([r.prep_trace for r in recs], [int(r.local_only) for r in recs], F.to_monty(0), igcs, 2)
    start = oracle.new_challenger()
    opk.observe_into(start)
    for claimed, ok in ((digest, True), ([digest[0] + 1] + list(digest[1:]), False)):
        proof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], recursion_public_values(claimed), fri, synth.NUM_PV_ELTS, start.copy())
        assert (oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0) == ok


def tally_of(recs):
    tally = {}
    for r in recs:
        t, pt = F.from_monty(r.trace), F.from_monty(r.prep_trace)
        main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
        prep = {c: pt[:, c].astype(np.uint64) for c in range(pt.shape[1])}
        for sign, lks in ((1, r.sends), (-1, r.receives)):
            for lk in lks:
                vals = np.stack([np.broadcast_to(v.apply_np(prep, main), (t.shape[0],)) for v in lk.values], axis=1)
                mult = np.broadcast_to(lk.multiplicity.apply_np(prep, main), (t.shape[0],))
                for row in np.nonzero(mult)[0]:
                    key = tuple(int(x) for x in vals[row])
                    tally[key] = (tally.get(key, 0) + sign * int(mult[row])) % F.P
    return tally


def test_select_and_mem_var(oracle):
    """Select (out1 = bit ? in2 : in1, out2 the other) and MemoryVar (witnessed values) join the program: constraints hold,
    a wrong output is caught, and the five chips' memory lookups still cancel exactly; the oracle's proof verifies."""
    from ziren_amd import synth
    rs = R.record_select(constraints_only=True)
    recs, streams = balanced_shard(400, 250, 40, seed=13, n_var=90, n_select=150)
    assert [r.name for r in recs] == ["BaseAlu", "ExtAlu", "MemoryConst", "MemoryVar", "Select"]
    sel = recs[4]
    main, prep = F.from_monty(sel.trace), F.from_monty(sel.prep_trace)
    assert air.debug_constraints(rs.b, main, prep=prep) == [] and main[:150, 0].max() == 1 and set(main[:150, 0]) == {0, 1}
    bad = main.copy()
    bad[7, 1] = (int(bad[7, 1]) + 1) % F.P
    assert {row for _, row in air.debug_constraints(rs.b, bad, prep=prep)} == {7}
    assert len(rs.sends) == 2 and len(rs.receives) == 3 and len(R.record_mem_var(constraints_only=True).sends) == 2
    t = tally_of(recs)
    assert t and not any(t.values())
    assert any(tally_of(recs[:3] + recs[4:]).values())      # without MemoryVar the selects' bits are never written
    fri = abi.FriConfig(2, 42, 16)
    pv = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint32)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    opk = oracle.Pk([r.prep_trace for r in recs], [1] * 5, F.to_monty(0), igcs, 2)
    ch = oracle.new_challenger()
    opk.observe_into(ch)
    start = ch.copy()
    proof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], pv, fri, synth.NUM_PV_ELTS, ch)
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0


def test_mem_const_and_balance():
    """MemoryConst has no constraints of its own (constant.rs:152-160); over a consistent program every memory lookup
    sent is received: the multiset of (address, block) with signed multiplicities sums to zero."""
    rec = R.record_mem_const(constraints_only=True)
    assert len(rec.sends) == 2 and not rec.receives and rec.b.assemble()[2] == 0
    recs, _ = balanced_shard(700, 300, 50, seed=9)
    assert [r.trace.shape[0] for r in recs] == [256, 128, 128]
    tally = {}
    for r in recs:
        t, pt = F.from_monty(r.trace), F.from_monty(r.prep_trace)
        main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
        prep = {c: pt[:, c].astype(np.uint64) for c in range(pt.shape[1])}
        for sign, lks in ((1, r.sends), (-1, r.receives)):
            for lk in lks:
                vals = np.stack([np.broadcast_to(v.apply_np(prep, main), (t.shape[0],)) for v in lk.values], axis=1)
                mult = np.broadcast_to(lk.multiplicity.apply_np(prep, main), (t.shape[0],))
                for row in np.nonzero(mult)[0]:
                    key = tuple(int(x) for x in vals[row])
                    tally[key] = (tally.get(key, 0) + sign * int(mult[row])) % F.P
    assert tally and all(v == 0 for v in tally.values())


def test_oracle_proves_balanced_recursion_shard(oracle):
    """The three real chips alone make a shard the restated verifier accepts (cumulative sum zero); one wrong
    multiplicity in the constant-memory table and it is rejected."""
    from ziren_amd import synth
    fri = abi.FriConfig(2, 42, 16)
    pv = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint32)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    for tamper, want in ((False, True), (True, False)):
        recs, _ = balanced_shard(500, 300, 40, seed=5)
        if tamper:
            recs[2].prep_trace[0, 5] = F.to_monty((int(F.from_monty(recs[2].prep_trace[0, 5])) + 1) % F.P)
        opk = oracle.Pk([r.prep_trace for r in recs], [1, 1, 1], F.to_monty(0), igcs, 2)
        ch = oracle.new_challenger()
        opk.observe_into(ch)
        start = ch.copy()
        proof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], pv, fri, synth.NUM_PV_ELTS, ch)
        assert (oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0) == want


def test_poseidon2_wide(oracle):
    """Poseidon2Wide (degree 3): the rows' output columns are the reference permutation (golden vectors), all 298 constraints
    vanish on them and on the zero-state padding rows, a changed intermediate is caught, and with hashes in the program (their
    inputs read, their outputs consumed by later instructions) the six chips' lookups cancel."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "poseidon2_kat.json")))["vectors"]
    ev = np.array([list(v["input"]) + list(v["output"]) for v in kat], dtype=np.uint64)
    rows = F.from_monty(oracle.tracegen_poseidon2_wide(F.to_monty(ev)))
    assert rows.shape[1] == R.POSEIDON2_WIDE_WIDTH and np.array_equal(rows[:len(kat), 156:172], ev[:, 16:]) and np.array_equal(rows[:len(kat), :16], ev[:, :16])
    assert R.poseidon2_permute([int(x) for x in ev[0, :16]]) == [int(x) for x in ev[0, 16:]]
    rp = R.record_poseidon2_wide(constraints_only=True)
    assert rp.b.assemble()[2] == 298 and len(rp.sends) == 32 and not rp.receives
    prep = np.zeros((rows.shape[0], R.POSEIDON2_WIDE_PREP_WIDTH), dtype=np.uint64)
    assert air.debug_constraints(rp.b, rows, prep=prep) == []
    for col in (3, 130, 150, 160, 200, 305):    # an input lane, the state entering the internal rounds, an s0, an output lane, S-box columns
        bad = rows.copy()
        bad[1, col] = (int(bad[1, col]) + 1) % F.P
        assert {row for _, row in air.debug_constraints(rp.b, bad, prep=prep)} == {1}, col
    with pytest.raises(RuntimeError, match="not the permutation"):
        wrong = ev.copy()
        wrong[0, 20] += 1
        oracle.tracegen_poseidon2_wide(F.to_monty(wrong))
    recs, _ = balanced_shard(300, 150, 40, seed=21, n_var=60, n_select=80, n_poseidon2=40, oracle=oracle)
    assert recs[-1].name == "Poseidon2WideDeg3" and recs[-1].main_width == 313
    t = tally_of(recs)
    assert t and not any(t.values())


@pytest.mark.gpu
def test_gpu_poseidon2_wide_tracegen(hip_ctx, oracle):
    for n, fixed in ((0, -1), (1, -1), (17, -1), (3000, -1), (100, 9)):
        prog = R.balanced_program(10, 10, 40, seed=n + 1, n_var=20, n_poseidon2=n)
        ev = prog["poseidon2_events"]
        want = oracle.tracegen_poseidon2_wide(ev, fixed)
        m = hip_ctx.tracegen_poseidon2_wide(ev, fixed)
        assert (m.height, m.width) == want.shape and np.array_equal(m.to_host(), want), n
        m.free()


@pytest.mark.gpu
def test_gpu_flat_tracegen(hip_ctx):
    for ext in (False, True):
        for n, fixed in ((0, -1), (1, -1), (64, -1), (65, -1), (5000, 12)):
            ins, ev, prep, main = traces(ext, n, seed=n + 3, fixed=fixed)
            m = hip_ctx.tracegen_flat(ev, main.shape[1], fixed)
            assert (m.height, m.width) == main.shape and np.array_equal(m.to_host(), main)
            m.free()
            m = hip_ctx.tracegen_flat(ins, prep.shape[1], fixed)
            assert np.array_equal(m.to_host(), prep)
            m.free()
    from ziren_amd import lib
    with pytest.raises(lib.ZkmError, match="too small"):
        hip_ctx.tracegen_flat(np.zeros(12 * 100, dtype=np.uint32), 12, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("log_blowup,queries", [(1, 84), (2, 42), (3, 28)],
                         ids=["compress-1-84", "shrink-2-42", "ultra-compressed-3-28"])
def test_gpu_recursion_alu_shard(hip_ctx, oracle, log_blowup, queries):
    """The complete compress machine — BaseAlu, ExtAlu, MemoryConst, MemoryVar, Select, Poseidon2Wide, ExpReverseBitsLen, BatchFRI, PublicValues —
    over one consistent program under the FRI configurations the reference builds its recursion provers with
    (crates/prover/src/lib.rs:192-196): **compress** = `InnerSC::default()` = log_blowup 1 / 84 queries / 16 PoW bits
    (crates/stark/src/kb31_poseidon2.rs:203-213), **shrink** = `InnerSC::compressed()` = 2 / 42 (:215-227); (3, 28) is the
    ultra-compressed KoalaBear configuration (:229-241), which no prover of the reference is built with today. Device-built traces, preprocessed tables in the proving key,
    memory lookups balancing between the real chips, proof bit-identical to the oracle's and accepted by the
    restated verifier."""
    from ziren_amd import prover, synth
    recs, streams, digest = all_chips_shard(oracle, scale=8, seed=40)      # the compress machine's nine chips + FriFold
    fri = abi.FriConfig(log_blowup, queries, 16)
    pv = recursion_public_values(digest)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    hp = prover.HipProver(recs, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    hp.specialize_quotient_kernels(recs)
    preps = [hip_ctx.tracegen_flat(ins, r.prep_trace.shape[1], r.log_height) for (ins, _), r in zip(streams, recs)]
    for m, r in zip(preps, recs):
        assert np.array_equal(m.to_host(), r.prep_trace)
    lo = [int(r.local_only) for r in recs]
    pk = hp.setup(preps, lo, F.to_monty(0), igcs)
    opk = oracle.Pk([r.prep_trace for r in recs], lo, F.to_monty(0), igcs, log_blowup)
    assert np.array_equal(pk.commit, opk.commitment())
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    born = [hip_ctx.tracegen_poseidon2_wide(ev, r.log_height) if r.name == "Poseidon2WideDeg3" else
            hip_ctx.tracegen_exp_reverse_bits(*ev, r.log_height) if r.name == "ExpReverseBitsLen" else
            hip_ctx.tracegen_flat(ev, r.trace.shape[1], r.log_height) if ev is not None else hip_ctx.upload(r.trace)
            for (_, ev), r in zip(streams, recs)]
    for m, r in zip(born, recs):
        assert np.array_equal(m.to_host(), r.trace), r.name
    proof = hp.prove_shard(pk, pv, born, ch).copy()
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], pv, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    for m in born:
        m.free()


@pytest.mark.gpu
def test_gpu_wrap_chips_under_ultra_compressed_koalabear_config(hip_ctx, oracle):
    """The eight chips of the wrap machine (Poseidon2Skinny, BatchFRI at DEGREE 9; crates/recursion/core/src/machine.rs:138-153) on the device
    under the ultra-compressed *KoalaBear* configuration (log_blowup 3, 28 queries; kb31_poseidon2.rs:229-241): quotient degrees 2 and 8
    in one shard, the skinny hash chip's eleven-row permutations built on the device, proof bit-identical to the oracle's. This is NOT the
    reference's wrap prover: that one commits under OuterSC — a BN254 Poseidon2 Merkle tree, log_blowup 4
    (crates/prover/src/lib.rs:199, crates/recursion/core/src/stark/config.rs:70-83) — a different hasher, outside this path; what is
    exercised here is that chip set's constraints (degree 9, quotient degree 8) through the KoalaBear commit + open."""
    from ziren_amd import prover, synth
    for n in (0, 1, 5, 700):
        prog = R.balanced_program(10, 10, 40, seed=n + 1, n_var=20, n_poseidon2=n)
        want = oracle.tracegen_poseidon2_skinny(prog["poseidon2_events"])
        m = hip_ctx.tracegen_poseidon2_skinny(prog["poseidon2_events"])
        assert (m.height, m.width) == want.shape and np.array_equal(m.to_host(), want), n
        m.free()
    recs, streams, digest = wrap_machine_shard(oracle, scale=6, seed=44)
    fri = abi.FriConfig(3, 28, 16)
    pv = recursion_public_values(digest)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    hp = prover.HipProver(recs, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    hp.specialize_quotient_kernels(recs)
    preps = [hip_ctx.tracegen_flat(ins, r.prep_trace.shape[1], r.log_height) for (ins, _), r in zip(streams, recs)]
    for m, r in zip(preps, recs):
        assert np.array_equal(m.to_host(), r.prep_trace), r.name
    lo = [int(r.local_only) for r in recs]
    pk = hp.setup(preps, lo, F.to_monty(0), igcs)
    opk = oracle.Pk([r.prep_trace for r in recs], lo, F.to_monty(0), igcs, 3)
    assert np.array_equal(pk.commit, opk.commitment())
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    born = [hip_ctx.tracegen_poseidon2_skinny(ev, r.log_height) if r.name.startswith("Poseidon2Skinny") else
            hip_ctx.tracegen_flat(ev, r.trace.shape[1], r.log_height) if ev is not None else hip_ctx.upload(r.trace)
            for (_, ev), r in zip(streams, recs)]
    for m, r in zip(born, recs):
        assert np.array_equal(m.to_host(), r.trace), r.name
    proof = hp.prove_shard(pk, pv, born, ch).copy()
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], pv, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    for m in born:
        m.free()
