"""Recursion-machine chips: BaseAlu and ExtAlu recorded from the reference's `eval`
(ziren_amd/recursion.py), their traces as padded record streams, proved with the compress FRI configuration."""
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


def mirror(rec):
    """Memory chips on the other side of the ALU chips' lookups (mem/constant.rs, mem/variable.rs), mirrored."""
    t, pt = F.from_monty(rec.trace), F.from_monty(rec.prep_trace)
    main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
    prep = {c: pt[:, c].astype(np.uint64) for c in range(pt.shape[1])}
    cols, sends, receives = [], [], []
    for src, dst in ((rec.sends, receives), (rec.receives, sends)):
        for lk in src:
            first = len(cols)
            for v in list(lk.values) + [lk.multiplicity]:
                cols.append(v.apply_np(prep, main))
            vals = [air.VirtualPairCol.single_main(first + j) for j in range(len(lk.values))]
            dst.append(air.Lookup(vals, air.VirtualPairCol.single_main(first + len(lk.values)), lk.kind))
    width = len(cols)
    b = air.AirBuilder(width, 0, air.local_permutation_trace_width(len(sends) + len(receives), 2))
    air.eval_permutation_constraints(b, sends, receives, 2, False)
    program = b.assemble()
    trace = np.empty((t.shape[0], width), dtype=np.uint32)
    for c in range(width):
        trace[:, c] = F.to_monty(cols[c])
    return chips.RecordedChip(name=rec.name + "Memory", log_height=rec.log_height, main_width=width, sends=sends, receives=receives,
                              program=program, lookups_blob=air.encode_lookups(sends, receives), num_constraints=int(program[2]),
                              trace=trace)


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
@pytest.mark.parametrize("log_blowup,queries", [(2, 42), (3, 28)])
def test_gpu_recursion_alu_shard(hip_ctx, oracle, log_blowup, queries):
    """BaseAlu + ExtAlu (+ mirrored memory chips) under the compress / shrink FRI configurations
    (crates/stark/src/kb31_poseidon2.rs:215-241): device-built traces, preprocessed tables in the proving key,
    proof bit-identical to the oracle's and accepted by the restated verifier."""
    from ziren_amd import prover, synth
    recs, streams = [], []
    for idx, (ext, lh, n) in enumerate(((False, 10, 3500), (True, 9, 2000))):
        ins, ev, prep, main = traces(ext, n, seed=40 + idx, fixed=lh)
        rc = R.record_chip(ext, lh, prep_index=idx)
        rc.trace, rc.prep_trace = main, prep
        recs.append(rc)
        streams.append((ins, ev, prep.shape[1], main.shape[1], lh))
    all_chips = recs + [mirror(r) for r in recs]
    fri = abi.FriConfig(log_blowup, queries, 16)
    pv = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint32)
    igcs = F.to_monty(F.SplitMix64(4).uniform_field(14))
    hp = prover.HipProver(all_chips, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    hp.specialize_quotient_kernels(recs)
    pk = hp.setup([hip_ctx.tracegen_flat(ins, pw, lh) for ins, _, pw, _, lh in streams], [1, 1], F.to_monty(0), igcs)
    opk = oracle.Pk([r.prep_trace for r in recs], [1, 1], F.to_monty(0), igcs, log_blowup)
    assert np.array_equal(pk.commit, opk.commitment())
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    born = [hip_ctx.tracegen_flat(ev, mw, lh) for _, ev, _, mw, lh in streams]
    born += [hip_ctx.upload(c.trace) for c in all_chips[2:]]
    proof = hp.prove_shard(pk, pv, born, ch).copy()
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, all_chips, [c.trace for c in all_chips], pv, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(opk, all_chips, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    for m in born:
        m.free()
