"""The ALU chips' AIRs, transcribed from the reference's `eval` (ziren_amd/chips.py), against the generated traces:
every constraint vanishes on every row (events, padding, the wrap-around row), a corrupted cell is caught, and the
lookups the chips issue have the reference's shape. GPU: a shard of the five real chips, traces born on the device, is
proved and accepted by the restated verifier, and the proof is bit-identical to the oracle's."""
import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F
from test_tracegen import golden_events


def rows_for(oracle, chip, n, seed=5, fixed=-1):
    ev = E.synthetic_alu_events(chip, n, seed=seed)
    return ev, F.from_monty(oracle.tracegen_alu(chip, ev, fixed))


@pytest.mark.parametrize("chip", sorted(E.CHIP_NAMES))
def test_constraints_hold_on_generated_rows(oracle, chip):
    rec = chips.record_constraints(chip)
    _, t = rows_for(oracle, chip, 3000)     # 3000 events + 1096 padding rows
    assert air.debug_constraints(rec.b, t) == []
    _, t = rows_for(oracle, chip, 0)        # padding only
    assert air.debug_constraints(rec.b, t) == []
    _, t = rows_for(oracle, chip, 64)       # no padding
    assert air.debug_constraints(rec.b, t) == []


def test_jump_constraints_hold(oracle):
    rec = chips.record_jump_constraints()
    for n in (0, 64, 3000):
        t = F.from_monty(oracle.tracegen_jump(E.synthetic_jump_events(n, seed=n + 1)))
        assert air.debug_constraints(rec.b, t) == []
    t = t.copy()
    t[5, 23] ^= 1            # a bit of next_next_pc's range checker
    assert {row for _, row in air.debug_constraints(rec.b, t)} == {5}
    assert [lk.kind for lk in rec.sends] == [air.KIND_INSTRUCTION] and len(rec.receives) == 1


def test_branch_constraints_hold(oracle):
    rec = chips.record_branch_constraints()
    for n in (0, 64, 3000):
        t = F.from_monty(oracle.tracegen_branch(E.synthetic_branch_events(n, seed=n + 1)))
        assert air.debug_constraints(rec.b, t) == []
    t = t.copy()
    t[5, 59] ^= 1            # is_branching
    assert {row for _, row in air.debug_constraints(rec.b, t)} == {5}
    kinds = [lk.kind for lk in rec.sends]
    assert kinds.count(air.KIND_INSTRUCTION) == 3 and kinds.count(air.KIND_BYTE) == 4 and len(rec.receives) == 1


def test_recorded_costs_match_the_reference():
    """Chip::cost (crates/stark/src/chip.rs:152-163) = preprocessed + main + 4 * permutation + 4 * quotient columns of every
    recorded chip equals the figure the reference pins in mips_costs.json (its core_air_cost_consistency test): the column
    layouts, the number of lookups (permutation width) and the constraint degree (quotient width) all enter."""
    import json
    import os
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    recs = [chips.record_chip(c, 10) for c in sorted(E.CHIP_NAMES)] + [
        chips.record_cpu_chip(10), chips.record_program_chip(10), chips.record_mul_chip(10), chips.record_divrem_chip(10),
        chips.record_branch_chip(10), chips.record_jump_chip(10), chips.record_mov_cond_chip(10), chips.record_memory_instrs_chip(10),
        chips.record_memory_local_chip(10), chips.record_syscall_instrs_chip(10), chips.record_misc_instrs_chip(10), chips.record_byte_chip(),
        chips.record_global_chip(10), chips.record_memory_global_chip(False, 10), chips.record_memory_global_chip(True, 10),
        chips.record_syscall_table_chip(False, 10), chips.record_syscall_table_chip(True, 10), chips.record_poseidon2_permute_chip(10),
        chips.record_keccak_sponge_chip(10), chips.record_sha_extend_chip(10), chips.record_sha_compress_chip(10),
        chips.record_ed_add_chip(10), chips.record_ed_decompress_chip(10), chips.record_uint256_mul_chip(10), chips.record_u256x2048_mul_chip(10), chips.record_boolean_circuit_garble_chip(10), chips.record_sys_linux_chip(10)] + [
        chips.record_weierstrass_chip(curve, double, 10) for curve in E.WEIERSTRASS_CURVES for double in (False, True)] + [
        chips.record_weierstrass_decompress_chip(curve, 10) for curve in E.WEIERSTRASS_DECOMPRESS] + [
        chips.record_fp_tower_chip(field, kind, 10) for field in E.FP_TOWER_CODES for kind in ("fp", "fp2_addsub", "fp2_mul")]
    rows_per_event = {"KeccakSponge": 24, "ShaExtend": 48, "ShaCompress": 80}      # MipsAir::costs scales a multi-row precompile by its rows per event (mips/mod.rs:588-595, applied where the cost table is built)
    got = {r.name: rows_per_event.get(r.name, 1) * (r.prep_width + r.main_width + 4 * r.perm_ext_width + (4 << r.log_quotient_degree)) for r in recs}
    assert got == want


def test_mul_constraints_hold(oracle):
    rec = chips.record_mul_constraints()
    for n in (0, 64, 3000):
        t = F.from_monty(oracle.tracegen_mul(E.synthetic_mul_events(n, seed=n + 1)))
        assert air.debug_constraints(rec.b, t) == []
    for col in (20, 30, 36, 53, 57):     # a carry, a product byte, b_sign_extend, the 16-bit limb of the clk difference, clk
        bad = t.copy()
        rows = np.nonzero(bad[:, 55] * bad[:, 52])[0]     # rows that write HI, previous access in the same shard
        bad[rows[2], col] = (int(bad[rows[2], col]) + 1) % F.P
        assert {row for _, row in air.debug_constraints(rec.b, bad)} == {rows[2]}, col
    kinds = [lk.kind for lk in rec.sends]
    assert kinds.count(air.KIND_BYTE) == 16 and kinds.count(air.KIND_MEMORY) == 1
    assert sorted(lk.kind for lk in rec.receives) == sorted([air.KIND_INSTRUCTION, air.KIND_MEMORY])
    # per-row cost = main + permutation + quotient columns, as pinned in the reference's mips_costs.json (Mul: 110)
    rc = chips.record_mul_chip(10)
    assert rc.main_width + 4 * rc.perm_ext_width + 8 == 110


def test_divrem_constraints_hold(oracle):
    rec = chips.record_divrem_constraints()
    for n in (0, 64, 3000):
        t = F.from_monty(oracle.tracegen_divrem(E.synthetic_divrem_events(n, seed=n + 1)))
        assert air.debug_constraints(rec.b, t) == []
    ev = E.synthetic_divrem_events(3000, seed=3001)
    row = int(np.nonzero((ev["opcode"] == E.DIV) & (ev["c"] != 0) & (ev["b"] > 0x80000000) & (ev["c"] < 0x80000000))[0][0])
    # remainder, c * quotient, a carry, is_c_0, rem_neg, the SLTU multiplicity, a HI limb (the quotient is tied only through lookups)
    for col in (14, 30, 38, 47, 88, 90, 103):
        bad = t.copy()
        bad[row, col] = (int(bad[row, col]) + 1) % F.P
        assert {r for _, r in air.debug_constraints(rec.b, bad)} == {row}, col
    kinds = [lk.kind for lk in rec.sends]
    assert kinds.count(air.KIND_INSTRUCTION) == 4 and kinds.count(air.KIND_BYTE) == 13 and kinds.count(air.KIND_MEMORY) == 1
    assert sorted(lk.kind for lk in rec.receives) == sorted([air.KIND_INSTRUCTION, air.KIND_INSTRUCTION, air.KIND_MEMORY])
    rc = chips.record_divrem_chip(10)   # mips_costs.json: DivRem 162
    assert rc.main_width + 4 * rc.perm_ext_width + 8 == 162
    # the special cases take the reference's values: x / 0 = 2^32 - 1 remainder x; i32::MIN / -1 wraps
    sp = E.make_divrem_events([E.DIV, E.DIVU, E.MOD, E.DIV, E.MOD], [7, 9, 0x80000000, 0x80000000, 0xfffffff9],
                              [0, 0, 0xffffffff, 0xffffffff, 0xfffffffe])
    assert sp["a"].tolist() == [0xffffffff, 0xffffffff, 0, 0x80000000, 0xffffffff] and sp["hi"].tolist() == [7, 9, 0, 0, 0]
    st = F.from_monty(oracle.tracegen_divrem(sp))
    assert air.debug_constraints(rec.b, st) == [] and st[:5, 61].tolist() == [0, 0, 1, 1, 0]   # is_overflow


def test_mov_cond_constraints_hold(oracle):
    rec = chips.record_mov_cond_constraints()
    for n in (0, 64, 3000):
        t = F.from_monty(oracle.tracegen_mov_cond(E.synthetic_mov_cond_events(n, seed=n + 1)))
        assert air.debug_constraints(rec.b, t) == []
    t = t.copy()
    t[5, 19] ^= 1            # c_eq_0.is_zero_byte[0].result
    assert {row for _, row in air.debug_constraints(rec.b, t)} == {5}
    # the instruction lookup carries prev_a in the `hi` word and is_rw_a = is_mne + is_meq
    ev = E.synthetic_mov_cond_events(8, seed=2)
    t = F.from_monty(oracle.tracegen_mov_cond(ev))
    main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
    vals = np.array([v.apply_np({}, main) for v in rec.receives[0].values]).T
    for i, e in enumerate(ev):
        assert vals[i, 6] == e["opcode"] and vals[i, 19:23].tolist() == [(int(e["prev_a"]) >> (8 * k)) & 0xff for k in range(4)]
        assert vals[i, 24] == (1 if e["opcode"] in (E.MEQ, E.MNE) else 0) and vals[i, 27] == 1


def test_constraints_hold_on_reference_vectors(oracle):
    for chip, ev in golden_events().items():
        if chip == E.CHIP_BITWISE:
            continue  # its NOR vector states a truncated result (see test_tracegen); the AIR has no arithmetic on a/b/c anyway
        rec = chips.record_constraints(chip)
        t = F.from_monty(oracle.tracegen_alu(chip, ev))
        assert air.debug_constraints(rec.b, t) == [], E.CHIP_NAMES[chip]


@pytest.mark.parametrize("chip,col", [(E.CHIP_ADD_SUB, 6), (E.CHIP_LT, 27), (E.CHIP_SHIFT_LEFT, 31), (E.CHIP_SHIFT_RIGHT, 30),
                                      (E.CHIP_BITWISE, 14), (E.CHIP_CLO_CLZ, 11)])
def test_corrupted_cell_is_caught(oracle, chip, col):
    rec = chips.record_constraints(chip)
    _, t = rows_for(oracle, chip, 100)
    t = t.copy()
    t[7, col] = (int(t[7, col]) + 1) % F.P
    bad = air.debug_constraints(rec.b, t)
    assert bad and all(row == 7 for _, row in bad)


def test_lookup_shapes():
    want = {  # (byte sends, instruction receives): counted from the reference's eval
        E.CHIP_ADD_SUB: (6, 2), E.CHIP_BITWISE: (4, 1), E.CHIP_LT: (3, 1), E.CHIP_SHIFT_LEFT: (4, 1), E.CHIP_SHIFT_RIGHT: (25, 1),
        E.CHIP_CLO_CLZ: (3, 1)}
    for chip, (ns, nr) in want.items():
        rec = chips.record_constraints(chip)
        sends = [lk for lk in rec.sends if lk.kind == air.KIND_BYTE]
        assert (len(sends), len(rec.receives)) == (ns, nr)
        assert all(len(lk.values) == 5 for lk in sends)
        # CloClz also sends one instruction (SRL) to the ShiftRight chip
        assert [lk.kind for lk in rec.sends if lk.kind != air.KIND_BYTE] == ([air.KIND_INSTRUCTION] if chip == E.CHIP_CLO_CLZ else [])
        assert all(lk.kind == air.KIND_INSTRUCTION and len(lk.values) == 28 for lk in rec.receives)


def test_lookup_values_on_a_row(oracle):
    """Bitwise, XOR 10 ^ 19 = 25: the byte lookups carry (XOR, a_i, 0, b_i, c_i) and the instruction receive the CPU opcode."""
    ev = E.make_alu_events([E.XOR], [10], [19], pc0=0x400)
    t = F.from_monty(oracle.tracegen_alu(E.CHIP_BITWISE, ev))
    rec = chips.record_constraints(E.CHIP_BITWISE)
    main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
    vals = [int(v.apply_np({}, main)[0]) for v in rec.sends[0].values]
    assert vals == [chips.B_XOR, 25, 0, 10, 19] and int(rec.sends[0].multiplicity.apply_np({}, main)[0]) == 1
    recv = [int(v.apply_np({}, main)[0]) for v in rec.receives[0].values]
    assert recv == [0, 0, 0x400, 0x404, 0x408, 0, E.XOR, 25, 0, 0, 0, 10, 0, 0, 0, 19, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]
    assert int(rec.receives[0].multiplicity.apply_np({}, main)[1]) == 0   # padding row: multiplicity 0


def test_to_virtual_pair_rejects_products():
    b = air.AirBuilder(3, 0, 0)
    l = b.main()[0]
    with pytest.raises(ValueError):
        air.to_virtual_pair(l[0] * l[1])
    with pytest.raises(ValueError):
        air.to_virtual_pair(b.main()[1][0])
    vp = air.to_virtual_pair((l[0] - l[2]) * 5 + 7)
    assert sorted(vp.terms) == [(True, 0, 5), (True, 2, F.P - 5)] and vp.constant == 7


# ---- GPU ------------------------------------------------------------------------------------------------------------

def alu_shard(oracle, log_rows, seed=11):
    """Eleven real chips (six ALU chips, Jump, MovCond, Branch, Mul, DivRem); AddSub at 2^log_rows rows, the others smaller; every
    trace also as oracle rows. The SRL events the executor derives from CLO/CLZ go to the ShiftRight chip
    (dependencies.rs:105-122), the ADD events it derives from JumpDirect to the AddSub chip, the ADD / MULT / SLTU events it
    derives from divisions to the AddSub, Mul and Lt chips."""
    spec = [(E.CHIP_ADD_SUB, log_rows, 0.85), (E.CHIP_BITWISE, log_rows - 1, 1.0), (E.CHIP_LT, log_rows - 1, 0.35),
            (E.CHIP_SHIFT_LEFT, log_rows - 2, 0.55), (E.CHIP_SHIFT_RIGHT, log_rows - 2, 0.6), (E.CHIP_CLO_CLZ, log_rows - 3, 0.8)]
    streams = {chip: E.synthetic_alu_events(chip, int((1 << lh) * fill), seed=seed + chip) for chip, lh, fill in spec}
    jumps = E.synthetic_jump_events(int((1 << (log_rows - 3)) * 0.75), seed=seed + 40)
    streams[E.CHIP_SHIFT_RIGHT] = np.concatenate([streams[E.CHIP_SHIFT_RIGHT], E.cloclz_dependencies(streams[E.CHIP_CLO_CLZ])])
    branches = E.synthetic_branch_events(int((1 << (log_rows - 3)) * 0.7), seed=seed + 60)
    lt_dep, add_dep = E.branch_dependencies(branches)                                                 # dependencies.rs:181-227
    divs = E.synthetic_divrem_events(int((1 << (log_rows - 3)) * 0.9), seed=seed + 80)
    div_add, mul_dep, div_lt = E.divrem_dependencies(divs)                                            # dependencies.rs:12-103
    streams[E.CHIP_ADD_SUB] = np.concatenate([streams[E.CHIP_ADD_SUB], E.jump_dependencies(jumps), add_dep, div_add])   # dependencies.rs:230-248
    streams[E.CHIP_LT] = np.concatenate([streams[E.CHIP_LT], lt_dep, div_lt])
    recs, evs = [], []
    for chip, lh, _ in spec:
        rc = chips.record_chip(chip, lh)
        rc.trace = oracle.tracegen_alu(chip, streams[chip], lh)
        recs.append(rc)
        evs.append((chip, streams[chip], lh))
    jc = chips.record_jump_chip(log_rows - 3)
    jc.trace = oracle.tracegen_jump(jumps, log_rows - 3)
    recs.append(jc)
    evs.append(("jump", jumps, log_rows - 3))
    movs = E.synthetic_mov_cond_events(int((1 << (log_rows - 3)) * 0.6), seed=seed + 50)
    mc = chips.record_mov_cond_chip(log_rows - 3)
    mc.trace = oracle.tracegen_mov_cond(movs, log_rows - 3)
    recs.append(mc)
    evs.append(("mov_cond", movs, log_rows - 3))
    bc = chips.record_branch_chip(log_rows - 3)
    bc.trace = oracle.tracegen_branch(branches, log_rows - 3)
    recs.append(bc)
    evs.append(("branch", branches, log_rows - 3))
    muls = np.concatenate([E.synthetic_mul_events(int((1 << (log_rows - 2)) * 0.5), seed=seed + 70), mul_dep])
    mu = chips.record_mul_chip(log_rows - 2)
    mu.trace = oracle.tracegen_mul(muls, log_rows - 2)
    recs.append(mu)
    evs.append(("mul", muls, log_rows - 2))
    dv = chips.record_divrem_chip(log_rows - 3)
    dv.trace = oracle.tracegen_divrem(divs, log_rows - 3)
    recs.append(dv)
    evs.append(("divrem", divs, log_rows - 3))
    return recs, evs


def device_trace(ctx, chip, ev, lh, blu=None):
    if chip == "jump":
        return ctx.tracegen_jump(ev, lh)
    if chip == "mov_cond":
        return ctx.tracegen_mov_cond(ev, lh)
    if chip == "branch":
        return ctx.tracegen_branch(ev, lh, blu)
    if chip == "mul":
        return ctx.tracegen_mul(ev, lh, blu)
    if chip == "divrem":
        return ctx.tracegen_divrem(ev, lh, blu)
    return ctx.tracegen_alu(chip, ev, lh, blu)


def mirror_chip(rec, kinds=None):
    """A chip that receives exactly what `rec` sends and sends what it receives (one column per lookup value plus a
    multiplicity column, filled from rec's trace), so that the pair's local cumulative sums cancel. It stands in for
    the chips on the other side of the ALU chips' lookups that are not built (Cpu: crates/core/machine/src/cpu/; with
    kinds=None also Byte). `kinds` restricts the mirror to lookups of those kinds. Lookups between two chips that are
    both in the shard are left alone: the SRL instructions CloClz sends (its instruction *sends*) and ShiftRight
    receives (its rows at the placeholder pc UNUSED_PC)."""
    t = F.from_monty(rec.trace)
    main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
    from_cpu = (main[0] != E.UNUSED_PC).astype(np.uint64)   # column 0 is `pc` in every ALU chip
    cols, sends, receives = [], [], []
    for src, dst in ((rec.sends, receives), (rec.receives, sends)):
        for lk in src:
            if kinds is not None and lk.kind not in kinds:
                continue
            if lk.kind == air.KIND_INSTRUCTION and src is rec.sends:
                continue
            first = len(cols)
            for v in lk.values:
                cols.append(v.apply_np({}, main))
            mult = lk.multiplicity.apply_np({}, main)
            cols.append(mult * from_cpu if lk.kind == air.KIND_INSTRUCTION else mult)
            vals = [air.VirtualPairCol.single_main(first + j) for j in range(len(lk.values))]
            dst.append(air.Lookup(vals, air.VirtualPairCol.single_main(first + len(lk.values)), lk.kind))
    width = len(cols)
    b = air.AirBuilder(width, 0, air.local_permutation_trace_width(len(sends) + len(receives), 2))
    air.eval_permutation_constraints(b, sends, receives, 2, False)
    program = b.assemble()
    trace = np.empty((t.shape[0], width), dtype=np.uint32)
    for c in range(width):
        trace[:, c] = F.to_monty(cols[c])
    return chips.RecordedChip(name=rec.name + "Mirror", log_height=rec.log_height, main_width=width, sends=sends,
                              receives=receives, program=program, lookups_blob=air.encode_lookups(sends, receives),
                              num_constraints=int(program[2]), trace=trace)


def lookup_tally(all_chips):
    """Signed multiset of everything the chips send (+) and receive (-): (kind, values) -> multiplicity mod p."""
    tally = {}
    for r in all_chips:
        t = F.from_monty(r.trace)
        main = {c: t[:, c].astype(np.uint64) for c in range(t.shape[1])}
        prep = {}
        if r.prep_trace is not None:
            pt = F.from_monty(r.prep_trace)
            prep = {c: pt[:, c].astype(np.uint64) for c in range(pt.shape[1])}
        for sign, lks in ((1, r.sends), (-1, r.receives)):
            for lk in lks:
                vals = np.stack([np.broadcast_to(v.apply_np(prep, main), (t.shape[0],)) for v in lk.values], axis=1)
                mult = np.broadcast_to(lk.multiplicity.apply_np(prep, main), (t.shape[0],))
                for row in np.nonzero(mult)[0]:
                    key = (lk.kind,) + tuple(int(x) for x in vals[row])
                    tally[key] = (tally.get(key, 0) + sign * int(mult[row])) % F.P
    return tally


def byte_shard(oracle, log_rows, seed):
    """alu_shard + the Byte chip (multiplicities from the same events) + mirrors for the chips that are not built (Cpu:
    instruction lookups; the memory chips: the HI register access of Mul)."""
    recs, evs = alu_shard(oracle, log_rows, seed=seed)
    streams = [(chip, ev) for chip, ev, _ in evs if not isinstance(chip, str)]
    byte = chips.record_byte_chip(prep_index=0)
    extra = np.zeros((1 << 16, 10), dtype=np.uint32)   # the Branch chip's range checks (not-taken branches), Mul's lookups
    for chip, ev, lh in evs:
        if chip == "branch":
            oracle.tracegen_branch(ev, lh, extra)
        if chip == "mul":
            oracle.tracegen_mul(ev, lh, extra)
        if chip == "divrem":
            oracle.tracegen_divrem(ev, lh, extra)
    byte.trace = oracle.tracegen_byte_mults(streams, extra)
    byte.prep_trace = oracle.tracegen_byte_table()
    mirrors = [mirror_chip(r, kinds=(air.KIND_INSTRUCTION, air.KIND_MEMORY)) for r in recs]
    return recs, evs, byte, mirrors


def test_lookups_balance_between_real_chips(oracle):
    """Every byte lookup an ALU / control-flow chip sends is received by the Byte chip with the multiplicity counted from
    the same events; the instructions CloClz, Jump and Branch send are received by ShiftRight, AddSub and Lt. What is left
    is exactly the traffic with the chips that are not built (instruction receives from Cpu, the HI-register accesses of Mul
    and DivRem); what DivRem sends (MULT / MULTU with the upper word, ADD, SLTU) is received by Mul, AddSub and Lt."""
    recs, evs, byte, mirrors = byte_shard(oracle, 8, seed=21)
    left = {k: v for k, v in lookup_tally(recs + [byte]).items() if v}
    assert left and {k[0] for k in left} == {air.KIND_INSTRUCTION, air.KIND_MEMORY}
    assert not any(lookup_tally(recs + [byte] + mirrors).values())


@pytest.mark.gpu
@pytest.mark.parametrize("log_rows,queries,pow_bits", [(6, 20, 8), (12, 84, 16)])
def test_gpu_alu_shard_proof(hip_ctx, oracle, log_rows, queries, pow_bits):
    from ziren_amd import prover, synth
    recs, evs = alu_shard(oracle, log_rows)
    mirrors = [mirror_chip(r) for r in recs]
    recs = recs + mirrors
    fri = abi.FriConfig(1, queries, pow_bits)
    pv = F.to_monty(F.SplitMix64(3).uniform_field(synth.PROOF_MAX_NUM_PVS))
    pv[synth.NUM_PV_ELTS:] = 0
    igcs = F.to_monty(F.SplitMix64(4).uniform_field(14))
    pc_start = F.to_monty(0x400000)
    hp = prover.HipProver(recs, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    if log_rows > 8:
        hp.specialize_quotient_kernels(recs[:len(evs)])   # the real chips through generated kernels, the mirrors interpreted
    pk = hp.setup([], [], pc_start, igcs)
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    born = [device_trace(hip_ctx, chip, ev, lh) for chip, ev, lh in evs]   # these traces never exist on the host
    born += [hip_ctx.upload(m.trace) for m in mirrors]
    proof = hp.prove_shard(pk, pv, born, ch).copy()
    # oracle: the same recorded chips over the oracle's own rows
    opk = oracle.Pk([], [], pc_start, igcs, 1)
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, recs, [r.trace for r in recs], pv, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert ch.as_tuple() == och.as_tuple()
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    for m in born:
        m.free()


@pytest.mark.gpu
def test_gpu_alu_and_byte_chips_shard(hip_ctx, oracle):
    """Five ALU chips + the real Byte chip: preprocessed table and multiplicity trace generated on the device from the
    same events, so the byte side of the LogUp argument balances between real chips; only the instruction lookups
    (whose sender is the Cpu chip) are mirrored. The pk holds the Byte table: four opening rounds."""
    from ziren_amd import prover, synth
    log_rows = 11
    recs, evs, byte, mirrors = byte_shard(oracle, log_rows, seed=21)
    all_chips = recs + [byte] + mirrors
    fri = abi.FriConfig(1, 84, 16)
    pv = F.to_monty(F.SplitMix64(3).uniform_field(synth.PROOF_MAX_NUM_PVS))
    pv[synth.NUM_PV_ELTS:] = 0
    igcs = F.to_monty(F.SplitMix64(4).uniform_field(14))
    pc_start = F.to_monty(0x400000)
    hp = prover.HipProver(all_chips, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    pk = hp.setup([hip_ctx.tracegen_byte_table()], [0], pc_start, igcs)       # device-born preprocessed table
    opk = oracle.Pk([byte.prep_trace], [0], pc_start, igcs, 1)
    assert np.array_equal(pk.commit, opk.commitment())
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    blu = hip_ctx.byte_lookups()
    born = [device_trace(hip_ctx, chip, ev, lh, blu) for chip, ev, lh in evs]   # traces + byte lookups in one pass
    born.append(hip_ctx.tracegen_byte_mults(blu))
    born += [hip_ctx.upload(m.trace) for m in mirrors]
    proof = hp.prove_shard(pk, pv, born, ch).copy()
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, all_chips, [c.trace for c in all_chips], pv, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(opk, all_chips, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    # without the Byte chip's multiplicities the byte lookups do not balance: the verifier's cumulative-sum check fails
    born[len(evs)].free()
    empty = hip_ctx.byte_lookups()
    born[len(evs)] = hip_ctx.tracegen_byte_mults(empty)
    ch2 = start.copy()
    bad = hp.prove_shard(pk, pv, born, ch2).copy()
    assert oracle.verify_shard(opk, all_chips, fri, synth.NUM_PV_ELTS, start.copy(), bad) != 0
    for m in born:
        m.free()
