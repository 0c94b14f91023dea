"""A coherent core shard: a program executed by ziren_amd/miniexec.py, its CpuEvents through the Cpu
chip, the per-chip events through the ALU / Mul / DivRem / Branch / Jump / MovCond chips, the Program and Byte tables —
every instruction, program, byte and memory lookup is exchanged between real chips (the MemoryLocal chip closes the
register accesses); only what MemoryLocal forwards to the Global chip (kind Global), which is not built, is mirrored."""
import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M

from test_chip_airs import lookup_tally, mirror_chip

PC_BASE, SHARD = 0x1000, 1


def log2_rows(n):
    h = 16
    while h < n:
        h <<= 1
    return h.bit_length() - 1


def cpu_shard(oracle, n_cycles, seed=1):
    """(chips with oracle traces, device work list, byte chip, program chip, public values)."""
    prog, rec0, pv = M.run(n_cycles, seed=seed, shard=SHARD, pc_base=PC_BASE, halt=True)   # ... eight COMMITs and HALT at the end
    rec = M.add_dependencies(rec0)
    extra = np.zeros((1 << 16, 10), dtype=np.uint32)
    recs, work = [], []
    lh = log2_rows(len(rec.cpu))
    cpu = chips.record_cpu_chip(lh)
    cpu.trace = oracle.tracegen_cpu(rec.cpu, prog, PC_BASE, SHARD, lh, extra)
    recs.append(cpu)
    work.append(("cpu", rec.cpu, lh))
    for chip in sorted(E.CHIP_NAMES):
        ev = rec.alu[chip]
        lh = log2_rows(len(ev))
        rc = chips.record_chip(chip, lh)
        rc.trace = oracle.tracegen_alu(chip, ev, lh)
        recs.append(rc)
        work.append((chip, ev, lh))
    for name, ev, record, gen in (("syscall_instrs", rec.syscall, chips.record_syscall_instrs_chip, oracle.tracegen_syscall_instrs),
                                  ("jump", rec.jump, chips.record_jump_chip, oracle.tracegen_jump),
                                  ("mov_cond", rec.mov_cond, chips.record_mov_cond_chip, oracle.tracegen_mov_cond)):
        lh = log2_rows(len(ev))
        rc = record(lh)
        rc.trace = gen(ev, lh)
        recs.append(rc)
        work.append((name, ev, lh))
    for name, ev, record, gen in (("branch", rec.branch, chips.record_branch_chip, oracle.tracegen_branch),
                                  ("memory_instrs", rec.mem_instr, chips.record_memory_instrs_chip, oracle.tracegen_memory_instrs),
                                  ("misc_instrs", rec.misc, chips.record_misc_instrs_chip, oracle.tracegen_misc_instrs),
                                  ("mul", rec.mul, chips.record_mul_chip, oracle.tracegen_mul),
                                  ("divrem", rec.divrem, chips.record_divrem_chip, oracle.tracegen_divrem)):
        lh = log2_rows(len(ev))
        rc = record(lh)
        rc.trace = gen(ev, lh, extra)
        recs.append(rc)
        work.append((name, ev, lh))
    lh = log2_rows(-(-len(rec.memory_local) // M.MEMORY_LOCAL_ENTRIES_PER_ROW))
    ml = chips.record_memory_local_chip(lh)
    ml.trace = oracle.tracegen_memory_local(rec.memory_local, lh)
    recs.append(ml)
    work.append(("memory_local", rec.memory_local, lh))
    ge = M.global_lookup_events(rec.memory_local)
    lh = log2_rows(len(ge))
    gl = chips.record_global_chip(lh)
    gl.trace = oracle.tracegen_global(ge, lh, extra)
    recs.append(gl)
    work.append(("global", ge, lh))
    byte = chips.record_byte_chip(prep_index=0)
    byte.trace = oracle.tracegen_byte_mults([(c, ev) for c, ev, _ in work if not isinstance(c, str)], extra)
    byte.prep_trace = oracle.tracegen_byte_table()
    plh = log2_rows(len(prog))
    program = chips.record_program_chip(plh, prep_index=1)
    program.trace = oracle.tracegen_program(1, rec.cpu, prog, PC_BASE, plh)
    program.prep_trace = oracle.tracegen_program(0, rec.cpu, prog, PC_BASE, plh)
    return recs, work, byte, program, prog, pv




def test_miniexec_record_is_coherent():
    prog, rec, pv = M.run(1500, seed=5)
    cpu = rec.cpu
    assert len(cpu) == 1500 and pv["start_pc"] == 0x1000 and (np.diff(cpu["clk"].astype(np.int64)) == 5).all()
    assert (cpu["pc"][1:] == cpu["next_pc"][:-1]).all() and (cpu["next_pc"][1:] == cpu["next_next_pc"][:-1]).all()
    assert len(np.unique(cpu["pc"])) == 1500                      # forward only: every pc runs once
    total = sum(len(v) for v in rec.alu.values()) + len(rec.mul) + len(rec.divrem) + len(rec.branch) + len(rec.jump) + len(rec.mov_cond) + len(rec.mem_instr) \
        + len(rec.syscall) + len(rec.misc)
    assert total == 1500                                          # one chip event per cycle (emit_events)
    assert min(len(v) for v in rec.alu.values()) > 0 and len(rec.jump) > 0 and len(rec.divrem) > 0
    assert set(rec.mem_instr["opcode"].tolist()) == set(range(E.LB, E.SC + 1))     # all fourteen loads and stores
    # register accesses chain: every record's previous (shard, timestamp) is the last access to that register
    ins = prog[(cpu["pc"] - 0x1000) // 4]
    last = {}
    for e, i in zip(cpu, ins):
        for name, reg, is_reg in (("c_record", int(i["op_c"]), not i["imm_c"]), ("b_record", int(i["op_b"]), not i["imm_b"]),
                                  ("a_record", int(i["op_a"]), True), ("hi_record", M.REG_HI, True)):
            r = e[name]
            if r["tag"] == M.TAG_NONE or not is_reg:
                continue
            if name == "a_record" and int(i["opcode"]) in (E.MULT, E.MULTU, E.DIV, E.DIVU):
                reg = M.REG_LO
            body = r["read"] if r["tag"] == M.TAG_READ else r["write"]
            assert (int(body["prev_shard"]), int(body["prev_timestamp"])) == last.get(reg, (0, 0))
            last[reg] = (int(body["shard"]), int(body["timestamp"]))
    taken = E.branch_taken(rec.branch)
    assert taken.any() and (~taken).any()
    # a longer run executes every opcode of the instruction set (crates/core/executor/src/opcode.rs:26-87: 0..55)
    prog, rec, _ = M.run(20000, seed=3, halt=True)
    assert set(prog["opcode"][(rec.cpu["pc"] - 0x1000) // 4].tolist()) == set(range(56))


def test_memory_instrs_constraints_hold(oracle):
    rec = chips.record_memory_instrs_constraints()
    prog, r, pv = M.run(8000, seed=11)
    ev = r.mem_instr
    assert len(ev) > 800 and (ev["b"] + ev["c"]).min() < 256          # some addresses fit one byte (the LTU lookup)
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = F.from_monty(oracle.tracegen_memory_instrs(ev, -1, counts))
    assert air.debug_constraints(rec.b, t) == [] and not t[len(ev):].any()
    # the loaded / stored values are the executor's: a load's `a` and a store's memory word follow events.load_value / store_value
    for e in ev[:200]:
        addr = (int(e["b"]) + int(e["c"])) & 0xffffffff
        if e["mem_tag"] == M.TAG_READ:
            assert int(e["a"]) == E.load_value(int(e["opcode"]), int(e["mem"][0]), addr, int(e["prev_a_val"]))
        else:
            assert int(e["mem"][0]) == E.store_value(int(e["opcode"]), int(e["mem"][3]), addr, int(e["prev_a_val"]))
    lw = int(np.nonzero(ev["opcode"] == E.LW)[0][0])
    for col in (4, 30, 35, 57, 70):   # op_a byte, addr byte, ls bits, loaded word byte, unsigned_mem_val byte (of a load)
        bad = t.copy()
        bad[lw, col] = (int(bad[lw, col]) + 1) % F.P
        assert {row for _, row in air.debug_constraints(rec.b, bad)} == {lw}, col
    kinds = [lk.kind for lk in rec.sends]
    assert kinds.count(air.KIND_INSTRUCTION) == 2 and kinds.count(air.KIND_BYTE) == 6 and kinds.count(air.KIND_MEMORY) == 1
    c = chips.record_memory_instrs_chip(10)     # mips_costs.json: MemoryInstrs 115
    assert c.main_width + 4 * c.perm_ext_width + 8 == 115
    # dependencies: one ADD per event, one SUB per negative LB / LH
    dep = E.memory_dependencies(ev)
    neg = int(t[:len(ev), 76].sum())
    assert (dep["opcode"] == E.ADD).sum() == len(ev) and (dep["opcode"] == E.SUB).sum() == neg > 0


def test_syscall_instrs_and_halting_program(oracle):
    """A program that ends as real ones do — eight COMMIT syscalls publishing a value digest, then HALT: the SyscallInstrs rows
    satisfy the AIR against the public values (committed_value_digest, exit_code), the Cpu chip's halt path holds (is_halt, last
    real row, public next_pc = 0), and wrong public values are caught by the chip that owns them."""
    rec = chips.record_syscall_instrs_constraints()
    prog, r, pv = M.run(400, seed=6, halt=True)
    assert len(r.syscall) == 9 and r.syscall["syscall_id"].tolist() == [E.SYS_COMMIT] * 8 + [E.SYS_HALT] and pv["next_pc"] == 0
    assert r.syscall["arg2"][:8].tolist() == pv["committed_value_digest"] and r.syscall["arg1"][:8].tolist() == list(range(8))
    pvs = F.from_monty(M.public_values(pv))
    t = F.from_monty(oracle.tracegen_syscall_instrs(r.syscall))
    assert air.debug_constraints(rec.b, t, public_values=pvs) == [] and not t[9:].any()
    assert t[:9, 5].tolist() == [0] * 8 + [1] and t[:8, 38:46].tolist() == np.eye(8, dtype=int).tolist()      # is_halt; index_bitmap
    wrong = dict(pv, committed_value_digest=[w ^ (1 << 9) if i == 3 else w for i, w in enumerate(pv["committed_value_digest"])])
    assert {row for _, row in air.debug_constraints(rec.b, t, public_values=F.from_monty(M.public_values(wrong)))} == {3}
    assert {row for _, row in air.debug_constraints(rec.b, t, public_values=F.from_monty(M.public_values(dict(pv, exit_code=1))))} == {8}
    cpu = chips.record_cpu_constraints()
    tc = F.from_monty(oracle.tracegen_cpu(r.cpu, prog, PC_BASE, SHARD))
    n = len(r.cpu)
    assert air.debug_constraints(cpu.b, tc, public_values=pvs) == [] and tc[n - 1, 24] == 1 and tc[n - 1, 25] == 0 and not tc[n:, 65].any()
    kinds = sorted(lk.kind for lk in rec.sends)
    assert kinds == [air.KIND_SYSCALL, air.KIND_SYSCALL_RESULT] and [lk.kind for lk in rec.receives] == [air.KIND_INSTRUCTION]


def test_misc_instrs_constraints_hold(oracle):
    rec = chips.record_misc_instrs_constraints()
    prog, r, pv = M.run(25000, seed=12)
    ev = r.misc
    assert set(ev["opcode"].tolist()) == {E.SEXT, E.EXT, E.INS, E.TEQ, E.MADDU, E.MSUBU, E.MADD, E.MSUB} and len(ev) > 700
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = F.from_monty(oracle.tracegen_misc_instrs(ev, -1, counts))
    assert air.debug_constraints(rec.b, t) == [] and not t[len(ev):].any()
    # the reference's own program vectors (misc/others/mod.rs:31-58): SEXT of 0x8f8f, EXT / INS fields
    assert E.misc_result(E.SEXT, 0x8f8f, 0) == 0xffffff8f and E.misc_result(E.SEXT, 0x8f8f, 1) == 0xffff8f8f and E.misc_result(E.SEXT, 0xf, 0) == 0xf
    assert E.misc_result(E.EXT, 0x8f8f, 0x21) == (0x8f8f >> 1) & 3 and E.misc_result(E.INS, 0xbeef, 0x3e0, prev_a=0x1234) == 0xbeef
    assert E.misc_result(E.INS, 0xdead, 0x2e8, prev_a=0xffffffff) == 0xff00adff | 0x00de0000
    # EXT and INS are proven entirely by the instructions they send (shifts, rotations, one addition): only the split of `c` is a constraint
    for op, cols in ((E.SEXT, (4, 21)), (E.EXT, (20,)), (E.INS, (21,)), (E.MADDU, (4, 24, 36, 58)), (E.MSUB, (8, 47)), (E.TEQ, (22,))):
        row = int(np.nonzero(ev["opcode"] == op)[0][1])      # a result byte / an intermediate of that opcode's layout
        for col in cols:
            bad = t.copy()
            bad[row, col] = (int(bad[row, col]) + 1) % F.P
            assert {x for _, x in air.debug_constraints(rec.b, bad)} == {row}, (op, col)
    kinds = [lk.kind for lk in rec.sends]
    assert kinds.count(air.KIND_INSTRUCTION) == 9 and kinds.count(air.KIND_BYTE) == 20 and kinds.count(air.KIND_MEMORY) == 1
    mul, sll, sr, add = E.misc_dependencies(ev)
    n = {op: int((ev["opcode"] == op).sum()) for op in (E.EXT, E.INS)}
    assert len(mul) == int((ev["opcode"] >= E.MADDU).sum() - (ev["opcode"] >= E.MEQ).sum()) and len(sll) == n[E.EXT] + n[E.INS]
    assert len(sr) == n[E.EXT] + 4 * n[E.INS] and len(add) == n[E.INS]


def test_cpu_row_by_hand(oracle):
    """The event of the reference's test_generate_cpu_trace_ffi_eq_rust (cpu/trace.rs:283-311; a trace-only test: its records are
    not a coherent execution) with the program [ADD $29, $0, 1]: every column follows cpu/trace.rs:117-257 by hand."""
    prog = np.zeros(1, dtype=M.INSTRUCTION)
    prog[0] = (E.ADD, 29, [0, 0], 0, 1, 0, 1, [0, 0], 1, [0, 0, 0], 0)
    ev = np.zeros(1, dtype=M.CPU_EVENT)
    e = ev[0]
    e["clk"], e["pc"], e["next_pc"], e["next_next_pc"], e["a"], e["b"], e["c"] = 0, 0, 1, 2, 5, 10, 15
    e["a_record"]["tag"], e["a_record"]["write"] = M.TAG_WRITE, (5, 1, 2, 1, 1, 1)
    e["b_record"]["tag"], e["b_record"]["read"] = M.TAG_READ, (5, 0, 1, 0, 0)
    e["c_record"]["tag"], e["c_record"]["read"] = M.TAG_READ, (5, 0, 2, 0, 0)
    e["hi"]["tag"], e["hi"]["value"] = 0, 1
    e["hi_record"]["tag"] = M.TAG_NONE
    e["memory_record"]["tag"], e["memory_record"]["read"] = M.TAG_READ, (5, 0, 3, 0, 0)
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    r = F.from_monty(oracle.tracegen_cpu(ev, prog, 0, 0, -1, counts))
    assert r.shape == (16, 67)
    row = r[0].tolist()
    assert row[0:8] == [0, 0, 0, 0, 0, 0, 1, 2]                                  # shard, clk limbs, shard/clk to send, pc, next_pc, next_next_pc
    assert row[8:21] == [E.ADD, 29, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1]             # opcode, op_a, op_b, op_c, op_a_0, imm_b, imm_c
    assert row[21:26] == [0, 0, 0, 0, 1]                                         # no extra cycles, not rw_a / check_memory / halt; sequential
    assert row[26:34] == [5, 0, 0, 0, 1, 0, 0, 0]                                # op_a_value, hi_or_prev_a = hi
    assert row[34:47] == [1, 0, 0, 0, 5, 0, 0, 0, 1, 1, 1, 0, 0]                 # a: prev_value, value, prev (shard 1, clk 1), same shard, 2 - 1 - 1 = 0
    assert row[47:56] == [5, 0, 0, 0, 0, 0, 1, 0, 0]                             # b (the record's value, not event.b): prev (0, 0), 1 - 0 - 1 = 0
    assert row[56:65] == [5, 0, 0, 0, 0, 0, 1, 1, 0]                             # c: 2 - 0 - 1 = 1
    assert row[65:67] == [1, 0]
    assert (r[1:, 19] == 1).all() and (r[1:, 20] == 1).all() and (r[1:, 22] == 1).all() and not r[1:, 65].any()    # padding rows
    # lookups: shard and clk limbs (three), three accesses (two each), the two byte pairs of `a`
    assert counts.sum() == 11 and counts[0, 8] == 4 and counts[1, 8] == 1 and counts[0, 4] == 5 and counts[5 << 8, 4] == 1


def test_cpu_constraints_hold(oracle):
    rec = chips.record_cpu_constraints()
    for n, seed in ((1, 1), (16, 2), (17, 3), (2500, 4)):
        prog, r, pv = M.run(n, seed=seed)
        t = F.from_monty(oracle.tracegen_cpu(r.cpu, prog, PC_BASE, SHARD))
        assert air.debug_constraints(rec.b, t, public_values=F.from_monty(M.public_values(pv))) == []
    # padding rows: imm_b = imm_c = is_rw_a = 1 (cpu/trace.rs:57-60)
    assert t[2500:, 19].all() and t[2500:, 20].all() and t[2500:, 22].all() and not t[2500:, 65].any()
    for col, rows in ((5, {1199, 1200}), (6, {1199, 1200}), (1, {1199, 1200}), (65, {1199, 1200}), (27, {1200}), (40, {1200})):
        bad = t.copy()     # pc / next_pc / clk limb tie neighbouring rows; op_a_value and the access value only their own
        bad[1200, col] = (int(bad[1200, col]) + 1) % F.P
        assert {row for _, row in air.debug_constraints(rec.b, bad, public_values=F.from_monty(M.public_values(pv)))} <= rows | {1200}, col
        assert air.debug_constraints(rec.b, bad, public_values=F.from_monty(M.public_values(pv))), col
    wrong = dict(pv, next_pc=pv["next_pc"] + 4)
    assert {row for _, row in air.debug_constraints(rec.b, t, public_values=F.from_monty(M.public_values(wrong)))} == {2499}
    c = chips.record_cpu_chip(10)    # mips_costs.json: Cpu 119
    assert c.main_width + 4 * c.perm_ext_width + 8 == 119


def random_global_events(n, seed):
    rng = np.random.default_rng(seed)
    ev = np.zeros(n, dtype=M.GLOBAL_LOOKUP_EVENT)
    ev["message"] = rng.integers(0, 1 << 24, size=(n, 7))
    ev["message"][:, 0] = rng.integers(0, 3, size=n)           # shard numbers: u16
    ev["message"][:, 3:] = rng.integers(0, 256, size=(n, 4))   # value bytes
    ev["is_receive"] = rng.integers(0, 2, size=n)
    ev["kind"] = rng.choice([1, 6], size=n)                    # LookupKind::Memory / Syscall
    return ev


def test_septic_arithmetic_matches_the_reference_tables(oracle):
    """The oracle computes Frobenius by exponentiation; the reference keeps (z^i)^p and (z^i)^(p^2) as tables (tests/golden/septic_frobenius.json,
    copied by gen_septic_frobenius.py). Equal tables pin the reduction polynomial and the product; the square root is checked against
    an independent Python product."""
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "septic_frobenius.json")))
    rng = np.random.default_rng(4)
    a, b = rng.integers(0, F.P, size=7), rng.integers(0, F.P, size=7)
    fp, fp2, prod, root = oracle.septic_known_answers(a, b)
    assert fp.tolist() == gold["z_pow_p"] and fp2.tolist() == gold["z_pow_p2"]

    def mul(x, y):     # schoolbook product modulo z^7 = 8 - 2z, Python integers
        t = [0] * 13
        for i in range(7):
            for j in range(7):
                t[i + j] += int(x[i]) * int(y[j])
        for k in range(12, 6, -1):
            t[k - 7] += 8 * t[k]
            t[k - 6] -= 2 * t[k]
        return [v % F.P for v in t[:7]]
    assert prod.tolist() == mul(a, b)
    assert mul(root, root) == mul(a, a) and 0 < root[6] <= (F.P - 1) // 2
    assert root.tolist() in ([int(v) for v in a], [(F.P - int(v)) % F.P for v in a])
    # Frobenius is the p-th power: applying the table's linear map to a equals a^p computed through the product (p = 2^31 - 2^24 + 1)
    frob = [int(a[0])] + [0] * 6
    for i in range(1, 7):
        frob = [(f + int(a[i]) * g) % F.P for f, g in zip(frob, gold["z_pow_p"][i - 1])]
    acc, base, e = [1] + [0] * 6, [int(v) for v in a], F.P
    while e:
        if e & 1:
            acc = mul(acc, base)
        base = mul(base, base)
        e >>= 1
    assert acc == frob


def test_global_constraints_hold(oracle):
    """The Global chip: rows built by the restated generate_trace satisfy the recorded AIR (message -> curve point, sign of y by
    direction, running sum from the start digest); the shard's digest does not depend on the order of the messages; cost pinned."""
    rec = chips.record_global_constraints()
    prog, r, pv = M.run(300, seed=3, shard=SHARD, pc_base=PC_BASE, halt=True)
    ge = M.global_lookup_events(r.memory_local)
    assert len(ge) == 2 * len(r.memory_local) and ge["is_receive"].sum() == len(r.memory_local)
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = F.from_monty(oracle.tracegen_global(ge, -1, counts))
    assert t.shape == (256, M.GLOBAL_WIDTH) and air.debug_constraints(rec.b, t) == []
    assert counts.sum() == len(ge) == counts[0, 8] + counts[SHARD, 8]          # U16Range(shard) per message
    for col in (0, 7, 8, 16, 23, 30, 60, 61, 78, 85, 98):     # message, kind, offset bit, x, y, y6 bit, witness, direction, checker, sum
        bad = t.copy()
        bad[5, col] = (int(bad[5, col]) + 1) % F.P
        assert {row for _, row in air.debug_constraints(rec.b, bad)} == {5}, col
    bad = t.copy()
    bad[5, 64] = (int(bad[5, 64]) + 1) % F.P                 # initial digest: also breaks the previous row's hand-over
    assert {row for _, row in air.debug_constraints(rec.b, bad)} == {4, 5}
    bad = t.copy()
    bad[len(ge) + 3, 85] = (int(bad[len(ge) + 3, 85]) + 1) % F.P     # a padding row must carry the sum unchanged
    assert {row for _, row in air.debug_constraints(rec.b, bad)} == {len(ge) + 3}
    # the start digest alone (no events) and order independence of the final sum
    empty = F.from_monty(oracle.tracegen_global(ge[:0], 4))
    assert air.debug_constraints(rec.b, empty) == [] and list(empty[-1, 85:92]) == list(chips.SEPTIC_START_X)
    ev = random_global_events(200, 5)
    a = F.from_monty(oracle.tracegen_global(ev, -1))
    b = F.from_monty(oracle.tracegen_global(ev[np.random.default_rng(1).permutation(len(ev))], -1))
    assert air.debug_constraints(rec.b, a) == [] and np.array_equal(a[-1, 85:], b[-1, 85:]) and not np.array_equal(a[100, 85:], b[100, 85:])
    c = chips.record_global_chip(10)
    assert c.commit_scope_global and c.main_width + 4 * c.perm_ext_width + 8 == 115
    with pytest.raises(Exception):
        bad_ev = ev.copy()
        bad_ev["message"][3, 0] = 1 << 16
        oracle.tracegen_global(bad_ev, -1)


def complement(events):
    """The other side of every message: what the shards (or MemoryGlobalInit / Finalize) at the far end of these lookups put on the
    global table — same messages, opposite direction."""
    out = events.copy()
    out["is_receive"] ^= 1
    return out


def test_global_digests_of_both_sides_sum_to_zero(oracle):
    """Machine::verify's last check (crates/stark/src/machine.rs:657-671): the shard proofs' global_cumulative_sums and the key's initial
    one sum (SepticDigest::sum) to the zero digest exactly when every message is received as often as it is sent."""
    prog, r, pv = M.run(400, seed=2, shard=SHARD, pc_base=PC_BASE, halt=True)
    ge = M.global_lookup_events(r.memory_local)
    zero = F.to_monty(np.array(chips.SEPTIC_START_X + chips.SEPTIC_START_Y, dtype=np.uint64))
    here = oracle.tracegen_global(ge, -1)[-1, 85:]
    far = complement(ge)
    split = len(far) // 3                                         # the far side spread over two more shards, in another order
    there = [oracle.tracegen_global(part[::-1], -1)[-1, 85:] for part in (far[:split], far[split:])]
    assert oracle.global_digest_sum([here] + there + [zero])[1]
    assert oracle.global_digest_sum([zero, zero])[1] and oracle.global_digest_sum([zero])[1]
    assert not oracle.global_digest_sum([here, zero])[1] and not oracle.global_digest_sum([here, there[0], zero])[1]
    far["message"][7, 3] ^= 1                                      # one value byte differs at the far end
    there = [oracle.tracegen_global(part, -1)[-1, 85:] for part in (far[:split], far[split:])]
    assert not oracle.global_digest_sum([here] + there + [zero])[1]


def test_shard_lookups_balance(oracle):
    """Cpu sends every instruction the eleven chips receive, the Program table receives every fetch, the Byte table
    every byte lookup, MemoryLocal opens and closes every register's access chain and the Global chip receives its two messages
    per touched address: nothing is left, with no stand-in on any side."""
    recs, work, byte, program, prog, pv = cpu_shard(oracle, 1200, seed=9)
    assert not any(lookup_tally(recs + [byte, program]).values())
    n_addr = len(work[-2][1])
    left = {k: v for k, v in lookup_tally(recs[:-1] + [byte, program]).items() if v}      # without the Global chip
    assert {k[0] for k in left} == {air.KIND_GLOBAL, air.KIND_BYTE} and len(left) == 2 * n_addr + 2 and n_addr > 34    # registers and memory words; U16Range of shard 0 and 1
    without = {k: v for k, v in lookup_tally(recs[:-2] + [byte, program]).items() if v}
    assert {k[0] for k in without} == {air.KIND_MEMORY, air.KIND_BYTE}          # without MemoryLocal the access chains stay open
    assert {r.name for r in recs} >= {"Cpu", "MiscInstrs", "MemoryInstrs", "SyscallInstrs", "Mul", "DivRem", "MemoryLocal", "Global"} and len(recs) == 17


def test_oracle_proves_coherent_shard(oracle):
    """The restated verifier accepts the oracle's proof of the whole shard (local cumulative sum zero over real chips only; the
    Global chip's running curve sum is the proof's global_cumulative_sum) and rejects one made for a different next_pc (the
    Cpu chip's boundary constraint)."""
    from ziren_amd import synth
    recs, work, byte, program, prog, pv = cpu_shard(oracle, 500, seed=7)
    all_chips = recs + [byte, program]
    fri = abi.FriConfig(1, 84, 16)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    opk = oracle.Pk([byte.prep_trace, program.prep_trace], [0, 0], F.to_monty(PC_BASE), igcs, 1)
    start = oracle.new_challenger()
    opk.observe_into(start)
    for claimed, ok in ((pv, True), (dict(pv, next_pc=pv["next_pc"] + 4), False)):
        proof, _ = oracle.prove_shard(opk, all_chips, [c.trace for c in all_chips], M.public_values(claimed), fri, synth.NUM_PV_ELTS, start.copy())
        assert (oracle.verify_shard(opk, all_chips, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0) == ok


# ---- GPU ------------------------------------------------------------------------------------------------------------

def device_trace(ctx, chip, ev, lh, blu, prog):
    if chip == "cpu":
        return ctx.tracegen_cpu(ev, prog, PC_BASE, SHARD, lh, blu)
    if chip == "memory_local":
        return ctx.tracegen_memory_local(ev, lh)
    if chip == "syscall_instrs":
        return ctx.tracegen_syscall_instrs(ev, lh)
    if chip == "memory_instrs":
        return ctx.tracegen_memory_instrs(ev, lh, blu)
    if chip == "misc_instrs":
        return ctx.tracegen_misc_instrs(ev, lh, blu)
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
    if chip == "global":
        return ctx.tracegen_global(ev, lh, blu)
    return ctx.tracegen_alu(chip, ev, lh, blu)


@pytest.mark.gpu
def test_gpu_cpu_and_program_tracegen_match_oracle(hip_ctx, oracle):
    from ziren_amd import lib
    assert lib.load().zkm_tracegen_cpu_width() == M.CPU_WIDTH
    for n, fixed in ((0, -1), (1, -1), (17, -1), (3000, -1), (5000, 14)):
        prog, rec, pv = M.run(n, seed=n + 2, halt=n > 0)
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_cpu(rec.cpu, prog, PC_BASE, SHARD, fixed, counts)
        blu = hip_ctx.byte_lookups()
        m = hip_ctx.tracegen_cpu(rec.cpu, prog, PC_BASE, SHARD, fixed, blu)
        assert (m.height, m.width) == want.shape and np.array_equal(m.to_host(), want), n
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert np.array_equal(F.from_monty(mults.to_host()), counts), n
        m.free(); mults.free(); blu.free()
        both = hip_ctx.tracegen_cpu_and_program(rec.cpu, prog, PC_BASE, SHARD, fixed)     # one upload of the events for both chips
        assert np.array_equal(both[0].to_host(), want)
        both[0].free()
        for which, born in ((0, hip_ctx.tracegen_program(prog, PC_BASE)), (1, hip_ctx.tracegen_program_mults(rec.cpu, len(prog), PC_BASE)),
                            (1, both[1])):
            want = oracle.tracegen_program(which, rec.cpu, prog, PC_BASE)
            assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), (n, which)
            born.free()
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_memory_instrs(rec.mem_instr, -1, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_memory_instrs(rec.mem_instr, -1, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), n
        assert np.array_equal(F.from_monty(mults.to_host()), counts), n
        born.free(); mults.free(); blu.free()
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_misc_instrs(rec.misc, -1, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_misc_instrs(rec.misc, -1, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), n
        assert np.array_equal(F.from_monty(mults.to_host()), counts), n
        born.free(); mults.free(); blu.free()
        want = oracle.tracegen_syscall_instrs(rec.syscall)
        born = hip_ctx.tracegen_syscall_instrs(rec.syscall)
        assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), n
        born.free()
        want = oracle.tracegen_memory_local(rec.memory_local)
        born = hip_ctx.tracegen_memory_local(rec.memory_local)
        assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), n
        born.free()
    prog, rec, pv = M.run(50, seed=1)
    with pytest.raises(lib.ZkmError, match="outside the program"):
        hip_ctx.tracegen_cpu(rec.cpu, prog[:10], PC_BASE, SHARD)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 5, 1023, 1024, 2500, 40000])
def test_gpu_global_tracegen_matches_oracle(hip_ctx, oracle, n):
    """zkm_tracegen_global == the restated GlobalChip::generate_trace, bit for bit: lift_x (square roots in the septic extension),
    the sign by direction, the range-check witness, and the running sum through one, two and three levels of the device scan
    (n + 1 <= 1024, <= 8192, above); the U16Range counts land in the shared byte-lookup table."""
    ev = random_global_events(n, 100 + n)
    want_counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    want = oracle.tracegen_global(ev, -1, want_counts)
    blu = hip_ctx.byte_lookups()
    m = hip_ctx.tracegen_global(ev, -1, blu)
    got = m.to_host()
    assert got.shape == want.shape and np.array_equal(got, want)
    mults = hip_ctx.tracegen_byte_mults(blu)
    assert np.array_equal(F.from_monty(mults.to_host()), want_counts)
    mults.free()
    m.free()
    if n in (5, 2500):
        # the events already in HBM (zkm_events_upload_async): nothing of them is read on the host, the u16 check included
        dev = hip_ctx.events_upload_async(ev)
        blu2 = hip_ctx.byte_lookups()
        m2 = hip_ctx.tracegen_global(dev, -1, blu2)
        assert np.array_equal(m2.to_host(), want)
        m2.free(); blu2.free(); dev.free()
    if n == 5:
        from ziren_amd import lib
        bad = ev.copy()
        bad["message"][2, 0] = 70000
        with pytest.raises(lib.ZkmError, match="not a u16"):
            hip_ctx.tracegen_global(bad, -1, blu)
        dev = hip_ctx.events_upload_async(bad)
        with pytest.raises(lib.ZkmError, match="not a u16"):
            hip_ctx.tracegen_global(dev, -1, blu)
        dev.free()
        with pytest.raises(lib.ZkmError, match="null byte lookups"):
            hip_ctx.tracegen_global(ev, -1, None)
    blu.free()


@pytest.mark.gpu
def test_gpu_global_digest_is_order_independent_at_full_size(hip_ctx, oracle):
    """2^20 messages (no oracle at this size): the final digest does not depend on the order of the events, both directions of the same
    messages cancel in the machine-level sum, every row's running sum hands over to the next row, and the byte table holds one U16Range
    count per event."""
    n = 1 << 20
    ev = random_global_events(n, 77)
    blu = hip_ctx.byte_lookups()
    a = hip_ctx.tracegen_global(ev, -1, blu)
    b = hip_ctx.tracegen_global(ev[np.random.default_rng(3).permutation(n)], -1, blu)
    c = hip_ctx.tracegen_global(complement(ev)[::-1].copy(), -1, blu)
    ta, tb, tc = a.to_host(), b.to_host(), c.to_host()
    assert np.array_equal(ta[-1, 85:], tb[-1, 85:]) and not np.array_equal(ta[n // 2, 85:], tb[n // 2, 85:])
    assert np.array_equal(ta[:-1, 85:], ta[1:, 64:78]) and not ta[:, 78:85].any()      # cumulative_sum -> next initial_digest; sum_checker 0 on real rows
    zero = F.to_monty(np.array(chips.SEPTIC_START_X + chips.SEPTIC_START_Y, dtype=np.uint64))
    assert np.array_equal(ta[0, 64:78], zero.astype(np.uint32))
    assert oracle.global_digest_sum([ta[-1, 85:], tc[-1, 85:], zero])[1] and not oracle.global_digest_sum([ta[-1, 85:], tb[-1, 85:], zero])[1]
    mults = hip_ctx.tracegen_byte_mults(blu)
    counts = F.from_monty(mults.to_host())
    assert int(counts[:, 8].sum()) == 3 * n and int(counts.sum()) == 3 * n
    # spot-check rows against the oracle: the point of an event does not depend on its neighbours
    pick = np.array([0, 1, 12345, n - 1])
    want = oracle.tracegen_global(ev[pick], -1)
    assert np.array_equal(ta[pick][:, :64], want[:len(pick), :64])
    for m in (a, b, c, mults):
        m.free()
    blu.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n_cycles", [40, 6000])
def test_gpu_coherent_shard_proof(hip_ctx, oracle, n_cycles):
    """Cpu + Program + the eleven instruction chips + Byte over one executed program: every trace and both preprocessed
    tables born on the device, proof bit-identical to the oracle's and accepted by the restated verifier; the Cpu chip's
    public values (start_pc, next_pc, execution_shard) are checked by its constraints."""
    from ziren_amd import prover, synth
    recs, work, byte, program, prog, pv = cpu_shard(oracle, n_cycles, seed=n_cycles)
    all_chips = recs + [byte, program]
    fri = abi.FriConfig(1, 84, 16)
    pvs = M.public_values(pv)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    pc_start = F.to_monty(PC_BASE)
    hp = prover.HipProver(all_chips, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    hp.specialize_quotient_kernels(recs + [byte, program])
    pk = hp.setup([hip_ctx.tracegen_byte_table(), hip_ctx.tracegen_program(prog, PC_BASE, program.log_height)], [0, 0], pc_start, igcs)
    opk = oracle.Pk([byte.prep_trace, program.prep_trace], [0, 0], pc_start, igcs, 1)
    assert np.array_equal(pk.commit, opk.commitment())
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    blu = hip_ctx.byte_lookups()
    born = [device_trace(hip_ctx, chip, ev, lh, blu, prog) for chip, ev, lh in work]
    born.append(hip_ctx.tracegen_byte_mults(blu))
    born.append(hip_ctx.tracegen_program_mults(work[0][1], len(prog), PC_BASE, program.log_height))
    proof = hp.prove_shard(pk, pvs, born, ch).copy()
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, all_chips, [c.trace for c in all_chips], pvs, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(opk, all_chips, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    # the machine-level check over global digests: this shard's (the last fourteen cells of the device-born Global trace = the proof's
    # global_cumulative_sum) and that of a shard holding the far side of every message sum to the zero digest
    gi = [c for c, _, _ in work].index("global")
    here = born[gi].to_host()[-1, 85:]
    assert np.array_equal(here, recs[gi].trace[-1, 85:]) and bytes(here) in bytes(proof)
    far = hip_ctx.tracegen_global(complement(work[gi][1])[::-1].copy(), -1, blu)
    zero = F.to_monty(np.array(chips.SEPTIC_START_X + chips.SEPTIC_START_Y, dtype=np.uint64))
    assert oracle.global_digest_sum([here, far.to_host()[-1, 85:], zero])[1] and not oracle.global_digest_sum([here, zero])[1]
    far.free()
    # a proof against a different claimed next_pc fails the Cpu chip's boundary constraint
    wrong = M.public_values(dict(pv, next_pc=pv["next_pc"] + 4))
    bad = hp.prove_shard(pk, wrong, born, start.copy()).copy()
    assert oracle.verify_shard(opk, all_chips, fri, synth.NUM_PV_ELTS, start.copy(), bad) != 0
    for m in born:
        m.free()
