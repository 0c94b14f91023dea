"""ziren_amd/codegen.py: the straight-line quotient kernels it writes for a constraint program.

ADVICE r03 (medium): a uniform-extension x column product that two different sums read (t = beta * col; assert t + perm[0]; assert t + perm[1])
was taken out of the deferred-form table by the first sum, and the second read an identifier that was never declared."""
import re

import numpy as np
import pytest

from ziren_amd import air, codegen, synth


def shared_term_program(main_width=6, perm_w=3, with_lookups=None):
    b = air.AirBuilder(main_width, 0, perm_w)
    loc, nxt = b.main()
    perm, _ = b.permutation()
    alpha, beta = b.permutation_randomness()
    t = beta * loc[0]                      # read by three sums below
    b.assert_zero(t + perm[0])
    b.assert_zero(t + perm[1])
    u = alpha * loc[1] + beta * loc[2]     # a two-term form, read twice, once merged into a longer form
    b.assert_zero(u + perm[0])
    b.assert_zero((u + t) + perm[1] * loc[3])
    b.assert_zero(t * perm[0] - perm[1])   # and read directly by a product
    b.when_transition().assert_eq(nxt[4], loc[4] + loc[5])
    if with_lookups is not None:
        sends, receives, batch = with_lookups
        air.eval_permutation_constraints(b, sends, receives, batch, False)
    return b.assemble()


def _check_defined_before_use(src):
    body = src[src.index("kb::FoldAcc acc"):]
    defined = set()
    for line in body.splitlines():
        for stmt in line.split(";"):
            m = re.search(r"const (?:uint32_t|kb::E4) ([be]\d+) =", stmt)
            rhs = stmt[stmt.index("=") + 1:] if "=" in stmt else stmt
            for name in re.findall(r"(?<![\w.])([be]\d+)\b", rhs if m else stmt):
                if m and name == m.group(1) and name not in rhs:
                    continue
                assert name in defined or (m and name == m.group(1)), f"{name} is read before it is declared in: {stmt.strip()[:160]}"
            if m:
                defined.add(m.group(1))
    return defined


def test_a_term_read_by_two_sums_is_declared_once_and_before_both():
    prog = shared_term_program()
    src = codegen.emit_source(prog)
    defined = _check_defined_before_use(src)
    assert len(defined) > 8
    for name in set(re.findall(r"const kb::E4 (e\d+) =", src)):
        assert len(re.findall(rf"const kb::E4 {name} =", src)) == 1


def test_every_recorded_core_chip_still_generates_well_formed_source():
    from ziren_amd import chips, events
    progs = [chips.record_chip(c, 10).program for c in sorted(events.CHIP_NAMES)] + [chips.record_cpu_chip(10).program, chips.record_global_chip(10).program,
                                                                                    chips.record_divrem_chip(10).program, chips.record_memory_local_chip(10).program]
    for p in progs:
        _check_defined_before_use(codegen.emit_source(p))


def test_the_shared_term_kernel_compiles_for_gfx950():
    """hipcc cross-compiles the generated kernel (no GPU needed): before the fix this failed with an undeclared identifier."""
    co = codegen.specialize(shared_term_program(), force=True)
    assert co is not None and len(co) > 1000


@pytest.mark.gpu
def test_gpu_shared_term_chip_specialised_equals_interpreter_and_oracle(oracle):
    """A chip whose program holds the shared terms (and its LogUp constraints): the proof made with the generated kernel, the proof made
    with the bytecode interpreter and the oracle's are the same words. (The trace does not satisfy the extra constraints — the quotient
    is then not a polynomial and no verifier would accept the proof — but all three compute the same function of the same inputs.)"""
    from ziren_amd import abi, prover
    sh = synth.syn_shard(8)
    c = sh.chips[1]
    c.program = shared_term_program(c.main_width, c.perm_ext_width, (c.sends, c.receives, 1 << c.log_quotient_degree))
    c.num_constraints = int(c.program[2])
    fri = abi.FriConfig(1, 8, 4)
    proofs = []
    for specialize in (False, True):
        ctx = prover.Context(0)
        try:
            hp = prover.HipProver(sh.chips, fri, synth.NUM_PV_ELTS, ctx=ctx, specialize=specialize)
            pk = hp.setup([], [], sh.pc_start, sh.initial_global_cumulative_sum)
            ch = prover.new_challenger()
            pk.observe_into(ch)
            tr = hp.upload_traces([x.trace for x in sh.chips])
            proofs.append(hp.prove_shard(pk, sh.public_values, tr, ch).copy())
            for t in tr:
                t.free()
            pk.free()
        finally:
            ctx.close()
    assert np.array_equal(proofs[0], proofs[1])
    opk = oracle.Pk([], [], sh.pc_start, sh.initial_global_cumulative_sum, 1)
    och = oracle.new_challenger()
    opk.observe_into(och)
    oproof, _ = oracle.prove_shard(opk, sh.chips, [x.trace for x in sh.chips], sh.public_values, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proofs[1], oproof)


# ---- permutation-trace kernels -----------------------------------------------------------------------------------------------------------

def test_perm_source_loads_every_column_once_and_parses_the_blob_back():
    from ziren_amd import chips
    c = chips.record_cpu_chip(10)
    n_sends, lookups = codegen.parse_lookups(c.lookups_blob)
    assert n_sends == len(c.sends) and len(lookups) == len(c.sends) + len(c.receives)
    assert np.array_equal(air.encode_lookups(c.sends, c.receives), c.lookups_blob)
    src = codegen.emit_perm_source(c.lookups_blob, c.log_quotient_degree)
    loads = re.findall(r"const uint32_t ([mp]\d+) = a\.(?:main|prep)\[", src)
    assert len(loads) == len(set(loads)) > 10
    assert src.count("kb::inv_batch<") == -(-(c.perm_ext_width - 1) // codegen.P_GROUP)      # one base-field inversion per P_GROUP fraction columns
    assert codegen.specialize_perm(chips.record_keccak_sponge_chip(10).lookups_blob, 1) is None      # too many lookups: the generic kernel


def test_perm_kernels_compile_for_gfx950():
    from ziren_amd import chips
    for c in (chips.record_cpu_chip(10), chips.record_global_chip(10), chips.record_byte_chip(), synth.syn_shard(8, with_prep=True, with_trace=False).chips[0]):
        co = codegen.specialize_perm(c.lookups_blob, c.log_quotient_degree)
        assert co is not None and len(co) > 1000


@pytest.mark.gpu
def test_gpu_generated_permutation_kernels_equal_the_generic_one_and_the_oracle(oracle):
    """zkm_permutation_trace of recorded core chips on executor traces and of synthetic chips (with preprocessed columns, batches of 2 and 4):
    the generic kernel (a context with nothing registered), the generated kernel (registered) and the oracle give the same cells and sums."""
    from ziren_amd import chips as CH, field as F, miniexec as M, prover
    import machine_lib as ML
    rng = np.random.default_rng(77)
    challenge = lambda: [int(x) for x in F.to_monty(rng.integers(0, F.P, 4, dtype=np.uint64).astype(np.uint32))]      # noqa: E731
    cases = [(c, c.trace, c.prep_trace) for c in synth.syn_shard(11, with_prep=True).chips[:4]]
    cases += [(c, c.trace, c.prep_trace) for c in synth.edge_shard(9, lqd=2).chips[:3]]
    m = M.run_machine(1200, seed=5, shard_cycles=1 << 20)
    cs = ML.build_shard(ML.Oracle(oracle), m, 0)
    cs[-2].prep_trace = oracle.tracegen_byte_table()
    cs[-1].prep_trace = oracle.tracegen_program(0, m.shards[0].record.cpu, m.program, m.pc_base, ML.log2_rows(len(m.program)))
    cases += [(c, c.trace, c.prep_trace) for c in cs]
    plain, special = prover.Context(0), prover.Context(0)
    try:
        hp = prover.HipProver([c for c, _, _ in cases], __import__("ziren_amd.abi", fromlist=["abi"]).FriConfig(1, 8, 4), synth.NUM_PV_ELTS, ctx=special)
        hp.specialize_perm_kernels()
        checked = 0
        for chip, trace, prep in cases:
            if chip.prep_width and prep is None:
                continue
            alpha, beta = challenge(), challenge()
            want, want_sum = oracle.permutation_trace(chip, trace, prep, alpha, beta)
            for ctx in (plain, special):
                main_d = ctx.upload(trace)
                prep_d = ctx.upload(prep) if chip.prep_width else None
                got, got_sum = ctx.permutation_trace(chip, main_d, prep_d, alpha, beta)
                assert np.array_equal(got.to_host(), want), (chip.name, ctx is special)
                assert np.array_equal(got_sum, want_sum), chip.name
                got.free(); main_d.free()
                if prep_d is not None:
                    prep_d.free()
            checked += 1
        assert checked >= 12 and {"Cpu", "Global", "Byte", "Program"} <= {c.name for c, _, _ in cases}
    finally:
        plain.close()
        special.close()


def test_pair_row_loads_puts_a_columns_two_reads_together_and_loses_nothing():
    """codegen.pair_row_loads: every statement exactly once, every value declared before its use, every column that is read at both rows
    has its two loads adjacent, and the next row's running sum is formed as the loads come (no pile of hoisted values left waiting)."""
    from ziren_amd import chips
    for prog in (chips.record_cpu_chip(10).program, chips.record_divrem_chip(10).program, chips.record_global_chip(10).program, shared_term_program()):
        lines, meta = codegen._ssa_lines(np.asarray(prog, dtype=np.uint32))
        out, meta2 = codegen.pair_row_loads(lines, meta)
        assert sorted(out) == sorted(lines) and len(meta2) == len(out)
        seen = set()
        for ln, (v, uses) in zip(out, meta2):
            assert all(u in seen for u in uses), ln
            if v is not None:
                seen.add(v)
        keys = [codegen._load_key(ln) for ln in out]
        where = {k: i for i, k in enumerate(keys) if k is not None}
        pairs = [(i, where[(k[0], k[1], not k[2])]) for i, k in enumerate(keys) if k is not None and (k[0], k[1], not k[2]) in where]
        assert all(abs(i - j) == 1 for i, j in pairs)
        _check_defined_before_use("kb::FoldAcc acc;\n" + "\n".join(out))
        # the hoisted values do not pile up: at no point are more than a few values that come from hoisted `next` loads waiting for their reader
        last_use = {}
        for i, (v, uses) in enumerate(meta2):
            for u in uses:
                last_use[u] = i
        hoisted = [meta2[j][0] for i, j in pairs if j == i + 1 and keys[j][2]]
        if hoisted:
            defs = {v: i for i, (v, _) in enumerate(meta2) if v is not None}
            peak = max(sum(1 for h in hoisted if defs[h] <= t < last_use.get(h, defs[h])) for t in range(len(out)))
            n_perm = sum(1 for k in keys if k is not None and k[0] == "perm" and k[2])
            assert peak <= max(8, len(hoisted) - n_perm + 3), (peak, len(hoisted), n_perm)


def test_prefetch_order_moves_loads_up_and_loses_nothing():
    """codegen.prefetch_order: every statement of the program appears exactly once, no load moves down, every value is still declared
    before its first use, and group k + 1's loads sit in front of group k's first use."""
    from ziren_amd import chips
    for prog in (chips.record_cpu_chip(10).program, chips.record_divrem_chip(10).program, shared_term_program()):
        lines, _ = codegen._ssa_lines(np.asarray(prog, dtype=np.uint32))
        out = codegen.prefetch_order(lines, 4)
        body = [ln for ln in out if "sched_barrier" not in ln]
        assert sorted(body) == sorted(lines) and len(out) - len(body) >= 1
        pos_before = {ln: k for k, ln in enumerate(lines)}
        seen_non_load = 0
        for k, ln in enumerate(body):
            if codegen._is_load(ln):
                assert seen_non_load <= sum(1 for x in lines[:pos_before[ln]] if not codegen._is_load(x)), "a load moved down"
            else:
                seen_non_load += 1
        _check_defined_before_use("kb::FoldAcc acc;\n" + "\n".join(body))


def test_split_uniform_moves_every_derived_uniform_value_into_the_table_kernel_and_loses_nothing():
    """codegen.split_uniform: the kernel proper keeps every row statement and assert, declares every value before its use, computes no
    wave-uniform value itself (it reads each from the table once), and the table kernel computes — from leaves only — exactly the values
    it stores, one slot for each distinct expression, 16-byte slots for extension values, inside what the library allocates."""
    from ziren_amd import chips
    n_cut = 0
    for prog in (chips.record_cpu_chip(10).program, chips.record_divrem_chip(10).program, chips.record_global_chip(10).program,
                 chips.record_byte_chip().program, shared_term_program()):
        lines, meta, uniform = codegen._ssa_lines(np.asarray(prog, dtype=np.uint32), with_uniform=True)
        cut = codegen.split_uniform(lines, meta, uniform)
        if cut is None:      # a program without wave-uniform arithmetic: nothing to move
            assert not any(v is not None and v in uniform and uses for v, uses in meta if any(v in u for _, u in meta))
            continue
        n_cut += 1
        main, main_meta, prologue = cut
        row = [(ln, m) for ln, m in zip(lines, meta) if m[0] is None or m[0] not in uniform]
        assert [ln for ln, m in zip(main, main_meta) if m[0] is None or m[0] not in uniform] == [ln for ln, _ in row]      # nothing lost, same order
        seen = set()
        for ln, (v, uses) in zip(main, main_meta):
            assert all(u in seen for u in uses), ln
            if v is not None:
                seen.add(v)
            if v is not None and v in uniform:
                assert not uses and ("a.uniforms" in ln or not any(f in ln for f in ("kb::emul", "kb::mul(", "kb::eadd", "kb::escale"))), ln
        _check_defined_before_use("kb::FoldAcc acc;\n" + "\n".join(main))
        stores = prologue[-1]
        slots = [(int(off), "e") for off in re.findall(r"\*\(kb::E4\*\)\(a\.uniforms \+ (\d+)\)", stores)] + \
                [(int(off), "b") for off in re.findall(r"a\.uniforms\[(\d+)\] =", stores)]
        used = sorted(w for off, kind in slots for w in range(off, off + (4 if kind == "e" else 1)))
        assert used == list(range(len(used))) and len(used) <= codegen.UNIFORM_TABLE_WORDS and all(off % 4 == 0 for off, kind in slots if kind == "e")
        loaded = re.findall(r"a\.uniforms(?: \+ |\[)(\d+)", "\n".join(main))
        assert {int(x) for x in loaded} == {off for off, _ in slots}          # every slot is read; equal uniform expressions share one
        _check_defined_before_use("kb::FoldAcc acc;\n" + "\n".join(prologue[:-1]))
    assert n_cut >= 4
    src = codegen.emit_source(chips.record_cpu_chip(10).program)
    assert src.count("__global__") == 2 and codegen.UNIFORMS_KERNEL_NAME in src
