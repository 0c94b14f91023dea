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
