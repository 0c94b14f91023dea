"""Two chips of the recursion machine as recorded AIRs, with synthetic programs: BaseAlu and ExtAlu
(crates/recursion/core/src/chips/alu_base.rs, alu_ext.rs). The compress / shrink provers run the same
commit + open over these AIRs with the compressed FRI configurations.

A recursion chip splits in two: the *program* fixes, per instruction, the memory addresses, the opcode flags and the
write multiplicity (preprocessed trace, committed in the proving key), the *execution* supplies the values (main
trace). Both traces are the instruction / event records laid end to end, four per row, zero padded
(alu_base.rs:99-137, 204-222; alu_ext.rs likewise), so "trace generation" is a padded upload (zkm_tracegen_flat).
"""
from typing import List, Tuple

import numpy as np

from . import air, chips, field as F

ENTRIES_PER_ROW = 4                       # NUM_BASE_ALU_ENTRIES_PER_ROW / NUM_EXT_ALU_ENTRIES_PER_ROW
ADD, SUB, MUL, DIV = range(4)             # BaseAluOpcode / ExtAluOpcode (runtime/opcode.rs)
BASE_VALUE_COLS, EXT_VALUE_COLS = 3, 12   # BaseAluIo<F>, ExtAluIo<Block<F>>: out, in1, in2
ACCESS_COLS = 8                           # addrs.out, addrs.in1, addrs.in2, is_add, is_sub, is_mul, is_div, mult
W = 3                                     # X^4 = 3 (crates/stark/src/air/extension.rs:55-74)


def _ext_mul_expr(b, x, y):
    """BinomialExtension::mul (extension.rs:58-74), symbolic."""
    out = [b.const(0)] * 4
    for i in range(4):
        for j in range(4):
            if i + j >= 4:
                out[i + j - 4] = out[i + j - 4] + b.const(W) * x[i] * y[j]
            else:
                out[i + j] = out[i + j] + x[i] * y[j]
    return out


class _RecRec(chips._Rec):
    """ZKMRecursionAirBuilder's memory lookups (crates/recursion/core/src/builder.rs:19-71): address + a 4-word block."""

    def _mem(self, addr, block, mult):
        vals = [addr] + list(block)
        return air.Lookup([air.to_virtual_pair(v) for v in vals], air.to_virtual_pair(mult), air.KIND_MEMORY)

    def send_block(self, addr, block, mult):
        self.sends.append(self._mem(addr, block, mult))

    def receive_block(self, addr, block, mult):
        self.receives.append(self._mem(addr, block, mult))

    def send_single(self, addr, val, mult):
        self.send_block(addr, [val, 0, 0, 0], mult)

    def receive_single(self, addr, val, mult):
        self.receive_block(addr, [val, 0, 0, 0], mult)


def _record(ext: bool) -> _RecRec:
    vw = EXT_VALUE_COLS if ext else BASE_VALUE_COLS
    r = _RecRec(ENTRIES_PER_ROW * vw, ENTRIES_PER_ROW * ACCESS_COLS)
    b, l, p = r.b, r.local, r.prep
    for k in range(ENTRIES_PER_ROW):
        v = l[k * vw:(k + 1) * vw]
        a_out, a_in1, a_in2, is_add, is_sub, is_mul, is_div, mult = p[k * ACCESS_COLS:(k + 1) * ACCESS_COLS]
        is_real = is_add + is_sub + is_mul + is_div
        b.assert_bool(is_real)
        if ext:   # ExtAluChip::eval, alu_ext.rs:272-303
            out, in1, in2 = v[0:4], v[4:8], v[8:12]
            for e in range(4):
                b.when(is_add).assert_eq(in1[e] + in2[e], out[e])
            for e in range(4):
                b.when(is_sub).assert_eq(in1[e], in2[e] + out[e])
            prod = _ext_mul_expr(b, in1, in2)
            for e in range(4):
                b.when(is_mul).assert_eq(prod[e], out[e])
            prod = _ext_mul_expr(b, in2, out)
            for e in range(4):
                b.when(is_div).assert_eq(in1[e], prod[e])
            r.receive_block(a_in1, in1, is_real)
            r.receive_block(a_in2, in2, is_real)
            r.send_block(a_out, out, mult)
        else:     # BaseAluChip::eval, alu_base.rs:279-304
            out, in1, in2 = v
            b.when(is_add).assert_eq(in1 + in2, out)
            b.when(is_sub).assert_eq(in1, in2 + out)
            b.when(is_mul).assert_eq(out, in1 * in2)
            b.when(is_div).assert_eq(in2 * out, in1)
            r.receive_single(a_in1, in1, is_real)
            r.receive_single(a_in2, in2, is_real)
            r.send_single(a_out, out, mult)
    return r


def record_constraints(ext: bool) -> _RecRec:
    return _record(ext)


def record_chip(ext: bool, log_height: int, prep_index: int, lqd: int = 1) -> chips.RecordedChip:
    r = _record(ext)
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return chips.RecordedChip(name="ExtAlu" if ext else "BaseAlu", log_height=log_height, main_width=r.b.main_width,
                              prep_width=r.b.prep_width, prep_index=prep_index, log_quotient_degree=lqd, local_only=True,
                              sends=r.sends, receives=r.receives, program=program,
                              lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


CONST_MEM_ENTRIES_PER_ROW = 2   # NUM_CONST_MEM_ENTRIES_PER_ROW (chips/mem/constant.rs:15)
CONST_MEM_ENTRY_COLS = 6         # (Block<F> value, MemoryAccessCols { addr, mult })


def record_mem_const(log_height: int = 10, prep_index: int = 0, lqd: int = 1, constraints_only: bool = False):
    """MemoryConstChip (crates/recursion/core/src/chips/mem/constant.rs): the program's memory writes (and reads, with
    negated multiplicity) as a preprocessed table of (value block, address, multiplicity), two per row; the main trace is
    one unused column; `eval` sends every entry (:152-160)."""
    r = _RecRec(1, CONST_MEM_ENTRIES_PER_ROW * CONST_MEM_ENTRY_COLS)
    p = r.prep
    for k in range(CONST_MEM_ENTRIES_PER_ROW):
        e = p[k * CONST_MEM_ENTRY_COLS:(k + 1) * CONST_MEM_ENTRY_COLS]
        r.send_block(e[4], e[0:4], e[5])
    if constraints_only:
        return r
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return chips.RecordedChip(name="MemoryConst", log_height=log_height, main_width=1, prep_width=r.b.prep_width,
                              prep_index=prep_index, log_quotient_degree=lqd, local_only=True, sends=r.sends, receives=r.receives,
                              program=program, lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


VAR_MEM_ENTRIES_PER_ROW = 2      # NUM_VAR_MEM_ENTRIES_PER_ROW (chips/mem/variable.rs:16)
SELECT_PREP_COLS, SELECT_COLS = 8, 5   # (is_real, addrs {bit, out1, out2, in1, in2}, mult1, mult2); SelectIo {bit, out1, out2, in1, in2}


def _finish_rec(r, name, log_height, main_width, prep_index, lqd=1, local_only=True):
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return chips.RecordedChip(name=name, log_height=log_height, main_width=main_width, prep_width=r.b.prep_width, prep_index=prep_index,
                              log_quotient_degree=lqd, local_only=local_only, sends=r.sends, receives=r.receives, program=program,
                              lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_mem_var(log_height: int = 10, prep_index: int = 0, constraints_only: bool = False):
    """MemoryVar chip (crates/recursion/core/src/chips/mem/variable.rs): witnessed values — two 4-word blocks per row in the
    main trace — written to the addresses / with the multiplicities of the preprocessed trace (two (addr, mult) pairs per row);
    `eval` sends every entry (:157-168)."""
    r = _RecRec(4 * VAR_MEM_ENTRIES_PER_ROW, 2 * VAR_MEM_ENTRIES_PER_ROW)
    for k in range(VAR_MEM_ENTRIES_PER_ROW):
        r.send_block(r.prep[2 * k], r.local[4 * k:4 * k + 4], r.prep[2 * k + 1])
    return r if constraints_only else _finish_rec(r, "MemoryVar", log_height, 4 * VAR_MEM_ENTRIES_PER_ROW, prep_index)


def record_select(log_height: int = 10, prep_index: int = 0, constraints_only: bool = False):
    """Select chip (crates/recursion/core/src/chips/select.rs:232-257): out1 = bit ? in2 : in1, out2 = bit ? in1 : in2; one
    instruction per row; reads bit, in1, in2 (multiplicity is_real), writes out1 / out2 with their multiplicities."""
    r = _RecRec(SELECT_COLS, SELECT_PREP_COLS)
    p, l, b = r.prep, r.local, r.b
    is_real, a_bit, a_out1, a_out2, a_in1, a_in2, mult1, mult2 = (p[i] for i in range(8))
    bit, out1, out2, in1, in2 = (l[i] for i in range(5))
    r.receive_single(a_bit, bit, is_real)
    r.receive_single(a_in1, in1, is_real)
    r.receive_single(a_in2, in2, is_real)
    r.send_single(a_out1, out1, mult1)
    r.send_single(a_out2, out2, mult2)
    b.assert_eq(out1, bit * in2 + (1 - bit) * in1)
    b.assert_eq(out2, bit * in1 + (1 - bit) * in2)
    return r if constraints_only else _finish_rec(r, "Select", log_height, SELECT_COLS, prep_index)


POSEIDON2_WIDE_WIDTH, POSEIDON2_WIDE_PREP_WIDTH = 313, 49


def _poseidon2_constants():
    """Round constants and internal diagonal (canonical) from the tables the kernels use (csrc/poseidon2_constants.inc, pinned
    against the reference's dump in tests/test_oracle_pins.py)."""
    import os
    import re
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "poseidon2_constants.inc")).read()
    tabs = {m.group(1): [int(x) for x in re.findall(r"(\d+)u", m.group(2))]
            for m in re.finditer(r"static const uint32_t (\w+)\[\d+\](?:\[16\])? = \{(.*?)\};", txt, re.S)}
    rc = F.from_monty(np.array(tabs["ZKM_RC_16_30_MONTY"], dtype=np.uint32)).reshape(30, 16)
    diag = F.from_monty(np.array(tabs["ZKM_INTERNAL_DIAG_16_MONTY"], dtype=np.uint32))
    return [[int(x) for x in row] for row in rc], [int(x) for x in diag]


def _external_layer(s):
    """mds_light_permutation (chips/poseidon2_wide/mod.rs:45-71) over ints or recorded expressions."""
    s = list(s)
    for j in range(0, 16, 4):
        x = s[j:j + 4]
        t01, t23 = x[0] + x[1], x[2] + x[3]
        t0123 = t01 + t23
        t01123, t01233 = t0123 + x[1], t0123 + x[3]
        s[j + 3] = t01233 + x[0] * 2
        s[j + 1] = t01123 + x[2] * 2
        s[j] = t01123 + t01
        s[j + 2] = t01233 + t23
    sums = [s[k] + s[4 + k] + s[8 + k] + s[12 + k] for k in range(4)]
    return [s[j] + sums[j % 4] for j in range(16)]


def _internal_layer(s, diag):
    total = s[0]
    for x in s[1:]:
        total = total + x
    return [s[i] * diag[i] + total for i in range(16)]


def poseidon2_permute(state):
    """The width-16 KoalaBear Poseidon2 permutation over canonical ints (the program generator's executor)."""
    rc, diag = _poseidon2_constants()
    P = F.P
    s = [x % P for x in _external_layer(state)]
    for r in range(4):
        s = [x % P for x in _external_layer([pow(s[i] + rc[r][i], 3, P) for i in range(16)])]
    for r in range(13):
        s[0] = pow(s[0] + rc[4 + r][0], 3, P)
        s = [x % P for x in _internal_layer(s, diag)]
    for r in range(4, 8):
        s = [x % P for x in _external_layer([pow(s[i] + rc[13 + r][i], 3, P) for i in range(16)])]
    return s


def record_poseidon2_wide(log_height: int = 10, prep_index: int = 0, constraints_only: bool = False):
    """Poseidon2WideChip<3>::eval (chips/poseidon2_wide/air.rs:34-177): the degree-normalising dummy constraint, sixteen
    input reads (multiplicity is_real_neg = -1) and sixteen output writes, eight external rounds with their S-box columns
    (x^3 as a degree-3 constraint, the linear layer folded into the next state's constraint) and the thirteen internal
    rounds (lane 0 witnessed per round, the other lanes carried as linear expressions)."""
    rc, diag = _poseidon2_constants()
    r = _RecRec(POSEIDON2_WIDE_WIDTH, POSEIDON2_WIDE_PREP_WIDTH)
    l, p, b = r.local, r.prep, r.b
    EXT_STATE, INT_STATE, INT_S0, OUTPUT, EXT_SBOX, INT_SBOX = 0, 128, 144, 156, 172, 300
    ext_state = [l[EXT_STATE + 16 * k:EXT_STATE + 16 * k + 16] for k in range(8)]
    int_state, s0, output = l[INT_STATE:INT_STATE + 16], l[INT_S0:INT_S0 + 12], l[OUTPUT:OUTPUT + 16]
    ext_sbox = [l[EXT_SBOX + 16 * k:EXT_SBOX + 16 * k + 16] for k in range(8)]
    int_sbox = l[INT_SBOX:INT_SBOX + 13]
    x0 = ext_state[0][0]
    b.assert_eq(x0 * x0 * x0, x0 * x0 * x0)
    for i in range(16):
        r.send_single(p[i], ext_state[0][i], p[48])
    for i in range(16):
        r.send_single(p[16 + 2 * i], output[i], p[17 + 2 * i])
    for rd in range(8):
        state = list(ext_state[rd])
        if rd == 0:
            state = _external_layer(state)
        rnd = rd if rd < 4 else rd + 13
        for i in range(16):
            add_rc = state[i] + rc[rnd][i]
            b.assert_eq(ext_sbox[rd][i], add_rc * add_rc * add_rc)
        state = _external_layer(list(ext_sbox[rd]))
        nxt = int_state if rd == 3 else output if rd == 7 else ext_state[rd + 1]
        for i in range(16):
            b.assert_eq(nxt[i], state[i])
    state = list(int_state)     # the internal rounds sit between external rounds 3 and 4; eval lists them after all external rounds
    for rd in range(13):
        add_rc = (state[0] if rd == 0 else s0[rd - 1]) + rc[4 + rd][0]
        b.assert_eq(int_sbox[rd], add_rc * add_rc * add_rc)
        state[0] = int_sbox[rd]
        state = _internal_layer(state, diag)
        if rd < 12:
            b.assert_eq(s0[rd], state[0])
    for i in range(16):
        b.assert_eq(ext_state[4][i], state[i])
    return r if constraints_only else _finish_rec(r, "Poseidon2WideDeg3", log_height, POSEIDON2_WIDE_WIDTH, prep_index)


EXP_REVERSE_BITS_COLS, EXP_REVERSE_BITS_PREP_COLS = 7, 10
BATCH_FRI_COLS, BATCH_FRI_PREP_COLS = 13, 6
PUBLIC_VALUES_PREP_COLS, PUBLIC_VALUES_LOG_HEIGHT, DIGEST_SIZE = 10, 4, 8
PV_DIGEST_POS = 223     # RecursionPublicValues::digest (crates/recursion/core/src/air/public_values.rs:79-145): the last eight of 231 words


def record_exp_reverse_bits(log_height: int = 10, prep_index: int = 0, constraints_only: bool = False):
    """ExpReverseBitsLenChip<3>::eval (chips/exp_reverse_bits.rs:345-405): one row per exponent bit, accum' = accum^2 * (bit ? x : 1);
    prep = x_mem, exponent_mem, result_mem (addr, mult each), iteration_num, is_first, is_last, is_real; main = x, current_bit,
    prev_accum_squared, prev_accum_squared_times_multiplier, accum, accum_squared, multiplier."""
    r = _RecRec(EXP_REVERSE_BITS_COLS, EXP_REVERSE_BITS_PREP_COLS)
    l, n, p, b = r.local, r.next, r.prep, r.b
    pn = b.preprocessed()[1]
    x, bit, prev_sq, prev_sq_mul, accum, accum_sq, mult = (l[i] for i in range(7))
    is_first, is_last, is_real = p[7], p[8], p[9]
    r.send_single(p[0], x, p[1])
    b.when_transition().when(pn[9]).when_not(is_last).assert_eq(x, n[0])
    r.send_single(p[2], bit, p[3])
    b.when(is_first).assert_eq(accum, mult)
    b.when(is_real).when(bit).assert_eq(mult, x)
    b.when(is_real).when_not(bit).assert_eq(mult, 1)
    b.when(is_real).assert_eq(prev_sq_mul, prev_sq * mult)
    b.when(is_real).when_not(is_first).assert_eq(accum, prev_sq_mul)
    b.when(is_real).assert_eq(accum_sq, accum * accum)
    b.when_transition().when(pn[9]).when_not(is_last).assert_eq(n[2], accum_sq)
    r.send_single(p[4], accum, p[5])
    return r if constraints_only else _finish_rec(r, "ExpReverseBitsLen", log_height, EXP_REVERSE_BITS_COLS, prep_index, local_only=False)


def record_batch_fri(log_height: int = 10, prep_index: int = 0, constraints_only: bool = False, degree: int = 3):
    """BatchFRIChip<3>::eval (chips/batch_fri.rs:289-350): acc = sum over an instruction's rows of alpha_pow * (p_at_z - p_at_x) in the
    extension; prep = is_real, is_end, acc / alpha_pow / p_at_z / p_at_x addresses; main = acc(4), alpha_pow(4), p_at_z(4), p_at_x."""
    r = _RecRec(BATCH_FRI_COLS, BATCH_FRI_PREP_COLS)
    l, n, p, b = r.local, r.next, r.prep, r.b
    is_real, is_end = p[0], p[1]
    b.assert_eq(_pow_expr(is_real, degree), _pow_expr(is_real, degree))
    r.receive_block(p[3], l[4:8], is_real)
    r.receive_block(p[4], l[8:12], is_real)
    r.receive_single(p[5], l[12], is_real)
    r.send_block(p[2], l[0:4], is_end)

    def term(row):    # alpha_pow * (p_at_z - p_at_x) over F[X]/(X^4 - 3)
        return _ext_mul_expr(b, row[4:8], [row[8] - row[12], row[9], row[10], row[11]])

    for i, t in enumerate(term(l)):
        b.when_first_row().assert_eq(l[i], t)
    tn = term(n)
    for i in range(4):
        b.when_transition().when(is_end).assert_eq(n[i], tn[i])
    for i in range(4):
        b.when_transition().when_not(is_end).assert_eq(n[i], l[i] + tn[i])
    lqd = max(1, (degree - 2).bit_length())
    return r if constraints_only else _finish_rec(r, "BatchFRI", log_height, BATCH_FRI_COLS, prep_index, lqd=lqd, local_only=False)


FRI_FOLD_COLS, FRI_FOLD_PREP_COLS = 33, 20


def record_fri_fold(log_height: int = 10, prep_index: int = 0, degree: int = 3, constraints_only: bool = False):
    """FriFoldChip<DEGREE>::eval (chips/fri_fold.rs:362-463; part of the reference's all-chips test machines only): per row
    alpha_pow' = alpha_pow * alpha and (ro' - ro) * (x - z) = (p(x) - p(z)) * alpha_pow over the extension; x, z, alpha are shared by the
    rows of one instruction. prep = is_first, nine (addr, mult) pairs [z, alpha, x, alpha_pow_in, ro_in, p_at_x, p_at_z, ro_out,
    alpha_pow_out], is_real; main = z(4), alpha(4), x, p_at_x(4), p_at_z(4), alpha_pow_in(4), ro_in(4), alpha_pow_out(4), ro_out(4)."""
    r = _RecRec(FRI_FOLD_COLS, FRI_FOLD_PREP_COLS)
    l, n, p, b = r.local, r.next, r.prep, r.b
    pn = b.preprocessed()[1]
    z, alpha, x, p_at_x, p_at_z, ap_in, ro_in, ap_out, ro_out = l[0:4], l[4:8], l[8], l[9:13], l[13:17], l[17:21], l[21:25], l[25:29], l[29:33]
    mem = lambda k: (p[1 + 2 * k], p[2 + 2 * k])   # noqa: E731
    b.assert_eq(_pow_expr(p[19], degree), _pow_expr(p[19], degree))
    cont = lambda: b.when_transition().when(pn[19]).when_not(pn[0])   # noqa: E731  (the next row continues this instruction)
    r.send_single(mem(2)[0], x, mem(2)[1])
    cont().assert_eq(x, n[8])
    r.send_block(mem(0)[0], z, mem(0)[1])
    for i in range(4):
        cont().assert_eq(z[i], n[i])
    r.send_block(mem(1)[0], alpha, mem(1)[1])
    for i in range(4):
        cont().assert_eq(alpha[i], n[4 + i])
    r.send_block(mem(3)[0], ap_in, mem(3)[1])
    r.send_block(mem(4)[0], ro_in, mem(4)[1])
    r.send_block(mem(6)[0], p_at_z, mem(6)[1])
    r.send_block(mem(5)[0], p_at_x, mem(5)[1])
    r.send_block(mem(8)[0], ap_out, mem(8)[1])
    r.send_block(mem(7)[0], ro_out, mem(7)[1])
    for got, want in zip(_ext_mul_expr(b, ap_in, alpha), ap_out):
        b.assert_eq(got, want)
    lhs = _ext_mul_expr(b, [ro_out[i] - ro_in[i] for i in range(4)], [x - z[0], 0 - z[1], 0 - z[2], 0 - z[3]])
    rhs = _ext_mul_expr(b, [p_at_x[i] - p_at_z[i] for i in range(4)], ap_in)
    for a_, b_ in zip(lhs, rhs):
        b.assert_eq(a_, b_)
    lqd = max(1, (degree - 2).bit_length())
    return r if constraints_only else _finish_rec(r, "FriFold", log_height, FRI_FOLD_COLS, prep_index, lqd=lqd, local_only=False)


def _ext_inv(a):
    """Inverse in F[X]/(X^4 - 3) by solving the 4 x 4 system of multiplication by `a` (a handful of elements: speed is irrelevant)."""
    P = F.P
    m = [[0] * 5 for _ in range(4)]
    for j in range(4):          # column j: a * X^j
        for i in range(4):
            k, coef = (i + j) % 4, (W if i + j >= 4 else 1)
            m[k][j] = (m[k][j] + a[i] * coef) % P
    m[0][4] = 1
    for c in range(4):
        piv = next(r_ for r_ in range(c, 4) if m[r_][c])
        m[c], m[piv] = m[piv], m[c]
        inv = pow(m[c][c], P - 2, P)
        m[c] = [v * inv % P for v in m[c]]
        for r_ in range(4):
            if r_ != c and m[r_][c]:
                f = m[r_][c]
                m[r_] = [(v - f * w) % P for v, w in zip(m[r_], m[c])]
    return [m[i][4] for i in range(4)]


SKINNY_WIDTH, SKINNY_PREP_WIDTH, SKINNY_ROWS = 28, 51, 11    # state_var[16] + internal_rounds_s0[12]; 16 x (addr, mult) + 3 flags + 16 constants


def _pow_expr(x, n):
    out = x
    for _ in range(n - 1):
        out = out * x
    return out


def record_poseidon2_skinny(log_height: int = 10, prep_index: int = 0, degree: int = 9, constraints_only: bool = False):
    """Poseidon2SkinnyChip<DEGREE>::eval (chips/poseidon2_skinny/air.rs:23-160), the hash chip of the wrap machine: eleven rows per
    permutation (input, four external rounds, one row holding all thirteen internal rounds, four external rounds, output), the state
    in 16 columns plus lane 0 after each internal round; round type and constants are preprocessed. Nothing but the state is
    witnessed, so the constraints reach degree 5 (external) and the chip's quotient degree follows the machine's DEGREE (the dummy
    constraint x^DEGREE = x^DEGREE): 9 -> eight quotient chunks."""
    _, diag = _poseidon2_constants()
    r = _RecRec(SKINNY_WIDTH, SKINNY_PREP_WIDTH)
    l, n, p, b = r.local, r.next, r.prep, r.b
    state, s0, nxt = l[0:16], l[16:28], n[0:16]
    is_input, is_external, is_internal, rc = p[32], p[33], p[34], p[35:51]
    b.assert_eq(_pow_expr(state[0], degree), _pow_expr(state[0], degree))
    for i in range(16):
        r.send_single(p[2 * i], state[i], p[2 * i + 1])
    lin = _external_layer(list(state))
    for i in range(16):
        b.when_transition().when(is_input).assert_eq(nxt[i], lin[i])
    ext = _external_layer([_pow_expr(state[i] + rc[i], 3) for i in range(16)])
    for i in range(16):
        b.when_transition().when(is_external).assert_eq(nxt[i], ext[i])
    st = list(state)
    for rd in range(13):
        add_rc = (st[0] if rd == 0 else s0[rd - 1]) + rc[rd]
        st[0] = _pow_expr(add_rc, 3)
        st = _internal_layer(st, diag)
        if rd < 12:
            b.when(is_internal).assert_eq(s0[rd], st[0])
    for i in range(16):
        b.when(is_internal).assert_eq(nxt[i], st[i])
    lqd = max(1, (degree - 2).bit_length())     # log2_ceil(degree - 1)
    return r if constraints_only else _finish_rec(r, f"Poseidon2SkinnyDeg{degree}", log_height, SKINNY_WIDTH, prep_index, lqd=lqd, local_only=False)


def poseidon2_skinny_prep(instrs) -> np.ndarray:
    """generate_preprocessed_trace (chips/poseidon2_skinny/trace.rs:177-243): instrs = [(input addrs[16], output addrs[16], output
    mults[16])]; eleven rows of 51 canonical words each."""
    rc, _ = _poseidon2_constants()
    rows = np.zeros((len(instrs) * SKINNY_ROWS, SKINNY_PREP_WIDTH), dtype=np.uint64)
    for k, (ins, outs, mults) in enumerate(instrs):
        blk = rows[k * SKINNY_ROWS:(k + 1) * SKINNY_ROWS]
        blk[0, 32], blk[5, 34] = 1, 1
        for i in (1, 2, 3, 4, 6, 7, 8, 9):
            blk[i, 33] = 1
            blk[i, 35:51] = rc[i - 1 if i < 5 else i - 1 + 12]
        blk[5, 35:51] = [rc[4 + j][0] for j in range(16)]
        blk[0, 0:32:2], blk[0, 1:32:2] = ins, F.P - 1
        blk[10, 0:32:2], blk[10, 1:32:2] = outs, mults
    return rows


def record_public_values(prep_index: int = 0, constraints_only: bool = False):
    """PublicValuesChip::eval (chips/public_values.rs:274-291): sixteen rows, eight real: row i reads digest element i from memory
    (multiplicity -1) and ties it to public value digest[i]; prep = pv_idx[8] one-hot, (addr, mult); main = pv_element."""
    r = _RecRec(1, PUBLIC_VALUES_PREP_COLS)
    l, p, b = r.local, r.prep, r.b
    r.send_single(p[8], l[0], p[9])
    for i in range(DIGEST_SIZE):
        b.when(p[i]).assert_eq(b.public_values(PV_DIGEST_POS + i), l[0])
    return r if constraints_only else _finish_rec(r, "PublicValues", PUBLIC_VALUES_LOG_HEIGHT, 1, prep_index, local_only=False)


def balanced_program(n_base: int, n_ext: int, n_const: int = 64, seed: int = 1, n_var: int = 0, n_select: int = 0, n_poseidon2: int = 0,
                     permute_batch=None, n_exp: int = 0, n_batch_fri: int = 0, commit_public_values: bool = False, n_fri_fold: int = 0,
                     inputs=None):
    """A synthetic recursion program whose memory lookups balance exactly, as a real one's do: constants are written by
    MemoryConst entries, every ALU instruction reads two earlier values (constants or earlier results of its own field)
    and writes one, and each write's multiplicity is the number of later reads; a few results are read back by
    MemoryConst `Read` entries (negative multiplicity). Returns a dict of flat Montgomery word arrays:
    base_instrs, base_events, ext_instrs, ext_events, mem_entries (6 words each), plus the counts.
    `inputs` (canonical field words, e.g. the commitments of the core shard proofs a compress program verifies) are witnessed through
    MemoryVar, absorbed eight at a time by a chain of Poseidon2 permutations (overwrite-mode sponge from the zero state), and the
    chain's final eight words become the committed public-values digest: the shard's public output is bound to those inputs the way
    a reduce program binds its output to the proofs it consumed (crates/prover/src/lib.rs:617-641)."""
    rng = np.random.default_rng(seed)
    P = F.P
    addr = [1]

    def new_addr():
        addr[0] += 1
        return addr[0]

    def ext_mul(x, y):
        out = [0, 0, 0, 0]
        for i in range(4):
            for j in range(4):
                t = x[i] * y[j] % P
                if i + j >= 4:
                    out[i + j - 4] = (out[i + j - 4] + W * t) % P
                else:
                    out[i + j] = (out[i + j] + t) % P
        return out

    entries = {}   # addr -> {"val": [4], "reads": int, "kind": "const" | "base" | "ext", "instr": index}
    pools = {"base": [], "ext": []}
    for i in range(n_const):
        a = new_addr()
        ext = i % 2 == 1
        v = [int(x) for x in rng.integers(0, P, 4)] if ext else [int(rng.integers(0, P)), 0, 0, 0]
        entries[a] = {"val": v, "reads": 0, "kind": "const"}
        pools["ext" if ext else "base"].append(a)
    # witnessed values (MemoryVar): base elements, every fourth one a bit; then the selects, whose outputs join the base pool
    bits, var_rows, select_rows = [], [], []
    for i in range(n_var):
        a = new_addr()
        v = [int(rng.integers(0, 2)) if i % 4 == 0 else int(rng.integers(0, P)), 0, 0, 0]
        entries[a] = {"val": v, "reads": 0, "kind": "var"}
        var_rows.append(a)
        (bits if i % 4 == 0 else pools["base"]).append(a)
    for _ in range(n_select if bits else 0):
        ab = bits[int(rng.integers(0, len(bits)))]
        a1, a2 = (pools["base"][int(rng.integers(0, len(pools["base"])))] for _ in range(2))
        bit, x, y = entries[ab]["val"][0], entries[a1]["val"][0], entries[a2]["val"][0]
        o1, o2 = (y, x) if bit else (x, y)
        ao1, ao2 = new_addr(), new_addr()
        for a in (ab, a1, a2):
            entries[a]["reads"] += 1
        entries[ao1] = {"val": [o1, 0, 0, 0], "reads": 0, "kind": "base"}
        entries[ao2] = {"val": [o2, 0, 0, 0], "reads": 0, "kind": "base"}
        pools["base"] += [ao1, ao2]
        select_rows.append((ab, ao1, ao2, a1, a2, bit, o1, o2, x, y))
    poseidon_rows = []
    input_digest = None
    if inputs is not None and len(inputs):
        az = new_addr()
        entries[az] = {"val": [0, 0, 0, 0], "reads": 0, "kind": "const"}
        words = [int(w) % P for w in inputs] + [0] * (-len(inputs) % 8)
        in_addrs = []
        for w in words:
            a = new_addr()
            entries[a] = {"val": [w, 0, 0, 0], "reads": 0, "kind": "var"}
            var_rows.append(a)
            in_addrs.append(a)
        state = [az] * 16
        for k in range(0, len(in_addrs), 8):
            ins = in_addrs[k:k + 8] + state[8:]
            out = poseidon2_permute([entries[a]["val"][0] for a in ins])
            outs = [new_addr() for _ in range(16)]
            for a in ins:
                entries[a]["reads"] += 1
            for a, v in zip(outs, out):
                entries[a] = {"val": [v, 0, 0, 0], "reads": 0, "kind": "base"}
            poseidon_rows.append((ins, outs))
            state = outs
        input_digest = state[:8]
        pools["base"] += state
    batch_out = None
    if permute_batch is not None and n_poseidon2:   # large programs: every hash reads values that exist already, one batched permutation
        pool0 = np.array(pools["base"], dtype=np.int64)     # (permute_batch: (n, 16) canonical -> (n, 16) canonical, e.g. on the device)
        picks = pool0[rng.integers(0, len(pool0), (n_poseidon2, 16))]
        vals = np.array([entries[a]["val"][0] for a in pool0], dtype=np.uint64)[np.searchsorted(pool0, picks)]
        batch_out = np.asarray(permute_batch(vals), dtype=np.uint64)
    for k in range(n_poseidon2):
        if batch_out is not None:
            ins = [int(a) for a in picks[k]]
            out = [int(v) for v in batch_out[k]]
        else:
            ins = [pools["base"][int(rng.integers(0, len(pools["base"])))] for _ in range(16)]
            out = poseidon2_permute([entries[a]["val"][0] for a in ins])
        outs = [new_addr() for _ in range(16)]
        for a in ins:
            entries[a]["reads"] += 1
        for a, v in zip(outs, out):
            entries[a] = {"val": [v, 0, 0, 0], "reads": 0, "kind": "base"}
        pools["base"] += outs
        poseidon_rows.append((ins, outs))
    # ExpReverseBitsLen: result = x^(bit-reversed exponent), one row per bit (accum' = accum^2 * (bit ? x : 1))
    exp_prep, exp_main = [], []
    for _ in range(n_exp if bits else 0):
        ax = pools["base"][int(rng.integers(0, len(pools["base"])))]
        xv = entries[ax]["val"][0]
        nbits = int(rng.integers(1, 32))
        abits = [bits[int(rng.integers(0, len(bits)))] for _ in range(nbits)]
        ares = new_addr()
        entries[ax]["reads"] += 1
        accum, prev = 1, 1
        rows_here = []
        for i, ab in enumerate(abits):
            entries[ab]["reads"] += 1
            bv = entries[ab]["val"][0]
            m = xv if bv else 1
            prev_sq = prev * prev % P
            accum = prev_sq * m % P
            exp_main.append([xv, bv, prev_sq, prev_sq * m % P, accum, accum * accum % P, m])
            rows_here.append([ax, (P - 1) if i == 0 else 0, ab, P - 1, ares, None, i, int(i == 0), int(i == nbits - 1), 1])
            prev = accum
        entries[ares] = {"val": [accum, 0, 0, 0], "reads": 0, "kind": "base"}
        pools["base"].append(ares)
        exp_prep.append((ares, rows_here))
    # BatchFRI: acc = sum_k alpha_pow_k * (p_at_z_k - p_at_x_k); one row per k, the accumulator written on the last one
    fri_prep, fri_main = [], []
    for _ in range(n_batch_fri):
        k = int(rng.integers(1, 9))
        aacc = new_addr()
        acc = [0, 0, 0, 0]
        rows_here = []
        for i in range(k):
            aal, apz = (pools["ext"][int(rng.integers(0, len(pools["ext"])))] for _ in range(2))
            apx = pools["base"][int(rng.integers(0, len(pools["base"])))]
            for a in (aal, apz, apx):
                entries[a]["reads"] += 1
            al, pz, px = entries[aal]["val"], entries[apz]["val"], entries[apx]["val"][0]
            t = ext_mul(al, [(pz[0] - px) % P, pz[1], pz[2], pz[3]])
            acc = [(acc[e] + t[e]) % P for e in range(4)]
            fri_main.append(acc + al + pz + [px])
            rows_here.append([1, int(i == k - 1), aacc, aal, apz, apx])
        entries[aacc] = {"val": acc, "reads": 1, "kind": "fri_acc"}   # written with multiplicity 1 (acc_mult, batch_fri.rs:103): read exactly once
        fri_prep += rows_here
    # FriFold: rows of one instruction share x, z, alpha; per row alpha_pow' = alpha_pow * alpha, ro' = ro + (p(x) - p(z)) * alpha_pow / (x - z)
    ff_prep, ff_main, ff_outs = [], [], []
    for _ in range(n_fri_fold):
        k = int(rng.integers(1, 6))
        az, aal = (pools["ext"][int(rng.integers(0, len(pools["ext"])))] for _ in range(2))
        ax = pools["base"][int(rng.integers(0, len(pools["base"])))]
        for a in (az, aal, ax):
            entries[a]["reads"] += 1
        zv, alv, xv = entries[az]["val"], entries[aal]["val"], entries[ax]["val"][0]
        inv_xz = _ext_inv([(xv - zv[0]) % P, (-zv[1]) % P, (-zv[2]) % P, (-zv[3]) % P])
        for i in range(k):
            aap, aro, apx, apz = (pools["ext"][int(rng.integers(0, len(pools["ext"])))] for _ in range(4))
            for a in (aap, aro, apx, apz):
                entries[a]["reads"] += 1
            ap, ro, px, pz = (entries[a]["val"] for a in (aap, aro, apx, apz))
            ap_out = ext_mul(ap, alv)
            ro_out = [(ro[e] + t) % P for e, t in enumerate(ext_mul(ext_mul([(px[e] - pz[e]) % P for e in range(4)], ap), inv_xz))]
            aapo, aroo = new_addr(), new_addr()
            entries[aapo] = {"val": ap_out, "reads": 0, "kind": "ext"}
            entries[aroo] = {"val": ro_out, "reads": 0, "kind": "ext"}
            ff_outs.append((aapo, aroo))
            ff_main.append(zv + alv + [xv] + px + pz + ap + ro + ap_out + ro_out)
            first = int(i == 0)
            ff_prep.append([first, az, (P - first) % P, aal, (P - first) % P, ax, (P - first) % P, aap, P - 1, aro, P - 1, apx, P - 1, apz, P - 1,
                            aroo, None, aapo, None, 1])
        for aapo, aroo in ff_outs[-k:]:
            pools["ext"] += [aapo, aroo]
    base_rows, ext_rows = [], []   # (opcode, addr_out, addr_in1, addr_in2, out, in1, in2)
    for which, n, rows in (("base", n_base, base_rows), ("ext", n_ext, ext_rows)):
        for _ in range(n):
            op = int(rng.integers(0, 4))
            a1, a2 = (pools[which][int(rng.integers(0, len(pools[which])))] for _ in range(2))
            x, y = entries[a1]["val"], entries[a2]["val"]
            if which == "base":
                if op == DIV and y[0] == 0:
                    op = ADD
                o = [[(x[0] + y[0]) % P, (x[0] - y[0]) % P, x[0] * y[0] % P, x[0] * pow(y[0], P - 2, P) % P][op], 0, 0, 0]
            else:
                if op == DIV:   # out = in1 / in2 needs an extension inverse: produce it as in1 := in2 * c for a fresh out = c
                    c = [int(v) for v in rng.integers(0, P, 4)]
                    # replace in1 by a fresh constant equal to y * c so that the division is exact
                    a1 = new_addr()
                    entries[a1] = {"val": ext_mul(y, c), "reads": 0, "kind": "const"}
                    x = entries[a1]["val"]
                    o = c
                else:
                    o = [[(x[e] + y[e]) % P for e in range(4)], [(x[e] - y[e]) % P for e in range(4)], ext_mul(x, y)][op]
            ao = new_addr()
            entries[a1]["reads"] += 1
            entries[a2]["reads"] += 1
            entries[ao] = {"val": o, "reads": 0, "kind": which, "row": len(rows)}
            pools[which].append(ao)
            rows.append([op, ao, a1, a2, o, x, y])
    # CommitPublicValues: the digest's eight elements are read from memory by the PublicValues chip
    pv_prep, pv_main, pv_digest = [], [], [0] * DIGEST_SIZE
    if commit_public_values:
        for i in range(DIGEST_SIZE):
            a = input_digest[i] if input_digest is not None else pools["base"][int(rng.integers(0, len(pools["base"])))]
            entries[a]["reads"] += 1
            pv_digest[i] = entries[a]["val"][0]
            pv_prep.append([int(j == i) for j in range(DIGEST_SIZE)] + [a, P - 1])
            pv_main.append([pv_digest[i]])
    # a few results are read back by the constant-memory table (MemAccessKind::Read: multiplicity negated)
    mem = []   # (value block, addr, signed multiplicity)
    results = [a for a, e in entries.items() if e["kind"] in ("base", "ext")]
    for a in results[:: max(1, len(results) // 8)]:
        entries[a]["reads"] += 1
        mem.append((entries[a]["val"], a, -1))
    for a, e in entries.items():
        if e["kind"] == "const":
            mem.append((e["val"], a, e["reads"]))
        elif e["kind"] == "fri_acc":
            mem.append((e["val"], a, -1))

    def pack_alu(rows, ext):
        ins = np.zeros((len(rows), ACCESS_COLS), dtype=np.uint64)
        ev = np.zeros((len(rows), EXT_VALUE_COLS if ext else BASE_VALUE_COLS), dtype=np.uint64)
        for i, (op, ao, a1, a2, o, x, y) in enumerate(rows):
            ins[i, 0:3] = (ao, a1, a2)
            ins[i, 3 + op] = 1
            ins[i, 7] = entries[ao]["reads"]
            ev[i] = (o + x + y) if ext else (o[0], x[0], y[0])
        return F.to_monty(ins).reshape(-1), F.to_monty(ev).reshape(-1)

    base_instrs, base_events = pack_alu(base_rows, False)
    ext_instrs, ext_events = pack_alu(ext_rows, True)
    m = np.zeros((len(mem), CONST_MEM_ENTRY_COLS), dtype=np.uint64)
    for i, (v, a, mult) in enumerate(mem):
        m[i, 0:4] = v
        m[i, 4] = a
        m[i, 5] = mult % P
    var_prep = np.array([[a, entries[a]["reads"]] for a in var_rows], dtype=np.uint64).reshape(-1, 2)
    var_values = np.array([entries[a]["val"] for a in var_rows], dtype=np.uint64).reshape(-1, 4)
    sel_prep = np.array([[1, ab, ao1, ao2, a1, a2, entries[ao1]["reads"], entries[ao2]["reads"]]
                         for ab, ao1, ao2, a1, a2, *_ in select_rows], dtype=np.uint64).reshape(-1, SELECT_PREP_COLS)
    sel_events = np.array([[bit, o1, o2, x, y] for *_, bit, o1, o2, x, y in select_rows], dtype=np.uint64).reshape(-1, SELECT_COLS)
    pos_prep = np.array([list(ins) + [x for a in outs for x in (a, entries[a]["reads"])] + [P - 1] for ins, outs in poseidon_rows],
                        dtype=np.uint64).reshape(-1, POSEIDON2_WIDE_PREP_WIDTH)
    pos_events = np.array([[entries[a]["val"][0] for a in ins] + [entries[a]["val"][0] for a in outs] for ins, outs in poseidon_rows],
                          dtype=np.uint64).reshape(-1, 32)
    for ares, rows_here in exp_prep:      # the result's multiplicity is known only now
        rows_here[-1][5] = entries[ares]["reads"]
        for row in rows_here[:-1]:
            row[5] = 0
    for row, (aapo, aroo) in zip(ff_prep, ff_outs):
        row[16], row[18] = entries[aroo]["reads"], entries[aapo]["reads"]
    exp_prep_rows = [row for _, rows_here in exp_prep for row in rows_here]
    arr = lambda rows, w: F.to_monty(np.array(rows, dtype=np.uint64).reshape(-1, w)).reshape(-1)   # noqa: E731
    more = {"fri_fold_prep": arr(ff_prep, FRI_FOLD_PREP_COLS), "fri_fold_main": arr(ff_main, FRI_FOLD_COLS), "n_fri_fold_rows": len(ff_prep),
            "exp_prep": arr(exp_prep_rows, EXP_REVERSE_BITS_PREP_COLS), "exp_main": arr(exp_main, EXP_REVERSE_BITS_COLS),
            "batch_fri_prep": arr(fri_prep, BATCH_FRI_PREP_COLS), "batch_fri_main": arr(fri_main, BATCH_FRI_COLS),
            "pv_prep": arr(pv_prep, PUBLIC_VALUES_PREP_COLS), "pv_main": arr(pv_main, 1), "pv_digest": pv_digest,
            "exp_bases": arr([entries[rows_here[0][0]]["val"][0] for _, rows_here in exp_prep], 1),
            "exp_bits": arr([entries[row[2]]["val"][0] for _, rows_here in exp_prep for row in rows_here], 1),
            "exp_offsets": np.cumsum([0] + [len(rows_here) for _, rows_here in exp_prep]).astype(np.uint32),
            "n_exp_rows": len(exp_prep_rows), "n_batch_fri_rows": len(fri_prep)}
    more["poseidon2_instrs"] = [(list(ins), list(outs), [entries[a]["reads"] for a in outs]) for ins, outs in poseidon_rows]
    extra = {**more, "poseidon2_prep": F.to_monty(pos_prep).reshape(-1), "poseidon2_events": F.to_monty(pos_events).reshape(-1),
             "n_poseidon2": len(poseidon_rows), "var_prep": F.to_monty(var_prep).reshape(-1), "var_values": F.to_monty(var_values).reshape(-1),
             "select_prep": F.to_monty(sel_prep).reshape(-1), "select_events": F.to_monty(sel_events).reshape(-1),
             "n_var": len(var_rows), "n_select": len(select_rows)}
    return {**extra, "base_instrs": base_instrs, "base_events": base_events, "ext_instrs": ext_instrs, "ext_events": ext_events,
            "mem_entries": F.to_monty(m).reshape(-1), "n_mem": len(mem)}


def _ext_mul_np(x, y):
    out = [np.zeros(len(x[0]), dtype=np.uint64) for _ in range(4)]
    for i in range(4):
        for j in range(4):
            t = F.mul(x[i], y[j])
            if i + j >= 4:
                out[i + j - 4] = F.add(out[i + j - 4], F.mul(t, W))
            else:
                out[i + j] = F.add(out[i + j], t)
    return out


def synthetic_program(ext: bool, n_instr: int, seed: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """n_instr ALU instructions and their execution: (instrs, events), both flat uint32 Montgomery words.
    instrs: n_instr x 8 (the BaseAluAccessCols / ExtAluAccessCols record of each instruction);
    events: n_instr x 3 (base) or x 12 (ext) values with out = in1 op in2 (division: out = in1 / in2 is produced as
    in1 = in2 * out). Addresses are distinct small integers, multiplicities 0..3."""
    rng = F.SplitMix64(0x52454300 + 17 * int(ext) + seed)
    op = (rng.next_u64(n_instr) % np.uint64(4)).astype(np.int64)
    addr = np.arange(3 * n_instr, dtype=np.uint64).reshape(n_instr, 3) + np.uint64(1)
    mult = rng.next_u64(n_instr) % np.uint64(4)
    instrs = np.zeros((n_instr, ACCESS_COLS), dtype=np.uint64)
    instrs[:, 0:3] = addr
    instrs[np.arange(n_instr), 3 + op] = 1
    instrs[:, 7] = mult
    k = 4 if ext else 1
    a = [rng.uniform_field(n_instr) for _ in range(k)]
    c = [rng.uniform_field(n_instr) for _ in range(k)]
    if ext:
        add = [F.add(a[e], c[e]) for e in range(4)]
        sub = [F.sub(a[e], c[e]) for e in range(4)]
        mul = _ext_mul_np(a, c)
        # per opcode: (out, in1, in2)
        out = [np.select([op == ADD, op == SUB, op == MUL, op == DIV], [add[e], sub[e], mul[e], c[e]]) for e in range(4)]
        in1 = [np.select([op == DIV], [mul[e]], a[e]) for e in range(4)]      # DIV: in1 = in2 * out with in2 = a, out = c
        in2 = [np.select([op == DIV], [a[e]], c[e]) for e in range(4)]
        events = np.stack(out + in1 + in2, axis=1)
    else:
        a, c = a[0], c[0]
        out = np.select([op == ADD, op == SUB, op == MUL, op == DIV], [F.add(a, c), F.sub(a, c), F.mul(a, c), c])
        in1 = np.where(op == DIV, F.mul(a, c), a)
        in2 = np.where(op == DIV, a, c)
        events = np.stack([out, in1, in2], axis=1)
    return F.to_monty(instrs).reshape(-1), F.to_monty(events).reshape(-1)


def padded_rows(n_records: int, fixed_log2_rows: int = -1, entries_per_row: int = ENTRIES_PER_ROW) -> int:
    rows = -(-n_records // entries_per_row)
    if fixed_log2_rows >= 0:
        if rows > (1 << fixed_log2_rows):
            raise ValueError("fixed log2 rows is too small")
        return 1 << fixed_log2_rows
    h = 16
    while h < rows:
        h <<= 1
    return h


def flat_trace(words: np.ndarray, width: int, fixed_log2_rows: int = -1, entries_per_row: int = ENTRIES_PER_ROW) -> np.ndarray:
    """Host form of the padded trace (what generate_trace / generate_preprocessed_trace return): records end to end,
    zeros after them."""
    recs = len(words) // (width // entries_per_row)
    h = padded_rows(recs, fixed_log2_rows, entries_per_row)
    out = np.zeros(h * width, dtype=np.uint32)
    out[:len(words)] = words
    return out.reshape(h, width)
