"""The reference's ALU chips as recorded AIRs: `Air::eval` of AddSub, Bitwise, Lt, ShiftLeft and ShiftRight
transcribed statement by statement onto air.AirBuilder, so the constraint order (hence the alpha powers) and the
lookup order (hence the permutation columns) are the reference's.

In the Rust integration this file does not exist: the shim runs the chips' own `eval` against a recording builder
(INTEGRATION.md). Here it stands in for that recorder so that real chips, with traces generated on the device
(csrc/tracegen.cuh), go through commit/open and the restated verifier. Sources (crates/core/machine/src/):
  AddSub      alu/add_sub/mod.rs:188-252, operations/add.rs:57-101
  Bitwise     alu/bitwise/mod.rs:201-257
  Lt          alu/lt/mod.rs:288-472
  ShiftLeft   alu/sll/mod.rs:289-415
  ShiftRight  alu/sr/mod.rs:352-545
  CloClz      alu/clo_clz/mod.rs:180-283
  helpers     air/word.rs:55-80 (slice_range_check_u8), crates/stark/src/air/builder.rs:119-280 (byte and
              instruction lookups), opcode numbers crates/core/executor/src/opcode.rs:26-48,195-216
"""
from dataclasses import dataclass, field as dc_field
from typing import List, Optional

import numpy as np

from . import air, events as E

# ByteOpcode (crates/core/executor/src/opcode.rs:195-216)
B_AND, B_OR, B_XOR, B_SLL, B_U8RANGE, B_SHRCARRY, B_LTU, B_MSB, B_U16RANGE, B_NOR = range(10)


@dataclass
class RecordedChip:
    """The fields HipProver / the oracle read from a chip (same as synth.SynChip)."""
    name: str
    log_height: int
    main_width: int
    prep_width: int = 0
    prep_index: int = -1
    log_quotient_degree: int = 1
    local_only: bool = True
    commit_scope_global: bool = False
    sends: List[air.Lookup] = dc_field(default_factory=list)
    receives: List[air.Lookup] = dc_field(default_factory=list)
    program: Optional[np.ndarray] = None
    lookups_blob: Optional[np.ndarray] = None
    num_constraints: int = 0
    trace: Optional[np.ndarray] = None
    prep_trace: Optional[np.ndarray] = None

    @property
    def perm_ext_width(self):
        return air.local_permutation_trace_width(len(self.sends) + len(self.receives), 1 << self.log_quotient_degree)


class _Rec:
    """ZKMAirBuilder surface used by the ALU chips, on top of air.AirBuilder."""

    def __init__(self, width, prep_width=0):
        self.b = air.AirBuilder(width, prep_width, 0)
        self.local, self.next = self.b.main()
        self.prep = self.b.preprocessed()[0]
        self.sends: List[air.Lookup] = []
        self.receives: List[air.Lookup] = []

    def const(self, v):
        return self.b.const(v)

    # ByteAirBuilder::send_byte / send_byte_pair (builder.rs:119-150)
    def send_byte_pair(self, opcode, a1, a2, b, c, mult):
        vals = [air.to_virtual_pair(v) for v in (opcode, a1, a2, b, c)]
        self.sends.append(air.Lookup(vals, air.to_virtual_pair(mult), air.KIND_BYTE))

    def send_byte(self, opcode, a, b, c, mult):
        self.send_byte_pair(opcode, a, 0, b, c, mult)

    # ByteAirBuilder::receive_byte / receive_byte_pair (builder.rs:152-186)
    def receive_byte_pair(self, opcode, a1, a2, b, c, mult):
        vals = [air.to_virtual_pair(v) for v in (opcode, a1, a2, b, c)]
        self.receives.append(air.Lookup(vals, air.to_virtual_pair(mult), air.KIND_BYTE))

    def receive_byte(self, opcode, a, b, c, mult):
        self.receive_byte_pair(opcode, a, 0, b, c, mult)

    # WordAirBuilder::slice_range_check_u8 (air/word.rs:55-80)
    def slice_range_check_u8(self, cols, mult):
        i = 0
        while i + 1 < len(cols):
            self.send_byte(B_U8RANGE, 0, cols[i], cols[i + 1], mult)
            i += 2
        if i < len(cols):
            self.send_byte(B_U8RANGE, 0, cols[i], 0, mult)

    # WordAirBuilder::slice_range_check_u16 (air/word.rs:83-97)
    def slice_range_check_u16(self, cols, mult):
        for c in cols:
            self.send_byte(B_U16RANGE, c, 0, 0, mult)

    def eval_range_check_24bits(self, value, limb16, limb8, do_check):
        """MemoryAirBuilder::eval_range_check_24bits (air/memory.rs:142-172)."""
        self.b.when(do_check).assert_eq(value, limb16 + limb8 * (1 << 16))
        self.send_byte(B_U16RANGE, limb16, 0, 0, do_check)
        self.send_byte(B_U8RANGE, 0, 0, limb8, do_check)

    def eval_memory_access(self, shard, clk, addr, prev_value, access, do_check):
        """MemoryAirBuilder::eval_memory_access (air/memory.rs:18-134): `access` = MemoryAccessCols (value(4), prev_shard,
        prev_clk, compare_clk, diff_16bit_limb, diff_8bit_limb); `prev_value` = the separate column group of
        MemoryWriteCols / MemoryReadWriteCols, or the value itself for MemoryReadCols."""
        b = self.b
        value = access[0:4]
        prev_shard, prev_clk, compare_clk, diff16, diff8 = access[4], access[5], access[6], access[7], access[8]
        b.assert_bool(do_check)
        # eval_memory_access_timestamp
        b.when(do_check).assert_bool(compare_clk)
        b.when(do_check).when(compare_clk).assert_eq(shard, prev_shard)
        prev_comp = compare_clk * prev_clk + (1 - compare_clk) * prev_shard
        cur_comp = compare_clk * clk + (1 - compare_clk) * shard
        self.eval_range_check_24bits(cur_comp - prev_comp - 1, diff16, diff8, do_check)
        # the previous access is sent, the current one received
        self.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [prev_shard, prev_clk, addr] + list(prev_value)],
                                     air.to_virtual_pair(do_check), air.KIND_MEMORY))
        self.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [shard, clk, addr] + list(value)],
                                        air.to_virtual_pair(do_check), air.KIND_MEMORY))

    # InstructionAirBuilder::receive_instruction (builder.rs:237-280) as the ALU chips call it: shard, clk,
    # num_extra_cycles, hi and the four flags are zero, is_sequential is one
    def send_alu(self, opcode, a, b, c, mult, hi=(0, 0, 0, 0)):
        """InstructionAirBuilder::send_alu / send_alu_with_hi (builder.rs:282-326): an instruction sent from one chip to
        an ALU chip, at the placeholder pc UNUSED_PC."""
        pc = E.UNUSED_PC
        vals = [0, 0, pc, pc + E.DEFAULT_PC_INC, pc + 2 * E.DEFAULT_PC_INC, 0, opcode] + list(a) + list(b) + list(c) + \
               list(hi) + [0, 0, 0, 0, 1]
        self.sends.append(air.Lookup([air.to_virtual_pair(v) for v in vals], air.to_virtual_pair(mult), air.KIND_INSTRUCTION))

    def receive_alu_instruction(self, pc, next_pc, opcode, a, b, c, mult):
        self.receive_instruction(pc, next_pc, next_pc + 4, opcode, a, b, c, 1, mult)

    def receive_instruction(self, pc, next_pc, next_next_pc, opcode, a, b, c, is_sequential, mult, hi=(0, 0, 0, 0), is_rw_a=0,
                            op_a_immutable=0, shard=0, clk=0, is_check_memory=0, num_extra_cycles=0, is_halt=0):
        """InstructionAirBuilder::receive_instruction (builder.rs:237-280)."""
        vals = [shard, clk, pc, next_pc, next_next_pc, num_extra_cycles, opcode] + list(a) + list(b) + list(c) + list(hi) + \
               [op_a_immutable, is_rw_a, is_check_memory, is_halt, is_sequential]
        self.receives.append(air.Lookup([air.to_virtual_pair(v) for v in vals], air.to_virtual_pair(mult), air.KIND_INSTRUCTION))


def _add_sub(r: _Rec):
    l, b = r.local, r.b
    PC, NEXT_PC, VALUE, CARRY, OP1, OP2, IS_ADD, IS_SUB = 0, 1, 2, 6, 9, 13, 17, 18
    value, carry = l[VALUE:VALUE + 4], l[CARRY:CARRY + 3]
    op1, op2 = l[OP1:OP1 + 4], l[OP2:OP2 + 4]
    is_real = l[IS_ADD] + l[IS_SUB]
    # AddOperation::eval(builder, operand_1, operand_2, add_operation, is_add + is_sub)
    real = b.when(is_real)
    base = 256
    ov0 = op1[0] + op2[0] - value[0]
    ov1 = op1[1] + op2[1] - value[1] + carry[0]
    ov2 = op1[2] + op2[2] - value[2] + carry[1]
    ov3 = op1[3] + op2[3] - value[3] + carry[2]
    real.assert_zero(ov3 * (ov3 - base))
    real.assert_zero(carry[0] * (ov0 - base))
    real.assert_zero(carry[1] * (ov1 - base))
    real.assert_zero(carry[2] * (ov2 - base))
    real.assert_zero((carry[0] - 1) * ov0)
    real.assert_zero((carry[1] - 1) * ov1)
    real.assert_zero((carry[2] - 1) * ov2)
    real.assert_bool(carry[0])
    real.assert_bool(carry[1])
    real.assert_bool(carry[2])
    real.assert_bool(is_real)
    r.slice_range_check_u8(op1, is_real)
    r.slice_range_check_u8(op2, is_real)
    r.slice_range_check_u8(value, is_real)
    # ADD: a = value, b = operand_1, c = operand_2; SUB: a = operand_1, b = value, c = operand_2
    r.receive_alu_instruction(l[PC], l[NEXT_PC], E.ADD, value, op1, op2, l[IS_ADD])
    r.receive_alu_instruction(l[PC], l[NEXT_PC], E.SUB, op1, value, op2, l[IS_SUB])
    b.assert_bool(l[IS_ADD])
    b.assert_bool(l[IS_SUB])
    b.assert_bool(is_real)


def _bitwise(r: _Rec):
    l, b = r.local, r.b
    PC, NEXT_PC, A, B, C, IS_NOR, IS_XOR, IS_OR, IS_AND = 0, 1, 2, 6, 10, 14, 15, 16, 17
    opcode = l[IS_XOR] * B_XOR + l[IS_OR] * B_OR + l[IS_AND] * B_AND + l[IS_NOR] * B_NOR
    mult = l[IS_XOR] + l[IS_OR] + l[IS_AND] + l[IS_NOR]
    for i in range(4):
        r.send_byte(opcode, l[A + i], l[B + i], l[C + i], mult)
    cpu_opcode = l[IS_XOR] * E.XOR + l[IS_OR] * E.OR + l[IS_AND] * E.AND + l[IS_NOR] * E.NOR
    r.receive_alu_instruction(l[PC], l[NEXT_PC], cpu_opcode, l[A:A + 4], l[B:B + 4], l[C:C + 4],
                              l[IS_XOR] + l[IS_OR] + l[IS_AND] + l[IS_NOR])
    is_real = l[IS_XOR] + l[IS_OR] + l[IS_AND] + l[IS_NOR]
    b.assert_bool(l[IS_XOR])
    b.assert_bool(l[IS_OR])
    b.assert_bool(l[IS_AND])
    b.assert_bool(l[IS_NOR])
    b.assert_bool(is_real)


def _lt(r: _Rec):
    l, b = r.local, r.b
    (PC, NEXT_PC, IS_SLT, IS_SLTU, A, B, C, BYTE_FLAGS, B_MASKED, C_MASKED, NOT_EQ_INV, MSB_B, MSB_C, BIT_B, BIT_C, SLTU,
     IS_COMP_EQ, IS_SIGN_EQ, CMP) = 0, 1, 2, 3, 4, 8, 12, 16, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30
    is_real = l[IS_SLT] + l[IS_SLTU]
    b_comp = [l[B + i] for i in range(4)]
    c_comp = [l[C + i] for i in range(4)]
    b_comp[3] = l[B + 3] * l[IS_SLTU] + l[B_MASKED] * l[IS_SLT]
    c_comp[3] = l[C + 3] * l[IS_SLTU] + l[C_MASKED] * l[IS_SLT]
    r.send_byte(B_AND, l[B_MASKED], l[B + 3], 0x7f, is_real)
    r.send_byte(B_AND, l[C_MASKED], l[C + 3], 0x7f, is_real)
    b.assert_eq(l[BIT_B], l[MSB_B] * l[IS_SLT])
    b.assert_eq(l[BIT_C], l[MSB_C] * l[IS_SLT])
    inv_128 = pow(128, air.F.P - 2, air.F.P)
    b.assert_eq(l[MSB_B], (l[B + 3] - l[B_MASKED]) * inv_128)
    b.assert_eq(l[MSB_C], (l[C + 3] - l[C_MASKED]) * inv_128)
    b.assert_bool(l[IS_SIGN_EQ])
    b.when(l[IS_SIGN_EQ]).assert_eq(l[BIT_B], l[BIT_C])
    b.when(is_real).when_not(l[IS_SIGN_EQ]).assert_one(l[BIT_B] + l[BIT_C])
    b.assert_eq(l[A], l[BIT_B] * (1 - l[BIT_C]) + l[IS_SIGN_EQ] * l[SLTU])
    b.assert_zero(l[A + 1])
    b.assert_zero(l[A + 2])
    b.assert_zero(l[A + 3])
    sum_flags = l[BYTE_FLAGS] + l[BYTE_FLAGS + 1] + l[BYTE_FLAGS + 2] + l[BYTE_FLAGS + 3]
    for i in range(4):
        b.assert_bool(l[BYTE_FLAGS + i])
    b.assert_bool(sum_flags)
    b.when(is_real).assert_eq(1 - l[IS_COMP_EQ], sum_flags)
    b.assert_bool(l[IS_COMP_EQ])
    visited = b.const(0)
    b_cmp = b.const(0)
    c_cmp = b.const(0)
    for i in (3, 2, 1, 0):
        flag = l[BYTE_FLAGS + i]
        visited = visited + flag
        b_cmp = b_cmp + b_comp[i] * flag
        c_cmp = c_cmp + c_comp[i] * flag
        b.when_not(visited).assert_eq(b_comp[i], c_comp[i])
        b.when(l[IS_COMP_EQ]).assert_zero(visited)
    b.assert_eq(l[CMP], b_cmp)
    b.assert_eq(l[CMP + 1], c_cmp)
    b.when_not(l[IS_COMP_EQ]).assert_eq(l[NOT_EQ_INV] * (l[CMP] - l[CMP + 1]), is_real)
    r.send_byte(B_LTU, l[SLTU], l[CMP], l[CMP + 1], is_real)
    b.assert_bool(l[IS_SLT])
    b.assert_bool(l[IS_SLTU])
    b.assert_bool(l[IS_SLT] + l[IS_SLTU])
    r.receive_alu_instruction(l[PC], l[NEXT_PC], l[IS_SLT] * E.SLT + l[IS_SLTU] * E.SLTU, l[A:A + 4], l[B:B + 4], l[C:C + 4],
                              is_real)


def _shift_left(r: _Rec):
    l, b = r.local, r.b
    PC, NEXT_PC, A, B, C, C_LSB, BY_BITS, MULT, RESULT, CARRY, BY_BYTES, IS_REAL = 0, 1, 2, 6, 10, 14, 22, 30, 31, 35, 39, 43
    base = 256
    c_byte_sum = b.const(0)
    for i in range(8):
        c_byte_sum = c_byte_sum + b.const(1 << i) * l[C_LSB + i]
    b.assert_eq(c_byte_sum, l[C])
    nbits = b.const(0)
    for i in range(3):
        nbits = nbits + l[C_LSB + i] * (1 << i)
    for i in range(8):
        b.when(l[BY_BITS + i]).assert_eq(nbits, b.const(i))
    for i in range(8):
        b.when(l[BY_BITS + i]).assert_eq(l[MULT], b.const(1 << i))
    for i in range(4):
        v = l[B + i] * l[MULT] - l[CARRY + i] * base
        if i > 0:
            v = v + l[CARRY + i - 1]
        b.assert_eq(l[RESULT + i], v)
    nbytes = l[C_LSB + 3] + l[C_LSB + 4] * 2
    for i in range(4):
        b.when(l[BY_BYTES + i]).assert_eq(nbytes, b.const(i))
    for k in range(4):
        shifting = b.when(l[BY_BYTES + k])
        for i in range(4):
            if i < k:
                shifting.assert_eq(l[A + i], b.const(0))
            else:
                shifting.assert_eq(l[A + i], l[RESULT + i - k])
    for i in range(8):
        b.assert_bool(l[C_LSB + i])
    for i in range(8):
        b.assert_bool(l[BY_BITS + i])
    s = b.const(0)
    for i in range(8):
        s = s + l[BY_BITS + i]
    b.assert_eq(s, b.const(1))
    r.slice_range_check_u8(l[RESULT:RESULT + 4], l[IS_REAL])
    r.slice_range_check_u8(l[CARRY:CARRY + 4], l[IS_REAL])
    for i in range(4):
        b.assert_bool(l[BY_BYTES + i])
    s = b.const(0)
    for i in range(4):
        s = s + l[BY_BYTES + i]
    b.assert_eq(s, b.const(1))
    b.assert_bool(l[IS_REAL])
    r.receive_alu_instruction(l[PC], l[NEXT_PC], E.SLL, l[A:A + 4], l[B:B + 4], l[C:C + 4], l[IS_REAL])


def _shift_right(r: _Rec):
    l, b = r.local, r.b
    (PC, NEXT_PC, B, C, BY_BITS, BY_BYTES, BYTE_RES, BIT_RES, SHR_CARRY, SHR_SHIFTED, B_MSB_, C_LSB, IS_SRL, IS_ROR, IS_SRA,
     IS_REAL) = 0, 1, 2, 6, 10, 18, 22, 30, 38, 46, 54, 55, 63, 64, 65, 66
    r.send_byte(B_MSB, l[B_MSB_], l[B + 3], 0, l[IS_REAL])
    c_byte_sum = b.const(0)
    for i in range(8):
        c_byte_sum = c_byte_sum + b.const(1 << i) * l[C_LSB + i]
    b.assert_eq(c_byte_sum, l[C])
    nbits = b.const(0)
    for i in range(3):
        nbits = nbits + l[C_LSB + i] * (1 << i)
    for i in range(8):
        b.when(l[BY_BITS + i]).assert_eq(nbits, b.const(i))
    s = b.const(0)
    for i in range(8):
        s = s + l[BY_BITS + i]
    b.assert_eq(s, b.const(1))
    nbytes = l[C_LSB + 3] + l[C_LSB + 4] * 2
    for i in range(4):
        b.when(l[BY_BYTES + i]).assert_eq(nbytes, b.const(i))
    s = b.const(0)
    for i in range(4):
        s = s + l[BY_BYTES + i]
    b.assert_eq(s, b.const(1))
    ext = [l[B + i] for i in range(4)]
    for i in range(4):
        ext.append(l[IS_SRA] * l[B_MSB_] * b.const(0xff) + l[IS_ROR] * l[B + i])
    for k in range(4):
        for i in range(8 - k):
            b.when(l[BY_BYTES + k]).assert_eq(l[BYTE_RES + i], ext[i + k])
    carry_mult = b.const(0)
    for i in range(8):
        carry_mult = carry_mult + b.const(1 << (8 - i)) * l[BY_BITS + i]
    nbits2 = b.const(0)
    for i in range(3):
        nbits2 = nbits2 + l[C_LSB + i] * (1 << i)
    for i in range(7, -1, -1):
        r.send_byte_pair(B_SHRCARRY, l[SHR_SHIFTED + i], l[SHR_CARRY + i], l[BYTE_RES + i], nbits2, l[IS_REAL])
    for i in range(7, -1, -1):
        v = l[SHR_SHIFTED + i]
        if i + 1 < 8:
            v = v + l[SHR_CARRY + i + 1] * carry_mult
        b.assert_eq(v, l[BIT_RES + i])
    for flag in (IS_SRL, IS_SRA, IS_ROR, IS_REAL, B_MSB_):
        b.assert_bool(l[flag])
    for i in range(4):
        b.assert_bool(l[BY_BYTES + i])
    for i in range(8):
        b.assert_bool(l[BY_BITS + i])
    for i in range(8):
        b.assert_bool(l[C_LSB + i])
    for start in (BYTE_RES, BIT_RES, SHR_CARRY, SHR_SHIFTED):
        r.slice_range_check_u8(l[start:start + 8], l[IS_REAL])
    b.assert_eq(l[IS_SRL] + l[IS_SRA] + l[IS_ROR], l[IS_REAL])
    r.receive_alu_instruction(l[PC], l[NEXT_PC], l[IS_SRL] * E.SRL + l[IS_SRA] * E.SRA + l[IS_ROR] * E.ROR,
                              l[BIT_RES:BIT_RES + 4], l[B:B + 4], l[C:C + 4], l[IS_REAL])


def _clo_clz(r: _Rec):
    l, b = r.local, r.b
    PC, NEXT_PC, A, B, BB, IS_BB_ZERO, IS_CLZ, IS_REAL = 0, 1, 2, 6, 10, 14, 15, 16
    is_clo = l[IS_REAL] - l[IS_CLZ]
    for i in range(4):
        b.when(is_clo).assert_eq(l[B + i] + l[BB + i], b.const(255))
        b.when(l[IS_CLZ]).assert_eq(l[B + i], l[BB + i])
    r.slice_range_check_u8(l[BB:BB + 4], l[IS_REAL])
    r.send_byte(B_LTU, 1, l[A], 33, l[IS_REAL])
    b.when(l[IS_REAL]).assert_zero(l[A + 1])
    b.when(l[IS_REAL]).assert_zero(l[A + 2])
    b.when(l[IS_REAL]).assert_zero(l[A + 3])
    cpu_opcode = is_clo * E.CLO + l[IS_CLZ] * E.CLZ
    r.receive_alu_instruction(l[PC], l[NEXT_PC], cpu_opcode, l[A:A + 4], l[B:B + 4], [0, 0, 0, 0], l[IS_REAL])
    b.assert_bool(l[IS_BB_ZERO])
    bb_reduced = b.const(1) * l[BB] + b.const(1 << 8) * l[BB + 1] + b.const(1 << 16) * l[BB + 2] + b.const(1 << 24) * l[BB + 3]
    b.when(l[IS_BB_ZERO]).assert_zero(bb_reduced)
    b.when(l[IS_BB_ZERO]).assert_zero(l[BB + 3])
    b.when(l[IS_BB_ZERO]).assert_eq(l[A], b.const(32))
    # bb >> (31 - a) = 1: the leading one of bb sits at bit 31 - a (skipped when bb = 0)
    r.send_alu(E.SRL, [1, 0, 0, 0], l[BB:BB + 4], [31 - l[A], 0, 0, 0], 1 - l[IS_BB_ZERO])
    b.assert_bool(l[IS_CLZ])
    b.assert_bool(l[IS_REAL])
    b.when(l[IS_CLZ]).assert_one(l[IS_REAL])


def _reduce(b, w):
    """Word::reduce (crates/stark/src/word.rs:58-63)."""
    return b.const(1) * w[0] + b.const(1 << 8) * w[1] + b.const(1 << 16) * w[2] + b.const(1 << 24) * w[3]


def _word_range_check(b, value, cols, is_real):
    """KoalaBearWordRangeChecker::range_check (operations/koala_bear_word.rs:44-100): the word is below p."""
    decomp, ands = cols[:8], cols[8:14]
    recomposed = b.const(0)
    for i in range(8):
        b.when(is_real).assert_bool(decomp[i])
        recomposed = recomposed + b.const(1 << i) * decomp[i]
    b.when(is_real).assert_eq(recomposed, value[3])
    b.when(is_real).assert_zero(decomp[7])
    b.when(is_real).assert_eq(ands[0], decomp[0] * decomp[1])
    for i in range(5):
        b.when(is_real).assert_eq(ands[1 + i], ands[i] * decomp[2 + i])
    b.when(is_real).when(ands[5]).assert_zero(value[0] + value[1] + value[2])


def _jump(r: _Rec):
    """JumpChip::eval (control_flow/jump/air.rs:21-114)."""
    l, b = r.local, r.b
    (PC, NEXT_PC, NEXT_PC_RC, NEXT_NEXT_PC, NEXT_NEXT_PC_RC, OP_A, OP_B, OP_C, IS_JUMP, IS_JUMPI, IS_JUMPDIRECT,
     OP_A_RC) = 0, 1, 5, 19, 23, 37, 41, 45, 49, 50, 51, 52
    next_pc, next_next_pc = l[NEXT_PC:NEXT_PC + 4], l[NEXT_NEXT_PC:NEXT_NEXT_PC + 4]
    op_a, op_b, op_c = l[OP_A:OP_A + 4], l[OP_B:OP_B + 4], l[OP_C:OP_C + 4]
    b.assert_bool(l[IS_JUMP])
    b.assert_bool(l[IS_JUMPI])
    b.assert_bool(l[IS_JUMPDIRECT])
    is_real = l[IS_JUMP] + l[IS_JUMPI] + l[IS_JUMPDIRECT]
    b.assert_bool(is_real)
    opcode = l[IS_JUMP] * E.JUMP + l[IS_JUMPI] * E.JUMPI + l[IS_JUMPDIRECT] * E.JUMPDIRECT
    r.receive_instruction(l[PC], _reduce(b, next_pc), _reduce(b, next_next_pc), opcode, op_a, op_b, op_c, 0, is_real)
    b.when(is_real).assert_eq(_reduce(b, op_a), _reduce(b, next_pc) + 4)
    _word_range_check(b, op_a, l[OP_A_RC:OP_A_RC + 14], is_real)
    _word_range_check(b, next_pc, l[NEXT_PC_RC:NEXT_PC_RC + 14], is_real)
    _word_range_check(b, next_next_pc, l[NEXT_NEXT_PC_RC:NEXT_NEXT_PC_RC + 14], is_real)
    for i in range(4):   # assert_word_eq under when(is_jump + is_jumpi)
        b.when(l[IS_JUMP] + l[IS_JUMPI]).assert_eq(next_next_pc[i], op_b[i])
    r.send_alu(E.ADD, next_next_pc, next_pc, op_b, l[IS_JUMPDIRECT])


def _branch(r: _Rec):
    """BranchChip::eval (control_flow/branch/air.rs:22-210)."""
    l, b = r.local, r.b
    (PC, NEXT_PC, NEXT_PC_RC, TARGET_PC, NEXT_NEXT_PC, NEXT_NEXT_PC_RC, OP_A, OP_B, OP_C, IS_BEQ, IS_BNE, IS_BLTZ, IS_BLEZ, IS_BGTZ,
     IS_BGEZ, IS_BRANCHING, A_GT_B, A_LT_B) = 0, 1, 5, 19, 23, 27, 41, 45, 49, 53, 54, 55, 56, 57, 58, 59, 60, 61
    next_pc, target_pc, next_next_pc = l[NEXT_PC:NEXT_PC + 4], l[TARGET_PC:TARGET_PC + 4], l[NEXT_NEXT_PC:NEXT_NEXT_PC + 4]
    op_a, op_b, op_c = l[OP_A:OP_A + 4], l[OP_B:OP_B + 4], l[OP_C:OP_C + 4]
    for f in (IS_BEQ, IS_BNE, IS_BLTZ, IS_BGEZ, IS_BLEZ, IS_BGTZ):
        b.assert_bool(l[f])
    is_real = l[IS_BEQ] + l[IS_BNE] + l[IS_BLTZ] + l[IS_BGEZ] + l[IS_BLEZ] + l[IS_BGTZ]
    b.assert_bool(is_real)
    opcode = (l[IS_BEQ] * E.BEQ + l[IS_BNE] * E.BNE + l[IS_BLTZ] * E.BLTZ + l[IS_BGEZ] * E.BGEZ + l[IS_BLEZ] * E.BLEZ
              + l[IS_BGTZ] * E.BGTZ)
    r.receive_instruction(l[PC], _reduce(b, next_pc), _reduce(b, next_next_pc), opcode, op_a, op_b, op_c, 0, is_real,
                          op_a_immutable=1)
    _word_range_check(b, next_pc, l[NEXT_PC_RC:NEXT_PC_RC + 14], is_real)
    _word_range_check(b, next_next_pc, l[NEXT_NEXT_PC_RC:NEXT_NEXT_PC_RC + 14], is_real)
    r.send_alu(E.ADD, target_pc, next_pc, op_c, l[IS_BRANCHING])
    b.when(is_real).when_not(l[IS_BRANCHING]).assert_eq(_reduce(b, next_pc) + 4, _reduce(b, next_next_pc))
    r.slice_range_check_u8(next_pc, is_real - l[IS_BRANCHING])
    r.slice_range_check_u8(next_next_pc, is_real - l[IS_BRANCHING])
    for i in range(4):
        b.when(is_real).when(l[IS_BRANCHING]).assert_eq(target_pc[i], next_next_pc[i])
    b.when_not(is_real).assert_zero(l[IS_BRANCHING])
    b.when(is_real).assert_bool(l[IS_BRANCHING])
    br, gt, lt = l[IS_BRANCHING], l[A_GT_B], l[A_LT_B]
    b.when(l[IS_BEQ] * br).assert_zero(gt + lt)
    b.when(l[IS_BEQ]).when_not(br).assert_one(gt + lt)
    b.when(l[IS_BNE] * br).assert_one(gt + lt)
    b.when(l[IS_BNE]).when_not(br).assert_zero(gt + lt)
    b.when(l[IS_BLTZ] * br).assert_one(lt)
    b.when(l[IS_BLTZ]).when_not(br).assert_zero(lt)
    b.when(l[IS_BLEZ] * br).assert_zero(gt)
    b.when(l[IS_BLEZ]).when_not(br).assert_one(gt)
    b.when(l[IS_BGTZ] * br).assert_one(gt)
    b.when(l[IS_BGTZ]).when_not(br).assert_zero(gt)
    b.when(l[IS_BGEZ] * br).assert_zero(lt)
    b.when(l[IS_BGEZ]).when_not(br).assert_one(lt)
    r.send_alu(E.SLT, [lt, 0, 0, 0], op_a, op_b, is_real)    # Word::extend_var
    r.send_alu(E.SLT, [gt, 0, 0, 0], op_b, op_a, is_real)


def _mul(r: _Rec):
    """MulChip::eval (alu/mul/mod.rs:345-499)."""
    l, b = r.local, r.b
    (PC, NEXT_PC, HI, A, B, C, CARRY, PRODUCT, MSB_B, MSB_C, B_SE, C_SE, IS_MUL, IS_MULT, IS_MULTU, IS_REAL, OP_HI, HI_REAL, SHARD,
     CLK) = 0, 1, 2, 6, 10, 14, 18, 26, 34, 35, 36, 37, 38, 39, 40, 41, 42, 55, 56, 57
    hi, a, wb, wc = l[HI:HI + 4], l[A:A + 4], l[B:B + 4], l[C:C + 4]
    carry, product = l[CARRY:CARRY + 8], l[PRODUCT:PRODUCT + 8]
    is_real = l[IS_REAL]
    r.send_byte(B_MSB, l[MSB_B], wb[3], 0, is_real)
    r.send_byte(B_MSB, l[MSB_C], wc[3], 0, is_real)
    b.assert_eq(l[B_SE], l[IS_MULT] * l[MSB_B])
    b.assert_eq(l[C_SE], l[IS_MULT] * l[MSB_C])
    bx = [wb[i] if i < 4 else l[B_SE] * 0xff for i in range(8)]
    cx = [wc[i] if i < 4 else l[C_SE] * 0xff for i in range(8)]
    m = [b.const(0) for _ in range(8)]
    for i in range(8):
        for j in range(8):
            if i + j < 8:
                m[i + j] = m[i + j] + bx[i] * cx[j]
    for i in range(8):
        if i == 0:
            b.assert_eq(m[i], carry[i] * 256 + product[i])
        else:
            b.assert_eq(product[i] - carry[i - 1] + carry[i] * 256, m[i])
    has_hi = l[IS_MULT] + l[IS_MULTU]
    for i in range(4):
        b.assert_eq(product[i], a[i])
        b.when(has_hi).assert_eq(product[i + 4], hi[i])
    for f in (MSB_B, MSB_C, B_SE, C_SE, IS_MUL, IS_MULT, IS_MULTU, IS_REAL, HI_REAL):
        b.assert_bool(l[f])
    b.when(l[B_SE]).assert_eq(l[MSB_B], 1)
    b.when(l[C_SE]).assert_eq(l[MSB_C], 1)
    b.when(is_real).assert_one(l[IS_MUL] + l[IS_MULT] + l[IS_MULTU])
    opcode = l[IS_MUL] * E.MUL + l[IS_MULT] * E.MULT + l[IS_MULTU] * E.MULTU
    r.slice_range_check_u16(carry, is_real)
    r.slice_range_check_u8(product, is_real)
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, opcode, a, wb, wc, 1, is_real, hi=hi, shard=l[SHARD], clk=l[CLK],
                          is_check_memory=l[HI_REAL])
    r.eval_memory_access(l[SHARD], l[CLK] + E.MEMORY_ACCESS_POSITION_HI, E.REGISTER_HI, l[OP_HI:OP_HI + 4], l[OP_HI + 4:OP_HI + 13], l[HI_REAL])
    b.when_not(is_real).assert_zero(l[HI_REAL])
    b.when(l[HI_REAL]).assert_one(l[IS_MULT] + l[IS_MULTU])
    for i in range(4):
        b.when(l[HI_REAL]).assert_eq(hi[i], l[OP_HI + 4 + i])
    b.when_not(l[HI_REAL]).assert_zero(l[CLK])
    b.when_not(l[HI_REAL]).assert_zero(l[SHARD])
    for i in range(4):
        b.when(l[IS_MUL]).assert_zero(hi[i])


def _is_zero_word(b, word, cols, is_real):
    """IsZeroWordOperation::eval (operations/is_zero_word.rs:40-72) over IsZeroOperation::eval (is_zero.rs:33-49)."""
    for i in range(4):
        inverse, result = cols[2 * i], cols[2 * i + 1]
        is_zero = 1 - inverse * word[i]
        b.when(is_real).assert_eq(is_zero, result)
        b.when(is_real).assert_bool(result)
        b.when(is_real).when(result).assert_zero(word[i])
    lower, upper, res = cols[8], cols[9], cols[10]
    b.assert_bool(is_real)
    real = b.when(is_real)
    real.assert_bool(lower)
    real.assert_bool(upper)
    real.assert_bool(res)
    real.assert_eq(lower, cols[1] * cols[3])
    real.assert_eq(upper, cols[5] * cols[7])
    real.assert_eq(res, lower * upper)


def _divrem(r: _Rec):
    """DivRemChip::eval (alu/divrem/mod.rs:398-768)."""
    l, b = r.local, r.b
    (PC, NEXT_PC, B, C, QUOT, REM, ABS_REM, ABS_C, MAX_ABS_C, CTQ, CARRY, IS_C_0, IS_DIV, IS_DIVU, IS_MOD, IS_MODU, IS_OVERFLOW,
     IS_OVERFLOW_B, IS_OVERFLOW_C, MSB_B, MSB_REM, MSB_C, B_NEG, REM_NEG, C_NEG, REM_CHECK_MULT, OP_HI, SHARD, CLK) = (
        0, 1, 2, 6, 10, 14, 18, 22, 26, 30, 38, 46, 57, 58, 59, 60, 61, 62, 73, 84, 85, 86, 87, 88, 89, 90, 91, 104, 105)
    wb, wc, quot, rem = l[B:B + 4], l[C:C + 4], l[QUOT:QUOT + 4], l[REM:REM + 4]
    abs_rem, abs_c, max_abs_c = l[ABS_REM:ABS_REM + 4], l[ABS_C:ABS_C + 4], l[MAX_ABS_C:MAX_ABS_C + 4]
    ctq, carry = l[CTQ:CTQ + 8], l[CARRY:CARRY + 8]
    is_c_0 = l[IS_C_0:IS_C_0 + 11]
    is_real = l[IS_DIV] + l[IS_DIVU] + l[IS_MOD] + l[IS_MODU]
    signed = l[IS_DIV] + l[IS_MOD]
    for msb, neg in ((MSB_B, B_NEG), (MSB_REM, REM_NEG), (MSB_C, C_NEG)):
        b.assert_eq(l[msb] * signed, l[neg])
    r.send_alu(signed * E.MULT + (l[IS_DIVU] + l[IS_MODU]) * E.MULTU, ctq[0:4], quot, wc, is_real, hi=ctq[4:8])
    # is_overflow = is_equal(b, -2^31) * is_equal(c, -1) * is_signed (IsEqualWordOperation: operations/is_equal_word.rs:30-47)
    b.assert_bool(is_real)
    _is_zero_word(b, [wb[0], wb[1], wb[2], wb[3] - 0x80], l[IS_OVERFLOW_B:IS_OVERFLOW_B + 11], is_real)
    b.assert_bool(is_real)
    _is_zero_word(b, [wc[i] - 0xff for i in range(4)], l[IS_OVERFLOW_C:IS_OVERFLOW_C + 11], is_real)
    b.assert_eq(l[IS_OVERFLOW], l[IS_OVERFLOW_B + 10] * l[IS_OVERFLOW_C + 10] * signed)
    # c * quotient + remainder = b over 64 bits
    sign_extension = l[REM_NEG] * 0xff
    total = []
    for i in range(8):
        v = ctq[i] + (rem[i] if i < 4 else sign_extension) - carry[i] * 256
        if i > 0:
            v = v + carry[i - 1]
        total.append(v)
    not_overflow = 1 - l[IS_OVERFLOW]
    for i in range(8):
        if i < 4:
            b.assert_eq(wb[i], total[i])
        else:
            b.when(not_overflow).when(l[B_NEG]).assert_eq(total[i], 0xff)
            b.when(not_overflow).when(1 - l[B_NEG]).assert_zero(total[i])
            b.when(l[IS_OVERFLOW]).assert_zero(total[i])
    rem_byte_sum = rem[0] + rem[1] + rem[2] + rem[3]
    b.when(l[REM_NEG]).assert_one(l[B_NEG])
    b.when(rem_byte_sum).when(1 - l[REM_NEG]).assert_zero(l[B_NEG])
    _is_zero_word(b, wc, is_c_0, is_real)
    for i in range(4):
        b.when(is_c_0[10]).assert_eq(quot[i], 0xff)
    for i in range(4):
        b.when_not(l[C_NEG]).assert_eq(wc[i], abs_c[i])
        b.when_not(l[REM_NEG]).assert_eq(rem[i], abs_rem[i])
    r.send_alu(E.ADD, [0, 0, 0, 0], wc, abs_c, l[C_NEG])
    r.send_alu(E.ADD, [0, 0, 0, 0], rem, abs_rem, l[REM_NEG])
    want_max = [is_c_0[10] + (1 - is_c_0[10]) * abs_c[0]] + [(1 - is_c_0[10]) * abs_c[i] for i in range(1, 4)]
    for i in range(4):
        b.when(is_real).assert_eq(max_abs_c[i], want_max[i])
    b.assert_eq((1 - is_c_0[10]) * is_real, l[REM_CHECK_MULT])
    r.send_alu(E.SLTU, [1, 0, 0, 0], abs_rem, max_abs_c, l[REM_CHECK_MULT])
    for msb, byte in ((MSB_B, wb[3]), (MSB_C, wc[3]), (MSB_REM, rem[3])):
        r.send_byte(B_MSB, l[msb], byte, 0, is_real)
    r.slice_range_check_u8(quot, is_real)
    r.slice_range_check_u8(rem, is_real)
    for cy in carry:
        b.assert_bool(cy)
    r.slice_range_check_u8(ctq, is_real)
    for f in (IS_DIV, IS_DIVU, IS_MOD, IS_MODU, IS_OVERFLOW, MSB_B, MSB_REM, MSB_C, B_NEG, REM_NEG, C_NEG):
        b.assert_bool(l[f])
    b.when(is_real).assert_eq(1, l[IS_DIVU] + l[IS_DIV] + l[IS_MOD] + l[IS_MODU])
    opcode = l[IS_DIVU] * E.DIVU + l[IS_DIV] * E.DIV + l[IS_MOD] * E.MOD + l[IS_MODU] * E.MODU
    div = l[IS_DIV] + l[IS_DIVU]
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, opcode, quot, wb, wc, 1, div, hi=rem, shard=l[SHARD], clk=l[CLK],
                          is_check_memory=1)
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, opcode, rem, wb, wc, 1, l[IS_MOD] + l[IS_MODU])
    r.eval_memory_access(l[SHARD], l[CLK] + E.MEMORY_ACCESS_POSITION_HI, E.REGISTER_HI, l[OP_HI:OP_HI + 4], l[OP_HI + 4:OP_HI + 13], div)
    for i in range(4):
        b.when(div).assert_eq(rem[i], l[OP_HI + 4 + i])


def _cpu(r: _Rec):
    """CpuChip::eval (cpu/air/mod.rs:22-108, eval_registers air/register.rs:12-75, eval_shard_clk / eval_pc / eval_is_real
    air/mod.rs:110-215). Public values: start_pc, next_pc, execution_shard (crates/stark/src/air/public_values.rs:22-60)."""
    from . import miniexec as M
    l, n, b = r.local, r.next, r.b
    (SHARD, CLK_16, CLK_8, SHARD_TO_SEND, CLK_TO_SEND, PC, NEXT_PC, NEXT_NEXT_PC, INSTR, NUM_EXTRA_CYCLES, IS_RW_A, IS_CHECK_MEMORY, IS_HALT,
     IS_SEQUENTIAL, OP_A_VALUE, HI_OR_PREV_A, OP_A_ACCESS, OP_B_ACCESS, OP_C_ACCESS, IS_REAL, OP_A_IMMUTABLE) = (
        0, 1, 2, 3, 4, 5, 6, 7, 8, 21, 22, 23, 24, 25, 26, 30, 34, 47, 56, 65, 66)
    OPCODE, OP_A, OP_B, OP_C, OP_A_0, IMM_B, IMM_C = INSTR, INSTR + 1, INSTR + 2, INSTR + 6, INSTR + 10, INSTR + 11, INSTR + 12
    instruction = l[INSTR:INSTR + 13]
    op_a_value, hi_or_prev_a = l[OP_A_VALUE:OP_A_VALUE + 4], l[HI_OR_PREV_A:HI_OR_PREV_A + 4]
    a_prev, a_access = l[OP_A_ACCESS:OP_A_ACCESS + 4], l[OP_A_ACCESS + 4:OP_A_ACCESS + 13]
    b_access, c_access = l[OP_B_ACCESS:OP_B_ACCESS + 9], l[OP_C_ACCESS:OP_C_ACCESS + 9]
    a_val, b_val, c_val = a_access[0:4], b_access[0:4], c_access[0:4]
    is_real = l[IS_REAL]
    clk = l[CLK_8] * (1 << 16) + l[CLK_16]
    # send_program (air/program.rs:14-26)
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [l[PC]] + list(instruction)], air.to_virtual_pair(is_real), air.KIND_PROGRAM))
    # eval_registers
    for i in range(4):
        b.when(l[IMM_B]).assert_eq(b_val[i], l[OP_B + i])
    for i in range(4):
        b.when(l[IMM_C]).assert_eq(c_val[i], l[OP_C + i])
    r.eval_memory_access(l[SHARD], clk + M.POS_B, l[OP_B], b_val, b_access, 1 - l[IMM_B])
    r.eval_memory_access(l[SHARD], clk + M.POS_C, l[OP_C], c_val, c_access, 1 - l[IMM_C])
    for i in range(4):
        b.when(l[OP_A_0]).assert_zero(a_val[i])
    for i in range(4):
        b.when_not(l[OP_A_0]).assert_eq(op_a_value[i], a_val[i])
    for i in range(4):
        b.when(l[IS_RW_A]).assert_eq(hi_or_prev_a[i], a_prev[i])
    r.eval_memory_access(l[SHARD], clk + M.POS_A, l[OP_A], a_prev, a_access, is_real)
    r.slice_range_check_u8(a_val, is_real)
    for i in range(4):
        b.when(l[OP_A_IMMUTABLE]).assert_eq(a_val[i], a_prev[i])
    # shard / clk to send
    b.when(is_real).assert_eq(l[SHARD_TO_SEND], l[IS_CHECK_MEMORY] * l[SHARD] + (1 - l[IS_CHECK_MEMORY]) * 0)
    b.when(is_real).assert_eq(l[CLK_TO_SEND], l[IS_CHECK_MEMORY] * clk + (1 - l[IS_CHECK_MEMORY]) * 0)
    vals = [l[SHARD_TO_SEND], l[CLK_TO_SEND], l[PC], l[NEXT_PC], l[NEXT_NEXT_PC], l[NUM_EXTRA_CYCLES], l[OPCODE]] + list(op_a_value) + \
        list(b_val) + list(c_val) + list(hi_or_prev_a) + [l[OP_A_IMMUTABLE], l[IS_RW_A], l[IS_CHECK_MEMORY], l[IS_HALT], l[IS_SEQUENTIAL]]
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in vals], air.to_virtual_pair(is_real), air.KIND_INSTRUCTION))
    # eval_shard_clk
    b.when_transition().when(n[IS_REAL]).assert_eq(l[SHARD], n[SHARD])
    r.send_byte(B_U16RANGE, l[SHARD], 0, 0, is_real)
    b.when_first_row().assert_zero(clk)
    next_clk = n[CLK_8] * (1 << 16) + n[CLK_16]
    b.when_transition().when(n[IS_REAL]).assert_eq(clk + 5 + l[NUM_EXTRA_CYCLES], next_clk)
    r.eval_range_check_24bits(clk, l[CLK_16], l[CLK_8], is_real)
    # eval_pc
    b.when(is_real).assert_eq(b.public_values(M.PV_EXECUTION_SHARD), l[SHARD])
    b.when_first_row().assert_eq(b.public_values(M.PV_START_PC), l[PC])
    b.when_first_row().when_not(l[IS_HALT]).assert_eq(l[PC] + 4, l[NEXT_PC])
    b.when_transition().when(n[IS_REAL]).assert_eq(l[NEXT_PC], n[PC])
    b.when_transition().when(n[IS_REAL]).when_not(n[IS_HALT]).assert_eq(l[NEXT_NEXT_PC], n[NEXT_PC])
    b.when_transition().when(is_real).when(l[IS_SEQUENTIAL]).assert_eq(l[NEXT_NEXT_PC], l[NEXT_PC] + 4)
    b.when_transition().when(is_real - n[IS_REAL]).assert_eq(b.public_values(M.PV_NEXT_PC), l[NEXT_PC])
    b.when_last_row().when(is_real).assert_eq(b.public_values(M.PV_NEXT_PC), l[NEXT_PC])
    # eval_is_real
    b.assert_bool(is_real)
    b.when_first_row().assert_one(is_real)
    b.when_transition().when_not(is_real).assert_zero(n[IS_REAL])
    b.when_transition().when(l[IS_HALT]).assert_zero(n[IS_REAL])
    not_real = 1 - is_real
    b.when(not_real).assert_zero(1 - l[IMM_B])
    b.when(not_real).assert_zero(1 - l[IMM_C])
    b.when(not_real).assert_zero(1 - l[IS_RW_A])


def _program(r: _Rec):
    """ProgramChip::eval (program/mod.rs:160-176): one receive of (pc, instruction) with the multiplicity column."""
    p, l = r.prep, r.local
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in p[0:14]], air.to_virtual_pair(l[0]), air.KIND_PROGRAM))


def _memory_instrs(r: _Rec):
    """MemoryInstructionsChip::eval (memory/instructions/air.rs:17-520)."""
    l, b = r.local, r.b
    (PC, NEXT_PC, SHARD, CLK, OP_A, OP_B, OP_C, FLAGS, ADDR_WORD, ADDR_ALIGNED, ADDR_LS, LS1, LS2, LS3, ADDR_RC, MEM, PREV_A, UMV, MSBIT, MSBYTE,
     IS_NEG, MSB_ZERO) = 0, 1, 2, 3, 4, 8, 12, 16, 30, 34, 35, 36, 37, 38, 39, 53, 66, 70, 74, 75, 76, 77
    is_lb, is_lbu, is_lh, is_lhu, is_lw, is_lwl, is_lwr, is_ll, is_sb, is_sh, is_sw, is_swl, is_swr, is_sc = (l[FLAGS + i] for i in range(14))
    flags = [is_lb, is_lbu, is_lh, is_lhu, is_lw, is_lwl, is_lwr, is_ll, is_sb, is_sh, is_sw, is_swl, is_swr, is_sc]
    a_val, b_val, c_val = l[OP_A:OP_A + 4], l[OP_B:OP_B + 4], l[OP_C:OP_C + 4]
    addr_word, prev_a, umv = l[ADDR_WORD:ADDR_WORD + 4], l[PREV_A:PREV_A + 4], l[UMV:UMV + 4]
    prev_mem, mem_access = l[MEM:MEM + 4], l[MEM + 4:MEM + 13]
    mem_val = mem_access[0:4]
    is_real = flags[0]
    for f in flags[1:]:
        is_real = is_real + f
    for f in flags:
        b.assert_bool(f)
    b.assert_bool(is_real)
    # eval_memory_address_and_access
    r.send_alu(E.ADD, addr_word, b_val, c_val, is_real)
    _word_range_check(b, addr_word, l[ADDR_RC:ADDR_RC + 14], is_real)
    r.slice_range_check_u8(addr_word[1:3], is_real)
    r.send_byte(B_LTU, 1, E.NUM_REGISTERS - 1, addr_word[0], l[MSB_ZERO + 1])
    b.when(l[MSB_ZERO + 1]).assert_one(is_real)
    upper = addr_word[1] + addr_word[2] + addr_word[3]      # IsZeroOperation::eval (operations/is_zero.rs:33-49)
    b.when(is_real).assert_eq(1 - l[MSB_ZERO] * upper, l[MSB_ZERO + 1])
    b.when(is_real).assert_bool(l[MSB_ZERO + 1])
    b.when(is_real).when(l[MSB_ZERO + 1]).assert_zero(upper)
    off0 = 1 - l[LS1] - l[LS2] - l[LS3]                      # eval_offset_value_flags
    b.assert_bool(l[LS1])
    b.assert_bool(l[LS2])
    b.assert_bool(l[LS3])
    b.assert_bool(off0)
    b.when(off0).assert_zero(l[ADDR_LS])
    b.when(l[LS1]).assert_one(l[ADDR_LS])
    b.when(l[LS2]).assert_eq(l[ADDR_LS], 2)
    b.when(l[LS3]).assert_eq(l[ADDR_LS], 3)
    b.when(is_real).assert_eq(l[ADDR_ALIGNED] + l[ADDR_LS], _reduce(b, addr_word))
    r.send_byte(B_AND, l[ADDR_LS], addr_word[0], 3, is_real)
    r.eval_memory_access(l[SHARD], l[CLK] + 0, l[ADDR_ALIGNED], prev_mem, mem_access, is_real)
    loads = is_lb + is_lbu + is_lh + is_lhu + is_lw + is_lwl + is_lwr + is_ll
    for i in range(4):
        b.when(loads).assert_eq(mem_val[i], prev_mem[i])
    # eval_memory_load: eval_unsigned_mem_value
    mem_byte = mem_val[0] * off0 + mem_val[1] * l[LS1] + mem_val[2] * l[LS2] + mem_val[3] * l[LS3]
    for i, want in enumerate([mem_byte, 0, 0, 0]):
        b.when(is_lb + is_lbu).assert_eq(want, umv[i])
    b.when(is_lh + is_lhu).assert_zero(l[LS1] + l[LS3])
    b.when(is_lw).assert_one(off0)
    half = [off0 * mem_val[0] + l[LS2] * mem_val[2], off0 * mem_val[1] + l[LS2] * mem_val[3], 0, 0]
    for i in range(4):
        b.when(is_lh + is_lhu).assert_eq(half[i], umv[i])
    for i in range(4):
        b.when(is_lw).assert_eq(mem_val[i], umv[i])
    lwr = [mem_val[0] * off0 + mem_val[1] * l[LS1] + mem_val[2] * l[LS2] + mem_val[3] * l[LS3],
           mem_val[1] * off0 + mem_val[2] * l[LS1] + mem_val[3] * l[LS2] + prev_a[1] * l[LS3],
           mem_val[2] * off0 + mem_val[3] * l[LS1] + prev_a[2] * (1 - l[LS1] - off0),
           mem_val[3] * off0 + prev_a[3] * (1 - off0)]
    for i in range(4):
        b.when(is_lwr).assert_eq(umv[i], lwr[i])
    lwl = [mem_val[0] * l[LS3] + prev_a[0] * (1 - l[LS3]),
           mem_val[1] * l[LS3] + mem_val[0] * l[LS2] + prev_a[1] * l[LS1] + prev_a[1] * off0,
           mem_val[2] * l[LS3] + mem_val[1] * l[LS2] + mem_val[0] * l[LS1] + prev_a[2] * off0,
           mem_val[3] * l[LS3] + mem_val[2] * l[LS2] + mem_val[1] * l[LS1] + mem_val[0] * off0]
    for i in range(4):
        b.when(is_lwl).assert_eq(umv[i], lwl[i])
    for i in range(4):
        b.when(is_ll).assert_eq(umv[i], mem_val[i])
    b.when(is_ll).assert_one(off0)
    b.assert_eq(l[IS_NEG], (is_lb + is_lh) * l[MSBIT])
    r.send_byte(B_MSB, l[MSBIT], l[MSBYTE], 0, is_lb + is_lh)
    b.assert_eq(l[MSBYTE], is_lb * umv[0] + is_lh * umv[1])
    r.send_alu(E.SUB, a_val, umv, [0, is_lb * 1, is_lh * 1, 0], l[IS_NEG])
    positive = (is_lb + is_lh - l[IS_NEG]) + is_lbu + is_lhu + is_lw + is_ll + is_lwl + is_lwr
    for i in range(4):
        b.when(positive).assert_eq(umv[i], a_val[i])
    # eval_memory_store
    sb = [a_val[0] * off0 + (1 - off0) * prev_mem[0], a_val[0] * l[LS1] + (1 - l[LS1]) * prev_mem[1],
          a_val[0] * l[LS2] + (1 - l[LS2]) * prev_mem[2], a_val[0] * l[LS3] + (1 - l[LS3]) * prev_mem[3]]
    for i in range(4):
        b.when(is_sb).assert_eq(mem_val[i], sb[i])
    b.when(is_sh).assert_zero(l[LS1] + l[LS3])
    b.when(is_sw).assert_one(off0)
    sh = [a_val[0] * off0 + (1 - off0) * prev_mem[0], a_val[1] * off0 + (1 - off0) * prev_mem[1],
          a_val[0] * l[LS2] + (1 - l[LS2]) * prev_mem[2], a_val[1] * l[LS2] + (1 - l[LS2]) * prev_mem[3]]
    for i in range(4):
        b.when(is_sh).assert_eq(mem_val[i], sh[i])
    for i in range(4):
        b.when(is_sw).assert_eq(mem_val[i], a_val[i])
    swl = [a_val[3] * off0 + a_val[2] * l[LS1] + a_val[1] * l[LS2] + a_val[0] * l[LS3],
           prev_mem[1] * off0 + a_val[3] * l[LS1] + a_val[2] * l[LS2] + a_val[1] * l[LS3],
           prev_mem[2] * (off0 + l[LS1]) + a_val[3] * l[LS2] + a_val[2] * l[LS3],
           prev_mem[3] * (1 - l[LS3]) + a_val[3] * l[LS3]]
    for i in range(4):
        b.when(is_swl).assert_eq(mem_val[i], swl[i])
    swr = [a_val[0] * off0 + prev_mem[0] * (1 - off0),
           a_val[1] * off0 + a_val[0] * l[LS1] + prev_mem[1] * (l[LS2] + l[LS3]),
           a_val[2] * off0 + a_val[1] * l[LS1] + a_val[0] * l[LS2] + prev_mem[2] * l[LS3],
           a_val[3] * off0 + a_val[2] * l[LS1] + a_val[1] * l[LS2] + a_val[0] * l[LS3]]
    for i in range(4):
        b.when(is_swr).assert_eq(mem_val[i], swr[i])
    b.when(is_sc).assert_one(off0)
    for i in range(4):
        b.when(is_sc).assert_eq(prev_a[i], mem_val[i])
    b.when(is_sc).assert_one(a_val[0])
    for i in range(1, 4):
        b.when(is_sc).assert_zero(a_val[i])
    opcode = flags[0] * E.LB
    for i in range(1, 14):
        opcode = opcode + flags[i] * (E.LB + i)
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, opcode, a_val, b_val, c_val, 1, is_real, hi=prev_a, is_rw_a=1,
                          op_a_immutable=is_sb + is_sh + is_sw + is_swl + is_swr, shard=l[SHARD], clk=l[CLK], is_check_memory=1)


def _add_double(r, a, a_hi, bw, b_hi, cols, is_real):
    """AddDoubleOperation::eval (operations/adddouble.rs:80-150): cols = value(4), value_hi(4), carry(7)."""
    b = r.b
    value, value_hi, carry = cols[0:4], cols[4:8], cols[8:15]
    ov = [a[0] + bw[0] - value[0]] + [a[i] + bw[i] - value[i] + carry[i - 1] for i in range(1, 4)]
    for i in range(4):
        b.when(is_real).assert_zero(ov[i] * (ov[i] - 256))
    for i in range(4):
        b.when(is_real).assert_zero(carry[i] * (ov[i] - 256))
    for i in range(4):
        b.when(is_real).assert_zero((carry[i] - 1) * ov[i])
    oh = [a_hi[i] + b_hi[i] - value_hi[i] + carry[3 + i] for i in range(4)]
    for i in range(4):
        b.when(is_real).assert_zero(oh[i] * (oh[i] - 256))
    for i in range(3):
        b.when(is_real).assert_zero(carry[4 + i] * (oh[i] - 256))
    for i in range(3):
        b.when(is_real).assert_zero((carry[4 + i] - 1) * oh[i])
    for i in range(7):
        b.when(is_real).assert_bool(carry[i])
    b.when(is_real).assert_bool(is_real)
    for w in (a, a_hi, bw, b_hi, value, value_hi):
        r.slice_range_check_u8(w, is_real)


def _misc_instrs(r: _Rec):
    """MiscInstrsChip::eval (misc/others/air.rs:18-446). Columns 20..63 are a union: every opcode's constraints are laid over the
    same cells and gated by its flag."""
    l, b = r.local, r.b
    SHARD, CLK, PC, NEXT_PC, OP_A, PREV_A, OP_B, OP_C, SP = 0, 1, 2, 3, 4, 8, 12, 16, 20
    is_sext, is_ins, is_ext, is_maddu, is_msubu, is_madd, is_msub, is_teq = (l[64 + i] for i in range(8))
    a_val, prev_a, b_val, c_val = l[OP_A:OP_A + 4], l[PREV_A:PREV_A + 4], l[OP_B:OP_B + 4], l[OP_C:OP_C + 4]
    opcode = (is_sext * E.SEXT + is_ins * E.INS + is_ext * E.EXT + is_maddu * E.MADDU + is_msubu * E.MSUBU + is_madd * E.MADD + is_msub * E.MSUB
              + is_teq * E.TEQ)
    is_real = is_sext + is_ins + is_ext + is_maddu + is_msubu + is_madd + is_msub + is_teq
    for f in (is_sext, is_ins, is_ext, is_maddu, is_msubu, is_madd, is_msub, is_teq):
        b.assert_bool(f)
    b.assert_bool(is_real)
    maddsub = is_maddu + is_msubu + is_madd + is_msub
    is_rw_a = maddsub + is_ins
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, opcode, a_val, b_val, c_val, 1, maddsub, hi=prev_a, is_rw_a=is_rw_a,
                          op_a_immutable=is_teq, shard=l[SHARD], clk=l[CLK], is_check_memory=maddsub)
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, opcode, a_val, b_val, c_val, 1, is_sext + is_teq + is_ext + is_ins, hi=prev_a,
                          is_rw_a=is_rw_a, op_a_immutable=is_teq)
    # eval_ext: lsb, msbd, sll_val
    lsb, msbd, sll_val = l[SP], l[SP + 1], l[SP + 2:SP + 6]
    r.send_alu(E.SLL, sll_val, b_val, [31 - lsb - msbd, 0, 0, 0], is_ext)
    r.send_alu(E.SRL, a_val, sll_val, [31 - msbd, 0, 0, 0], is_ext)
    b.when(is_ext).assert_eq(_reduce(b, c_val), lsb + msbd * 32)
    r.send_byte(B_U8RANGE, 0, lsb, msbd, is_ext)
    r.send_byte(B_LTU, 1, lsb + msbd, 32, is_ext)
    # eval_ins: lsb, msb, ror_val, srl1_val, srl_val, sll_val, add_val
    msb = l[SP + 1]
    ror_val, srl1_val, srl_val, ins_sll, add_val = (l[SP + 2 + 4 * k:SP + 6 + 4 * k] for k in range(5))
    r.send_alu(E.ROR, ror_val, prev_a, [0 + lsb, 0, 0, 0], is_ins)
    r.send_alu(E.SRL, srl1_val, ror_val, [1, 0, 0, 0], is_ins)
    r.send_alu(E.SRL, srl_val, srl1_val, [0 + msb - lsb, 0, 0, 0], is_ins)
    r.send_alu(E.SLL, ins_sll, b_val, [31 - msb + lsb, 0, 0, 0], is_ins)
    r.send_alu(E.ADD, add_val, srl_val, ins_sll, is_ins)
    r.send_alu(E.ROR, a_val, add_val, [31 - msb, 0, 0, 0], is_ins)
    b.when(is_ins).assert_eq(_reduce(b, c_val), lsb + msb * 32)
    r.send_byte(B_U8RANGE, 0, lsb, msb, is_ins)
    r.send_byte(B_LTU, 1, lsb, msb + 1, is_ins)
    r.send_byte(B_LTU, 1, msb, 32, is_ins)
    # eval_maddsub: mul_lo, mul_hi, add_operation (15), src2_hi, src2_lo, op_hi_access (13)
    mul_lo, mul_hi, add_op = l[SP:SP + 4], l[SP + 4:SP + 8], l[SP + 8:SP + 23]
    src2_hi, src2_lo = l[SP + 23:SP + 27], l[SP + 27:SP + 31]
    hi_prev, hi_access = l[SP + 31:SP + 35], l[SP + 35:SP + 44]
    hi_val = hi_access[0:4]
    is_add, is_sub = is_maddu + is_madd, is_msubu + is_msub
    r.send_alu((is_madd + is_msub) * E.MULT + (is_maddu + is_msubu) * E.MULTU, mul_lo, b_val, c_val, maddsub, hi=mul_hi)
    for i in range(4):
        b.when(maddsub).assert_eq(src2_hi[i], hi_prev[i] * is_add + hi_val[i] * is_sub)
        b.when(maddsub).assert_eq(src2_lo[i], prev_a[i] * is_add + a_val[i] * is_sub)
    _add_double(r, mul_lo, mul_hi, src2_lo, src2_hi, add_op, maddsub)
    for i in range(4):
        b.when(is_add).assert_eq(a_val[i], add_op[i])
    for i in range(4):
        b.when(is_add).assert_eq(hi_val[i], add_op[4 + i])
    for i in range(4):
        b.when(is_sub).assert_eq(prev_a[i], add_op[i])
    for i in range(4):
        b.when(is_sub).assert_eq(hi_prev[i], add_op[4 + i])
    r.eval_memory_access(l[SHARD], l[CLK] + E.MEMORY_ACCESS_POSITION_HI, E.REGISTER_HI, hi_prev, hi_access, maddsub)
    # eval_sext: most_sig_bit, sig_byte, a_eq_b (11), is_seb, is_seh
    most_sig_bit, sig_byte, a_eq_b, is_seb, is_seh = l[SP], l[SP + 1], l[SP + 2:SP + 13], l[SP + 13], l[SP + 14]
    b.assert_bool(is_teq)
    _is_zero_word(b, [a_val[i] - b_val[i] for i in range(4)], a_eq_b, is_teq)
    b.when(is_teq).assert_zero(a_eq_b[10])
    r.send_byte(B_MSB, most_sig_bit, sig_byte, 0, is_sext)
    b.when(is_sext).assert_bool(c_val[0])
    b.when(is_sext).assert_bool(is_seb)
    b.when(is_sext).assert_bool(is_seh)
    b.when(is_sext).assert_one(is_seh + is_seb)
    b.when(is_sext).when(is_seb).assert_zero(c_val[0])
    b.when(is_sext).when(is_seh).assert_one(c_val[0])
    b.when(is_sext).when(is_seb).assert_eq(b_val[0], sig_byte)
    b.when(is_sext).when(is_seh).assert_eq(b_val[1], sig_byte)
    sign_byte = most_sig_bit * 0xff
    b.when(is_sext).assert_eq(a_val[0], b_val[0])
    b.when(is_sext).when(is_seb).assert_eq(a_val[1], sign_byte)
    b.when(is_sext).when(is_seh).assert_eq(a_val[1], b_val[1])
    b.when(is_sext).assert_eq(a_val[2], sign_byte)
    b.when(is_sext).assert_eq(a_val[3], sign_byte)
    for i in range(4):
        b.when(is_sext + is_ext + is_teq).assert_zero(prev_a[i])
    b.when(is_ins + is_ext).assert_zero(c_val[2])
    b.when(is_ins + is_ext).assert_zero(c_val[3])


def _is_zero(b, a, cols, is_real):
    """IsZeroOperation::eval (operations/is_zero.rs:33-49): cols = (inverse, result)."""
    inverse, result = cols[0], cols[1]
    b.when(is_real).assert_eq(1 - inverse * a, result)
    b.when(is_real).assert_bool(result)
    b.when(is_real).when(result).assert_zero(a)


def _syscall_instrs(r: _Rec):
    """SyscallInstrsChip::eval (syscall/instructions/air.rs:22-400). Public values: committed_value_digest (words 0..8 as bytes),
    deferred_proofs_digest (32..40), exit_code (42)."""
    l, b = r.local, r.b
    (PC, NEXT_PC, SHARD, CLK, EXTRA, IS_HALT, IS_LINUX, A1_ZERO, SYSCALL_ID, OP_A, OP_B, OP_C, PREV_A, IS_ENTER, IS_HINT, HALT_CHECK, EXIT_CHECK,
     IS_COMMIT, IS_DEFERRED, BITMAP, B_RC, C_RC, B_CHECK, C_CHECK, IS_REAL) = (
        0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 14, 18, 22, 26, 28, 30, 32, 34, 36, 38, 46, 60, 74, 75, 76)
    a_val, b_val, c_val, prev_a = l[OP_A:OP_A + 4], l[OP_B:OP_B + 4], l[OP_C:OP_C + 4], l[PREV_A:PREV_A + 4]
    is_real = l[IS_REAL]
    syscall_id = prev_a[0] + prev_a[1] * 256
    send_to_table = l[IS_LINUX] + prev_a[2]
    b.assert_bool(is_real)
    # eval_is_halt_syscall
    _is_zero(b, syscall_id - E.SYS_HALT, l[HALT_CHECK:HALT_CHECK + 2], is_real)
    _is_zero(b, syscall_id - E.SYS_EXT_GROUP, l[EXIT_CHECK:EXIT_CHECK + 2], is_real)
    b.assert_eq(l[IS_HALT], (l[HALT_CHECK + 1] + l[EXIT_CHECK + 1]) * is_real)
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, E.SYSCALL, a_val, b_val, c_val, 1 - l[IS_HALT], is_real, hi=prev_a, is_rw_a=1,
                          shard=l[SHARD], clk=l[CLK], is_check_memory=1, num_extra_cycles=l[EXTRA], is_halt=l[IS_HALT])
    b.assert_eq(l[EXTRA], prev_a[3] * is_real)
    # eval_syscall
    b.assert_bool(prev_a[2])
    b.assert_bool(l[IS_LINUX])
    b.assert_bool(send_to_table)
    _is_zero(b, prev_a[1], l[A1_ZERO:A1_ZERO + 2], is_real)
    b.when(is_real).assert_eq(l[IS_LINUX], 1 - l[A1_ZERO + 1])
    b.when(1 - is_real).assert_zero(send_to_table)
    b.assert_bool(l[B_CHECK])
    b.assert_bool(l[C_CHECK])
    b.when(send_to_table).assert_one(l[B_CHECK])
    b.when(l[IS_HALT]).assert_one(l[B_CHECK])
    b.when(send_to_table).assert_one(l[C_CHECK])
    b.when(l[IS_DEFERRED + 1]).assert_one(l[C_CHECK])
    b.when_not(is_real).assert_zero(l[B_CHECK])
    b.when_not(is_real).assert_zero(l[C_CHECK])
    _word_range_check(b, b_val, l[B_RC:B_RC + 14], l[B_CHECK])
    _word_range_check(b, c_val, l[C_RC:C_RC + 14], l[C_CHECK])
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], syscall_id, _reduce(b, b_val), _reduce(b, c_val)]],
                              air.to_virtual_pair(send_to_table), air.KIND_SYSCALL))
    halves = lambda w: [w[0] + w[1] * 256, w[2] + w[3] * 256]   # noqa: E731  (word_to_halves, builder.rs:384-390)
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK]] + halves(a_val) + halves(b_val) + halves(c_val)],
                              air.to_virtual_pair(l[IS_LINUX]), air.KIND_SYSCALL_RESULT))
    _is_zero(b, syscall_id - E.SYS_ENTER_UNCONSTRAINED, l[IS_ENTER:IS_ENTER + 2], is_real)
    is_enter = l[IS_ENTER + 1]
    b.when(is_real).when_not(is_enter).assert_eq(l[SYSCALL_ID], syscall_id)
    b.when(is_real).when(is_enter).assert_eq(l[SYSCALL_ID], E.SYS_EXIT_UNCONSTRAINED)
    _is_zero(b, syscall_id - E.SYS_HINT_LEN, l[IS_HINT:IS_HINT + 2], is_real)
    is_hint = l[IS_HINT + 1]
    for i in range(4):
        b.when(is_real).when(is_enter).assert_eq(a_val[i], 0)
    for i in range(4):
        b.when(is_real).when_not(is_enter + is_hint + l[IS_LINUX]).assert_eq(a_val[i], prev_a[i])
    # eval_commit
    _is_zero(b, syscall_id - E.SYS_COMMIT, l[IS_COMMIT:IS_COMMIT + 2], is_real)
    _is_zero(b, syscall_id - E.SYS_COMMIT_DEFERRED_PROOFS, l[IS_DEFERRED:IS_DEFERRED + 2], is_real)
    is_commit, is_deferred = l[IS_COMMIT + 1], l[IS_DEFERRED + 1]
    bitmap = l[BITMAP:BITMAP + 8]
    bitmap_sum = b.const(0)
    for bit in bitmap:
        b.when(is_real).assert_bool(bit)
        bitmap_sum = bitmap_sum + bit
    b.when(is_real).when(is_commit + is_deferred).assert_one(bitmap_sum)
    b.when(is_real).when(1 - (is_commit + is_deferred)).assert_zero(bitmap_sum)
    for i, bit in enumerate(bitmap):
        b.when(is_real).when(bit).assert_eq(b_val[0], i)
    for i in range(3):
        b.when(is_real).when(is_commit + is_deferred).assert_zero(b_val[i + 1])
    for k in range(4):      # index_word_array: sum over the bitmap of the digest words' bytes
        want = b.const(0)
        for i in range(8):
            want = want + bitmap[i] * b.public_values(4 * i + k)
        b.when(is_real).when(is_commit).assert_eq(want, c_val[k])
    want = b.const(0)
    for i in range(8):
        want = want + bitmap[i] * b.public_values(32 + i)
    b.when(is_real).when(is_deferred).assert_eq(want, _reduce(b, c_val))
    # eval_halt_unimpl
    b.when(l[IS_HALT]).assert_zero(l[NEXT_PC])
    b.when(l[IS_HALT]).assert_eq(_reduce(b, b_val), b.public_values(42))


def _memory_local(r: _Rec):
    """MemoryLocalChip::eval (memory/local.rs:213-283): per entry, receive the access the shard starts from, send the one it
    ends with (kind Memory), and send both to the global table (kind Global: message, is_receive, is_send, kind)."""
    l, b = r.local, r.b
    for k in range(4):
        e = l[14 * k:14 * k + 14]
        addr, ish, fsh, iclk, fclk, ival, fval, is_real = e[0], e[1], e[2], e[3], e[4], e[5:9], e[9:13], e[13]
        b.assert_bool(is_real)
        r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [ish, iclk, addr] + list(ival)], air.to_virtual_pair(is_real), air.KIND_MEMORY))
        r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [ish, iclk, addr] + list(ival) + [is_real * 0, is_real * 1, air.KIND_MEMORY]],
                                  air.to_virtual_pair(is_real), air.KIND_GLOBAL))
        r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [fsh, fclk, addr] + list(fval) + [is_real * 1, is_real * 0, air.KIND_MEMORY]],
                                  air.to_virtual_pair(is_real), air.KIND_GLOBAL))
        r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [fsh, fclk, addr] + list(fval)], air.to_virtual_pair(is_real), air.KIND_MEMORY))


# septic extension F_p[z] / (z^7 + 2z - 8) over expressions (crates/stark/src/septic_extension.rs) and the curve y^2 = x^3 + 3z x - 3
# (crates/stark/src/septic_curve.rs:100-122, sum checkers :159-176)
SEPTIC_START_X = (637514027, 1595065213, 1998064738, 72333738, 1211544370, 822986770, 1518535784)     # septic_digest.rs:9-16
SEPTIC_START_Y = (1604177449, 90440090, 259343427, 140470264, 1162099742, 941559812, 1064053343)


def _s_mul(a, b):
    t = [None] * 13
    for i in range(7):
        for j in range(7):
            prod = a[i] * b[j]
            t[i + j] = prod if t[i + j] is None else t[i + j] + prod
    for k in range(12, 6, -1):          # z^k = z^(k-7) (8 - 2z)
        t[k - 7] = t[k - 7] + t[k] * 8
        t[k - 6] = t[k - 6] - t[k] * 2
    return t[:7]


def _s_add(a, b):
    return [x + y for x, y in zip(a, b)]


def _s_sub(a, b):
    return [x - y for x, y in zip(a, b)]


def _curve_formula(b, x):
    three_z_x = _s_mul(x, [b.const(0), b.const(3)] + [b.const(0)] * 5)
    cube = _s_add(_s_mul(_s_mul(x, x), x), three_z_x)
    return [cube[0] - 3] + cube[1:]


def _sum_checker_x(p1, p2, p3):
    dx, dy = _s_sub(p2[0], p1[0]), _s_sub(p2[1], p1[1])
    return _s_sub(_s_mul(_s_add(_s_add(p1[0], p2[0]), p3[0]), _s_mul(dx, dx)), _s_mul(dy, dy))


def _sum_checker_y(p1, p2, p3):
    return _s_sub(_s_mul(_s_add(p1[1], p3[1]), _s_sub(p2[0], p1[0])), _s_mul(_s_sub(p2[1], p1[1]), _s_sub(p1[0], p3[0])))


def _global(r: _Rec):
    """GlobalChip::eval (crates/core/machine/src/global/mod.rs:214-275) = the receive of every global message of the shard,
    GlobalLookupOperation::eval_single_digest (operations/global_lookup.rs:96-175: message -> curve point, sign of y by direction)
    and GlobalAccumulationOperation::<1>::eval_accumulation (operations/global_accumulation.rs:116-223: running sum of the points)."""
    MESSAGE, KIND, OFFSET_BITS, X, Y, Y6_BITS, RC_WITNESS, IS_RECEIVE, IS_SEND, IS_REAL, INITIAL, SUM_CHECKER, CUMULATIVE = \
        0, 7, 8, 16, 23, 30, 60, 61, 62, 63, 64, 78, 85
    l, n, b = r.local, r.next, r.b
    message, kind, is_real, is_receive, is_send = l[MESSAGE:MESSAGE + 7], l[KIND], l[IS_REAL], l[IS_RECEIVE], l[IS_SEND]
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in list(message) + [is_send, is_receive, kind]], air.to_virtual_pair(is_real),
                                 air.KIND_GLOBAL))
    # eval_single_digest
    b.assert_bool(is_real)
    offset = b.const(0)
    for i in range(8):
        b.assert_bool(l[OFFSET_BITS + i])
        offset = offset + l[OFFSET_BITS + i] * (1 << i)
    r.send_byte(B_U16RANGE, message[0], 0, 0, is_real)
    x, y = list(l[X:X + 7]), list(l[Y:Y + 7])
    b.when(is_real).assert_eq(x[0], message[0] + kind * 65536)
    for i in range(1, 6):
        b.when(is_real).assert_eq(x[i], message[i])
    b.when(is_real).assert_eq(x[6], message[6] * 256 + offset)
    for lhs, rhs in zip(_s_mul(y, y), _curve_formula(b, x)):
        b.assert_eq(lhs, rhs)
    y6_value, top_7_bits = b.const(0), b.const(0)
    for i in range(30):
        b.assert_bool(l[Y6_BITS + i])
        y6_value = y6_value + l[Y6_BITS + i] * (1 << i)
        if i >= 23:
            top_7_bits = top_7_bits + l[Y6_BITS + i]
    b.when(is_real).assert_eq(l[RC_WITNESS] * (top_7_bits - 7), b.const(1))
    b.when(is_receive).assert_eq(y[6], y6_value + 1)
    b.when(is_send).assert_eq(y[6], y6_value + ((1 << 30) - (1 << 23) + 1))
    # eval_accumulation, N = 1
    b.assert_bool(is_real)
    b.when_transition().when_not(is_real).assert_zero(n[IS_REAL])
    initial = (list(l[INITIAL:INITIAL + 7]), list(l[INITIAL + 7:INITIAL + 14]))
    total = (list(l[CUMULATIVE:CUMULATIVE + 7]), list(l[CUMULATIVE + 7:CUMULATIVE + 14]))
    witnessed = list(l[SUM_CHECKER:SUM_CHECKER + 7])
    for i in range(7):
        b.when_first_row().assert_eq(initial[0][i], b.const(SEPTIC_START_X[i]))
    for i in range(7):
        b.when_first_row().assert_eq(initial[1][i], b.const(SEPTIC_START_Y[i]))
    for lhs, rhs in zip(_sum_checker_x(initial, (x, y), total), witnessed):
        b.assert_eq(lhs, rhs)
    for w in witnessed:
        b.when(is_real).assert_zero(w)
    for c in _sum_checker_y(initial, (x, y), total):
        b.when(is_real).assert_zero(c)
    for k in range(2):
        for i in range(7):
            b.when_not(is_real).assert_eq(initial[k][i], total[k][i])
    for k in range(2):
        for i in range(7):
            b.when_transition().assert_eq(total[k][i], n[INITIAL + 7 * k + i])


def _mov_cond(r: _Rec):
    """MovCondChip::eval (misc/mov_cond/mod.rs:172-257)."""
    l, b = r.local, r.b
    PC, NEXT_PC, OP_A, PREV_A, OP_B, OP_C, C_EQ_0, IS_MNE, IS_MEQ, IS_WSBH = 0, 1, 2, 6, 10, 14, 18, 29, 30, 31
    op_a, prev_a, op_b, op_c = l[OP_A:OP_A + 4], l[PREV_A:PREV_A + 4], l[OP_B:OP_B + 4], l[OP_C:OP_C + 4]
    c_eq_0 = l[C_EQ_0:C_EQ_0 + 11]
    result = c_eq_0[10]
    is_real = l[IS_MNE] + l[IS_MEQ] + l[IS_WSBH]
    cpu_opcode = l[IS_WSBH] * E.WSBH + l[IS_MEQ] * E.MEQ + l[IS_MNE] * E.MNE
    r.receive_instruction(l[PC], l[NEXT_PC], l[NEXT_PC] + 4, cpu_opcode, op_a, op_b, op_c, 1, is_real, hi=prev_a,
                          is_rw_a=l[IS_MNE] + l[IS_MEQ])
    _is_zero_word(b, op_c, c_eq_0, is_real)
    for i in range(4):
        b.when(l[IS_MEQ]).when(result).assert_eq(op_a[i], op_b[i])
    for i in range(4):
        b.when(l[IS_MEQ]).when_not(result).assert_eq(op_a[i], prev_a[i])
    for i in range(4):
        b.when(l[IS_MNE]).when_not(result).assert_eq(op_a[i], op_b[i])
    for i in range(4):
        b.when(l[IS_MNE]).when(result).assert_eq(op_a[i], prev_a[i])
    b.when(l[IS_WSBH]).assert_eq(op_a[0], op_b[1])
    b.when(l[IS_WSBH]).assert_eq(op_a[1], op_b[0])
    b.when(l[IS_WSBH]).assert_eq(op_a[2], op_b[3])
    b.when(l[IS_WSBH]).assert_eq(op_a[3], op_b[2])
    for i in range(4):
        b.when(l[IS_WSBH]).assert_zero(prev_a[i])
    b.assert_bool(l[IS_MNE])
    b.assert_bool(l[IS_MEQ])
    b.assert_bool(l[IS_WSBH])
    b.assert_bool(is_real)


def _byte(r: _Rec):
    """ByteChip::eval (bytes/air.rs:20-74): one receive per ByteOpcode, in ByteOpcode::all() order."""
    m, t = r.local, r.prep
    B, C, AND, OR, XOR, NOR, SLL, SHR, SHR_CARRY, LTU, MSB, VALUE_U16 = range(12)  # BytePreprocessedCols
    r.receive_byte(B_AND, t[AND], t[B], t[C], m[B_AND])
    r.receive_byte(B_OR, t[OR], t[B], t[C], m[B_OR])
    r.receive_byte(B_XOR, t[XOR], t[B], t[C], m[B_XOR])
    r.receive_byte(B_SLL, t[SLL], t[B], t[C], m[B_SLL])
    r.receive_byte(B_U8RANGE, 0, t[B], t[C], m[B_U8RANGE])
    r.receive_byte_pair(B_SHRCARRY, t[SHR], t[SHR_CARRY], t[B], t[C], m[B_SHRCARRY])
    r.receive_byte(B_LTU, t[LTU], t[B], t[C], m[B_LTU])
    r.receive_byte(B_MSB, t[MSB], t[B], 0, m[B_MSB])
    r.receive_byte(B_U16RANGE, t[VALUE_U16], 0, 0, m[B_U16RANGE])
    r.receive_byte(B_NOR, t[NOR], t[B], t[C], m[B_NOR])


_EVAL = {E.CHIP_ADD_SUB: _add_sub, E.CHIP_BITWISE: _bitwise, E.CHIP_LT: _lt, E.CHIP_SHIFT_LEFT: _shift_left,
         E.CHIP_SHIFT_RIGHT: _shift_right, E.CHIP_CLO_CLZ: _clo_clz}
# MachineAir::local_only (add_sub/mod.rs:152, bitwise/mod.rs:154, lt/mod.rs:201, sll/mod.rs:226; ShiftRight keeps the default)
_LOCAL_ONLY = {E.CHIP_ADD_SUB: True, E.CHIP_BITWISE: True, E.CHIP_LT: True, E.CHIP_SHIFT_LEFT: True, E.CHIP_SHIFT_RIGHT: False,
               E.CHIP_CLO_CLZ: False}


def record_constraints(chip: int) -> _Rec:
    """The chip's own `eval`, recorded (without the permutation constraints Chip::eval appends)."""
    r = _Rec(E.CHIP_WIDTH[chip])
    _EVAL[chip](r)
    return r


def record_chip(chip: int, log_height: int) -> RecordedChip:
    """`Chip::new` + `Chip::eval` (crates/stark/src/chip.rs:54-96,263-275): lookups collected from `eval`, the chip's
    constraints followed by the LogUp constraints, quotient degree 2 (all five AIRs have degree 3)."""
    r = record_constraints(chip)
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name=E.CHIP_NAMES[chip], log_height=log_height, main_width=E.CHIP_WIDTH[chip],
                        log_quotient_degree=lqd, local_only=_LOCAL_ONLY[chip], sends=r.sends, receives=r.receives,
                        program=program, lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


BYTE_LOG_ROWS, BYTE_MULT_COLS, BYTE_PREP_COLS = 16, 10, 12


def record_byte_chip(prep_index: int = 0) -> RecordedChip:
    """The Byte chip (crates/core/machine/src/bytes/): 65536 rows, 12 preprocessed columns (the table), 10 main
    columns (multiplicities), no constraints of its own, ten byte receives."""
    r = _Rec(BYTE_MULT_COLS, BYTE_PREP_COLS)
    _byte(r)
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name="Byte", log_height=BYTE_LOG_ROWS, main_width=BYTE_MULT_COLS, prep_width=BYTE_PREP_COLS,
                        prep_index=prep_index, log_quotient_degree=lqd, local_only=False, sends=r.sends, receives=r.receives,
                        program=program, lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_jump_chip(log_height: int) -> RecordedChip:
    """The Jump chip (crates/core/machine/src/control_flow/jump/): JumpEvents, 66 columns, local_only (trace.rs:87-89)."""
    r = _Rec(E.JUMP_WIDTH)
    _jump(r)
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name="Jump", log_height=log_height, main_width=E.JUMP_WIDTH, log_quotient_degree=lqd, local_only=True,
                        sends=r.sends, receives=r.receives, program=program,
                        lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_jump_constraints() -> _Rec:
    r = _Rec(E.JUMP_WIDTH)
    _jump(r)
    return r


def record_mov_cond_constraints() -> _Rec:
    r = _Rec(E.MOV_COND_WIDTH)
    _mov_cond(r)
    return r


def record_mov_cond_chip(log_height: int) -> RecordedChip:
    """The MovCond chip (crates/core/machine/src/misc/mov_cond/mod.rs): MovCondEvents, 32 columns, local_only (:135-137)."""
    r = record_mov_cond_constraints()
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name="MovCond", log_height=log_height, main_width=E.MOV_COND_WIDTH, log_quotient_degree=lqd,
                        local_only=True, sends=r.sends, receives=r.receives, program=program,
                        lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_mul_constraints() -> _Rec:
    r = _Rec(E.MUL_WIDTH)
    _mul(r)
    return r


def record_mul_chip(log_height: int) -> RecordedChip:
    """The Mul chip (crates/core/machine/src/alu/mul/mod.rs): CompAluEvents, 58 columns, local_only (:215-217); besides the
    instruction and byte lookups it carries the HI register's memory access (one send, one receive of kind Memory)."""
    r = record_mul_constraints()
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name="Mul", log_height=log_height, main_width=E.MUL_WIDTH, log_quotient_degree=lqd,
                        local_only=True, sends=r.sends, receives=r.receives, program=program,
                        lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_divrem_constraints() -> _Rec:
    r = _Rec(E.DIVREM_WIDTH)
    _divrem(r)
    return r


def record_divrem_chip(log_height: int) -> RecordedChip:
    """The DivRem chip (crates/core/machine/src/alu/divrem/mod.rs): CompAluEvents, 106 columns, local_only (:390-392). It
    proves c * quotient through the Mul chip (a MULT / MULTU sent with the product's upper word), abs(c) and abs(remainder)
    through the AddSub chip and abs(remainder) < max(abs(c), 1) through the Lt chip; DIV / DIVU write HI."""
    r = record_divrem_constraints()
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name="DivRem", log_height=log_height, main_width=E.DIVREM_WIDTH, log_quotient_degree=lqd,
                        local_only=True, sends=r.sends, receives=r.receives, program=program,
                        lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_cpu_constraints() -> _Rec:
    from . import miniexec as M
    r = _Rec(M.CPU_WIDTH)
    _cpu(r)
    return r


def _finish(r: _Rec, name, log_height, width, local_only, prep_width=0, prep_index=-1, commit_scope_global=False) -> RecordedChip:
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, commit_scope_global)
    program = r.b.assemble()
    return RecordedChip(name=name, log_height=log_height, main_width=width, prep_width=prep_width, prep_index=prep_index,
                        log_quotient_degree=lqd, local_only=local_only, commit_scope_global=commit_scope_global, sends=r.sends, receives=r.receives,
                        program=program,
                        lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


def record_cpu_chip(log_height: int) -> RecordedChip:
    """The Cpu chip (crates/core/machine/src/cpu/): CpuEvents + the program, 67 columns, constraints between consecutive
    rows (not local_only). It is the *sender* of every instruction the other chips receive, of the program lookups and of
    the register accesses."""
    from . import miniexec as M
    return _finish(record_cpu_constraints(), "Cpu", log_height, M.CPU_WIDTH, False)


def record_memory_instrs_constraints() -> _Rec:
    r = _Rec(E.MEMORY_INSTRS_WIDTH)
    _memory_instrs(r)
    return r


def record_memory_instrs_chip(log_height: int) -> RecordedChip:
    """The MemoryInstructions chip (crates/core/machine/src/memory/instructions/): the fourteen loads and stores, MemInstrEvents,
    79 columns, local_only (trace.rs:93-95). Sends the address ADD and the sign-extension SUB to the AddSub chip."""
    return _finish(record_memory_instrs_constraints(), "MemoryInstrs", log_height, E.MEMORY_INSTRS_WIDTH, True)


def record_misc_instrs_constraints() -> _Rec:
    r = _Rec(E.MISC_INSTRS_WIDTH)
    _misc_instrs(r)
    return r


def record_misc_instrs_chip(log_height: int) -> RecordedChip:
    """The MiscInstrs chip (crates/core/machine/src/misc/others/): SEXT, EXT, INS, MADDU, MSUBU, MADD, MSUB, TEQ; MiscEvents, 72 columns
    of which 44 are a union of per-opcode layouts. It proves its shifts, rotations, additions and multiplications by sending
    them to the ShiftLeft / ShiftRight / AddSub / Mul chips."""
    return _finish(record_misc_instrs_constraints(), "MiscInstrs", log_height, E.MISC_INSTRS_WIDTH, False)


def record_syscall_instrs_constraints() -> _Rec:
    r = _Rec(E.SYSCALL_INSTRS_WIDTH)
    _syscall_instrs(r)
    return r


def record_syscall_instrs_chip(log_height: int) -> RecordedChip:
    """The SyscallInstrs chip (crates/core/machine/src/syscall/instructions/): SyscallEvents, 77 columns. HALT (next_pc = 0, exit code
    public), COMMIT / COMMIT_DEFERRED_PROOFS (the committed word is the public digest's), the bridge to the precompile tables."""
    return _finish(record_syscall_instrs_constraints(), "SyscallInstrs", log_height, E.SYSCALL_INSTRS_WIDTH, False)


def record_memory_local_chip(log_height: int) -> RecordedChip:
    """The MemoryLocal chip (crates/core/machine/src/memory/local.rs): MemoryLocalEvents, four per row, 56 columns. It closes
    the shard's memory argument: receives each touched address's state on entry, sends its state on exit, and forwards both
    to the Global chip (not built)."""
    from . import miniexec as M
    r = _Rec(M.MEMORY_LOCAL_WIDTH)
    _memory_local(r)
    return _finish(r, "MemoryLocal", log_height, M.MEMORY_LOCAL_WIDTH, False)


def record_global_constraints() -> _Rec:
    from . import miniexec as M
    r = _Rec(M.GLOBAL_WIDTH)
    _global(r)
    return r


def record_global_chip(log_height: int) -> RecordedChip:
    """The Global chip (crates/core/machine/src/global/mod.rs): GlobalLookupEvents, 99 columns, constraints between consecutive rows.
    Receives every message the shard's chips send across shards (here MemoryLocal's), maps each to a point of the septic curve
    and accumulates them; commit scope Global: the last row's running sum is the shard proof's global_cumulative_sum."""
    from . import miniexec as M
    return _finish(record_global_constraints(), "Global", log_height, M.GLOBAL_WIDTH, False, commit_scope_global=True)


def record_program_chip(log_height: int, prep_index: int = 0) -> RecordedChip:
    """The Program chip (crates/core/machine/src/program/mod.rs): preprocessed (pc, instruction) table, one multiplicity column."""
    from . import miniexec as M
    r = _Rec(M.PROGRAM_MULT_WIDTH, M.PROGRAM_PREP_WIDTH)
    _program(r)
    return _finish(r, "Program", log_height, M.PROGRAM_MULT_WIDTH, False, M.PROGRAM_PREP_WIDTH, prep_index)


def record_branch_constraints() -> _Rec:
    r = _Rec(E.BRANCH_WIDTH)
    _branch(r)
    return r


def record_branch_chip(log_height: int) -> RecordedChip:
    """The Branch chip (crates/core/machine/src/control_flow/branch/): BranchEvents, 62 columns, local_only (trace.rs:88-90)."""
    r = record_branch_constraints()
    lqd = 1
    r.b.perm_ext_width = air.local_permutation_trace_width(len(r.sends) + len(r.receives), 1 << lqd)
    air.eval_permutation_constraints(r.b, r.sends, r.receives, 1 << lqd, False)
    program = r.b.assemble()
    return RecordedChip(name="Branch", log_height=log_height, main_width=E.BRANCH_WIDTH, log_quotient_degree=lqd,
                        local_only=True, sends=r.sends, receives=r.receives, program=program,
                        lookups_blob=air.encode_lookups(r.sends, r.receives), num_constraints=int(program[2]))


# ---- the chips that close the machine: memory initialisation / finalisation, the syscall tables, the first precompile ------------------

PV_PREV_INIT_BITS, PV_LAST_INIT_BITS, PV_PREV_FINALIZE_BITS, PV_LAST_FINALIZE_BITS = 45, 77, 109, 141    # public_values.rs:22-60


def _assert_lt_bits(b, flags, a, bb, is_real):
    """AssertLtColsBits::eval (operations/cmp.rs:323-389): `flags` marks the most significant bit where a < b."""
    total = b.const(0)
    for f in flags:
        b.assert_bool(f)
        total = total + f
    b.when(is_real).assert_one(total)
    visited, a_cmp, b_cmp = b.const(0), b.const(0), b.const(0)
    for a_bit, b_bit, f in zip(reversed(a), reversed(bb), reversed(flags)):
        visited = visited + f
        a_cmp = a_cmp + a_bit * f
        b_cmp = b_cmp + b_bit * f
        b.when(is_real).when_not(visited).assert_eq(a_bit, b_bit)
    b.when(is_real).assert_eq(a_cmp, 0)
    b.when(is_real).assert_eq(b_cmp, 1)


def _memory_global(r: _Rec, finalize: bool):
    """MemoryGlobalChip::eval (memory/global.rs:263-446), kind Initialize / Finalize. Sends one message per address to the global table:
    Initialize the value every address starts from (shard 0, timestamp 0), Finalize *receives* the last access. Addresses strictly
    increase down the rows and across shards (public values previous_* / last_*_addr_bits)."""
    l, n, b = r.local, r.next, r.b
    SHARD, TIMESTAMP, ADDR, LT, ADDR_BITS, AND_DECOMP, VALUE, IS_REAL, IS_NEXT_COMP, IS_PREV_ZERO, IS_FIRST_COMP, IS_LAST_ADDR = \
        0, 1, 2, 3, 35, 67, 73, 105, 106, 107, 109, 110
    is_real = l[IS_REAL]
    b.assert_bool(is_real)
    for i in range(32):
        b.assert_bool(l[VALUE + i])
    value = []
    for k in range(4):
        byte = b.const(0)
        for i in range(8):
            byte = byte + l[VALUE + 8 * k + i] * (1 << i)
        value.append(byte)
    if not finalize:
        vals = [b.const(0), b.const(0), l[ADDR]] + value + [is_real * 1, is_real * 0, air.KIND_MEMORY]
    else:
        vals = [l[SHARD], l[TIMESTAMP], l[ADDR]] + value + [is_real * 0, is_real * 1, air.KIND_MEMORY]
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in vals], air.to_virtual_pair(is_real), air.KIND_GLOBAL))
    # KoalaBearBitDecomposition::range_check (operations/koala_bear_range.rs:50-110)
    bits = list(l[ADDR_BITS:ADDR_BITS + 32])
    recon = b.const(0)
    for i, bit in enumerate(bits):
        b.when(is_real).assert_bool(bit)
        recon = recon + bit * ((1 << i) % air.F.P)
    b.when(is_real).assert_eq(recon, l[ADDR])
    top = bits[24:32]
    b.when(is_real).assert_zero(top[7])
    b.when(is_real).assert_eq(l[AND_DECOMP], top[0] * top[1])
    for i in range(5):
        b.when(is_real).assert_eq(l[AND_DECOMP + 1 + i], l[AND_DECOMP + i] * top[2 + i])
    low = b.const(0)
    for bit in bits[:24]:
        low = low + bit
    b.when(is_real).when(l[AND_DECOMP + 5]).assert_zero(low)
    # addr < next.addr when the next row is real
    b.when_transition().assert_eq(n[IS_NEXT_COMP], n[IS_REAL])
    _assert_lt_bits(b, list(n[LT:LT + 32]), bits, list(n[ADDR_BITS:ADDR_BITS + 32]), n[IS_NEXT_COMP])
    b.when_transition().when_not(is_real).assert_zero(n[IS_REAL])
    # the first row against the previous shard's last address
    prev_base, last_base = (PV_PREV_FINALIZE_BITS, PV_LAST_FINALIZE_BITS) if finalize else (PV_PREV_INIT_BITS, PV_LAST_INIT_BITS)
    prev_bits = [b.public_values(prev_base + i) for i in range(32)]
    prev_addr = b.const(0)
    for i, bit in enumerate(prev_bits):
        prev_addr = prev_addr + bit * ((1 << i) % air.F.P)
    _is_zero(b, prev_addr, l[IS_PREV_ZERO:IS_PREV_ZERO + 2], b.is_first_row())
    prev_zero = l[IS_PREV_ZERO + 1]
    b.assert_bool(l[IS_FIRST_COMP])
    b.when_first_row().assert_eq(l[IS_FIRST_COMP], 1 - prev_zero)
    b.when_first_row().assert_one(is_real)
    _assert_lt_bits(b, list(l[LT:LT + 32]), prev_bits, bits, l[IS_FIRST_COMP])
    b.when_first_row().when(prev_zero).assert_zero(l[ADDR])
    b.when_first_row().when(prev_zero).assert_one(n[IS_REAL])
    b.when_first_row().when(prev_zero).assert_one(n[IS_NEXT_COMP])
    if not finalize:
        b.when(is_real).assert_eq(l[TIMESTAMP], 1)
    for i in range(32):
        b.when_first_row().when_not(l[IS_FIRST_COMP]).assert_zero(l[VALUE + i])
    # the last real address is the public one
    b.when_transition().assert_eq(l[IS_LAST_ADDR], is_real * (1 - n[IS_REAL]))
    for i in range(32):
        pub = b.public_values(last_base + i)
        b.when_last_row().when(is_real).assert_eq(bits[i], pub)
        b.when_transition().when(l[IS_LAST_ADDR]).assert_eq(bits[i], pub)


def _syscall_table(r: _Rec, precompile: bool):
    """SyscallChip::eval (syscall/chip.rs:308-497), shard kind Core / Precompile. Core receives what the SyscallInstrs chip sends and
    forwards it to the global table; Precompile receives it there and sends it on to the precompile chip of its shard."""
    l, b = r.local, r.b
    SHARD, CLK, ID, A1_LO, A1_HI, A2_LO, A2_HI, R_LO, R_HI, IS_LINUX, IS_REAL = range(11)
    is_real = l[IS_REAL]
    b.assert_bool(is_real)
    b.assert_bool(l[IS_LINUX])
    b.when(1 - is_real).assert_zero(l[IS_LINUX])
    b.when_not(l[IS_LINUX]).assert_zero(l[R_LO])
    b.when_not(l[IS_LINUX]).assert_zero(l[R_HI])
    arg1 = l[A1_LO] + l[A1_HI] * 65536
    arg2 = l[A2_LO] + l[A2_HI] * 65536
    for c in (A1_LO, A1_HI, A2_LO, A2_HI):
        r.send_byte(B_U16RANGE, l[c], 0, 0, is_real)
    local_vals = [air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], l[ID], arg1, arg2]]
    packed = [air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], l[R_LO], l[R_HI], l[A1_LO], l[A1_HI], l[A2_LO], l[A2_HI]]]
    (r.sends if precompile else r.receives).append(air.Lookup(local_vals, air.to_virtual_pair(is_real), air.KIND_SYSCALL))
    (r.sends if precompile else r.receives).append(air.Lookup(packed, air.to_virtual_pair(l[IS_LINUX]), air.KIND_SYSCALL_RESULT))
    is_send, is_recv = (is_real * 0, is_real * 1) if precompile else (is_real * 1, is_real * 0)
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], l[ID], l[A1_LO], l[A1_HI], l[A2_LO], l[A2_HI], is_send, is_recv,
                                                                air.KIND_SYSCALL]], air.to_virtual_pair(is_real), air.KIND_GLOBAL))
    r.sends.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], l[ID], l[R_LO], l[R_HI], b.const(0), b.const(0), is_send, is_recv,
                                                                air.KIND_SYSCALL_RESULT]], air.to_virtual_pair(is_real), air.KIND_GLOBAL))


def _poseidon2_permute(r: _Rec):
    """Poseidon2PermuteChip::eval (syscall/precompiles/poseidon2/air.rs:25-107) with the permutation constraints of
    operations/poseidon2/air.rs:79-157 (eval_external_round x 8, then eval_internal_rounds)."""
    from . import recursion as R
    rc, diag = R._poseidon2_constants()
    l, b = r.local, r.b
    EXT_STATE, INT_STATE, INT_S0, OUTPUT, EXT_SBOX, INT_SBOX = 0, 128, 144, 156, 172, 300
    SHARD, CLK, STATE_ADDR, STATE_MEM, PRE_RC, POST_RC, IS_REAL = 313, 314, 315, 316, 524, 748, 972
    is_real = l[IS_REAL]
    ext_state = [l[EXT_STATE + 16 * k:EXT_STATE + 16 * k + 16] for k in range(8)]
    int_state, s0, output = l[INT_STATE:INT_STATE + 16], l[INT_S0:INT_S0 + 12], l[OUTPUT:OUTPUT + 16]
    ext_sbox = [l[EXT_SBOX + 16 * k:EXT_SBOX + 16 * k + 16] for k in range(8)]
    int_sbox = l[INT_SBOX:INT_SBOX + 13]
    mem = [l[STATE_MEM + 13 * i:STATE_MEM + 13 * i + 13] for i in range(16)]       # MemoryWriteCols: prev_value(4), access(9)
    b.assert_bool(is_real)
    for i in range(16):
        prev = mem[i][0:4]
        _word_range_check(b, prev, l[PRE_RC + 14 * i:PRE_RC + 14 * i + 14], is_real)
        b.when(is_real).assert_eq(ext_state[0][i], _reduce(b, prev))
    for rd in range(8):
        state = list(ext_state[rd])
        if rd == 0:
            state = R._external_layer(state)
        rnd = rd if rd < 4 else rd + 13
        for i in range(16):
            add_rc = state[i] + rc[rnd][i]
            b.assert_eq(ext_sbox[rd][i], add_rc * add_rc * add_rc)
        state = R._external_layer(list(ext_sbox[rd]))
        nxt = int_state if rd == 3 else output if rd == 7 else ext_state[rd + 1]
        for i in range(16):
            b.assert_eq(nxt[i], state[i])
    state = list(int_state)
    for rd in range(13):
        add_rc = (state[0] if rd == 0 else s0[rd - 1]) + rc[4 + rd][0]
        b.assert_eq(int_sbox[rd], add_rc * add_rc * add_rc)
        state[0] = int_sbox[rd]
        state = R._internal_layer(state, diag)
        if rd < 12:
            b.assert_eq(s0[rd], state[0])
    for i in range(16):
        b.assert_eq(ext_state[4][i], state[i])
    for i in range(16):
        cur = mem[i][4:8]
        _word_range_check(b, cur, l[POST_RC + 14 * i:POST_RC + 14 * i + 14], is_real)
        b.when(is_real).assert_eq(output[i], _reduce(b, cur))
    for i in range(16):     # eval_memory_access_slice (air/memory.rs:65-82)
        r.eval_memory_access(l[SHARD], l[CLK], l[STATE_ADDR] + 4 * i, mem[i][0:4], mem[i][4:13], is_real)
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_POSEIDON2_PERMUTE & 0xffff), l[STATE_ADDR], b.const(0)]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))


def _keccak_air(b, l, n):
    """KeccakAir::eval of p3-keccak-air on the first 2633 columns, as SubAirBuilder hands them over (keccak_sponge/air.rs:121-124). The crate
    is a git dependency that is not under /root/reference (Cargo.toml:62); this restates its published air.rs / round_flags.rs / columns.rs:
    one row per round of keccak-f[1600], the state in 16-bit limbs, theta through the bits of C / C1 / A1 (the crate's c, c_prime, a_prime),
    rho and pi as an index map into A1 (`b`), chi into A2 (a_prime_prime), iota on the bits of A2[0][0]; arrays are indexed [y][x]."""
    STEP, EXPORT, PRE, A, C, CP, AP, APP, APP00, APPP00 = 0, 24, 25, 125, 225, 545, 865, 2465, 2565, 2629
    xor = lambda x, y: x + y - x * y * 2                   # noqa: E731
    xor3 = lambda x, y, z: xor(x, xor(y, z))               # noqa: E731
    # round_flags.rs: step 0 on the first row, then the flags rotate
    b.when_first_row().assert_one(l[STEP])
    for i in range(1, 24):
        b.when_first_row().assert_zero(l[STEP + i])
    for i in range(24):
        b.when_transition().assert_eq(n[STEP + (i + 1) % 24], l[STEP + i])
    first_step, final_step = l[STEP], l[STEP + 23]
    not_final = 1 - final_step
    for i in range(100):                                   # on the first step the input is the preimage
        b.when(first_step).assert_eq(l[PRE + i], l[A + i])
    for i in range(100):                                   # the preimage stays while the permutation runs
        b.when(not_final).when(b.is_transition()).assert_eq(l[PRE + i], n[PRE + i])
    b.assert_bool(l[EXPORT])
    b.when(not_final).assert_zero(l[EXPORT])
    # C1[x, z] = C[x, z] ^ C[x - 1, z] ^ C[x + 1, z - 1]
    for x in range(5):
        for z in range(64):
            b.assert_bool(l[C + 64 * x + z])
            b.assert_eq(l[CP + 64 * x + z], xor3(l[C + 64 * x + z], l[C + 64 * ((x + 4) % 5) + z], l[C + 64 * ((x + 1) % 5) + (z + 63) % 64]))
    # A[x, y, z] = A1[x, y, z] ^ C[x, z] ^ C1[x, z], limb by limb
    for y in range(5):
        for x in range(5):
            for limb in range(4):
                acc = b.const(0)
                for z in reversed(range(16 * limb, 16 * limb + 16)):
                    bit = l[AP + 64 * (5 * y + x) + z]
                    b.assert_bool(bit)
                    acc = acc * 2 + xor3(bit, l[C + 64 * x + z], l[CP + 64 * x + z])
                b.assert_eq(acc, l[A + 4 * (5 * y + x) + limb])
    # the column parity of A1 is C1: sum - C1 is 0, 2 or 4
    for x in range(5):
        for z in range(64):
            total = l[AP + 64 * x + z]
            for y in range(1, 5):
                total = total + l[AP + 64 * (5 * y + x) + z]
            diff = total - l[CP + 64 * x + z]
            b.assert_zero(diff * (diff - 2) * (diff - 4))

    def rho_pi(x, y, z):          # KeccakCols::b: B[x, y] = rot(A1[(x + 3y) mod 5, x], R[(x + 3y) mod 5][x])
        xa = (x + 3 * y) % 5
        return l[AP + 64 * (5 * x + xa) + (z + 64 - E.KECCAK_ROT[xa][x]) % 64]

    # A2[x, y] = B[x, y] ^ (~B[x + 1, y] & B[x + 2, y])
    for y in range(5):
        for x in range(5):
            for limb in range(4):
                acc = b.const(0)
                for z in reversed(range(16 * limb, 16 * limb + 16)):
                    acc = acc * 2 + xor(rho_pi(x, y, z), (1 - rho_pi((x + 1) % 5, y, z)) * rho_pi((x + 2) % 5, y, z))
                b.assert_eq(acc, l[APP + 4 * (5 * y + x) + limb])
    # A3[0, 0] = A2[0, 0] ^ RC: the bits of A2[0][0], then the round constant chosen by the step flags
    for limb in range(4):
        acc = b.const(0)
        for z in reversed(range(16 * limb, 16 * limb + 16)):
            b.assert_bool(l[APP00 + z])
            acc = acc * 2 + l[APP00 + z]
        b.assert_eq(acc, l[APP + limb])
    for limb in range(4):
        acc = b.const(0)
        for z in reversed(range(16 * limb, 16 * limb + 16)):
            rc_bit = b.const(0)
            for rd in range(24):
                rc_bit = rc_bit + l[STEP + rd] * ((E.KECCAK_RC[rd] >> z) & 1)
            acc = acc * 2 + xor(l[APP00 + z], rc_bit)
        b.assert_eq(acc, l[APPP00 + limb])
    # this round's output is the next round's input
    for x in range(5):
        for y in range(5):
            for limb in range(4):
                out = l[APPP00 + limb] if x == 0 and y == 0 else l[APP + 4 * (5 * y + x) + limb]
                b.when(b.is_transition()).when(not_final).assert_eq(out, n[A + 4 * (5 * y + x) + limb])


def _keccak_sponge(r: _Rec):
    """KeccakSpongeChip::eval (syscall/precompiles/keccak_sponge/air.rs:27-285): twenty-four rows per 36-word block; the block is read and
    xored into the running state on the first of them, the permuted state is handed to the next block (is_absorbed) or written out
    (write_output) on the last."""
    l, n, b = r.local, r.next, r.b
    NK = E.NUM_KECCAK_COLS
    (BLOCK_MEM, SHARD, CLK, IS_REAL, READ_BLOCK, INPUT_ADDRESS, OUTPUT_ADDRESS, INPUT_LEN, ALREADY_ABSORBED, IS_ABSORBED, RECEIVE_SYSCALL, WRITE_OUTPUT,
     IS_FIRST, IS_FINAL, ORIGINAL_STATE, XORED_RATE, INPUT_LENGTH_MEM, OUTPUT_MEM) = (NK, 2957, 2958, 2959, 2960, 2961, 2962, 2963, 2964, 2965, 2966,
                                                                                      2967, 2968, 2969, 2970, 3170, 3314, 3323)
    STEP, A, APP, APPP00 = 0, 125, 2465, 2629
    is_real, first_block, final_block = l[IS_REAL], l[IS_FIRST], l[IS_FINAL]
    first_step, final_step = l[STEP], l[STEP + 23]
    not_final_step = 1 - final_step
    not_final_sponge = 1 - l[WRITE_OUTPUT]
    block_mem = [l[BLOCK_MEM + 9 * i:BLOCK_MEM + 9 * i + 9] for i in range(36)]
    output_mem = [l[OUTPUT_MEM + 13 * i:OUTPUT_MEM + 13 * i + 13] for i in range(16)]
    length_mem = l[INPUT_LENGTH_MEM:INPUT_LENGTH_MEM + 9]
    # eval_flags (:126-145)
    b.assert_eq(first_block * first_step * is_real, l[RECEIVE_SYSCALL])
    b.assert_eq(final_block * final_step * is_real, l[WRITE_OUTPUT])
    b.assert_eq(l[IS_ABSORBED], final_step * (1 - final_block) * is_real)
    # eval_memory_access (:147-192); MemoryReadCols::prev_value is its value (memory/consistency/columns.rs:77-79), so "the input has not
    # changed" compares a word with itself: constraints that are identically zero but still take their place in the random combination
    r.eval_memory_access(l[SHARD], l[CLK], l[OUTPUT_ADDRESS] + 64, length_mem[0:4], length_mem, l[RECEIVE_SYSCALL])
    for k in range(4):
        b.when(is_real).assert_eq(length_mem[k], length_mem[k])
    for i in range(36):
        r.eval_memory_access(l[SHARD], l[CLK], l[INPUT_ADDRESS] + 4 * i, block_mem[i][0:4], block_mem[i], l[READ_BLOCK])
    for i in range(36):
        for k in range(4):
            b.when(is_real).assert_eq(block_mem[i][k], block_mem[i][k])
    for i in range(16):
        r.eval_memory_access(l[SHARD], l[CLK] + 1, l[OUTPUT_ADDRESS] + 4 * i, output_mem[i][0:4], output_mem[i][4:13], l[WRITE_OUTPUT])

    # eval_state_keccakf (:193-284): memory words against the permutation's 16-bit limbs
    def limbs(lo, hi):
        return [lo[0] + lo[1] * 256, lo[2] + lo[3] * 256, hi[0] + hi[1] * 256, hi[2] + hi[3] * 256]

    def word_cols(row, base, i):
        return row[base + 4 * i:base + 4 * i + 4]

    def appp(i, j):
        return l[APPP00 + j] if i == 0 else l[APP + 4 * i + j]

    for i in range(25):
        src = XORED_RATE if i < 18 else ORIGINAL_STATE      # the rate part after the block is xored in, the capacity part as it was
        mem = limbs(word_cols(l, src, 2 * i), word_cols(l, src, 2 * i + 1))
        for j in range(4):
            b.when(first_step * is_real).assert_eq(mem[j], l[A + 4 * i + j])
        mem = limbs(word_cols(n, ORIGINAL_STATE, 2 * i), word_cols(n, ORIGINAL_STATE, 2 * i + 1))
        for j in range(4):
            b.when(l[IS_ABSORBED]).assert_eq(mem[j], appp(i, j))
    for i in range(8):
        mem = limbs(output_mem[2 * i][4:8], output_mem[2 * i + 1][4:8])
        for j in range(4):
            b.when(l[WRITE_OUTPUT]).assert_eq(mem[j], appp(i, j))
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_KECCAK_SPONGE & 0xffff), l[INPUT_ADDRESS],
                                                                   l[OUTPUT_ADDRESS]]], air.to_virtual_pair(l[RECEIVE_SYSCALL]), air.KIND_SYSCALL))
    for c in (SHARD, CLK, IS_REAL, INPUT_LEN, OUTPUT_ADDRESS):       # a call's inputs stay the same down its rows
        b.when_transition().when(not_final_sponge).assert_eq(l[c], n[c])
    b.when_last_row().assert_zero(is_real)        # 24 is not a power of two: the table cannot end on a real row
    for i in range(36):                           # XorOperation::eval (operations/xor.rs:39-57)
        for k in range(4):
            r.send_byte(B_XOR, l[XORED_RATE + 4 * i + k], l[ORIGINAL_STATE + 4 * i + k], block_mem[i][k], l[READ_BLOCK])
    b.when_transition().when(not_final_step).assert_eq(l[ALREADY_ABSORBED], n[ALREADY_ABSORBED])
    b.when(first_block).assert_eq(l[ALREADY_ABSORBED], b.const(0))
    b.when(final_block).assert_eq(l[ALREADY_ABSORBED], l[INPUT_LEN] - 36)
    b.when(l[IS_ABSORBED]).assert_eq(l[ALREADY_ABSORBED], n[ALREADY_ABSORBED] - 36)
    b.when(l[IS_ABSORBED]).assert_eq(l[INPUT_ADDRESS], n[INPUT_ADDRESS] - 144)
    _keccak_air(b, l, n)


# ---- SHA-256 precompiles: the operation gadgets of crates/core/machine/src/operations/ they are built from ------------------------------

def _fixed_shift_or_rotate(r: _Rec, x, rotation, cols, is_real, rotate):
    """FixedRotateRightOperation::eval (fixed_rotate_right.rs:90-136) / FixedShiftRightOperation::eval (fixed_shift_right.rs:86-134):
    cols = value(4), shift(4), carry(4)."""
    b = r.b
    nbytes, nbits = rotation // 8, rotation % 8
    mult = 1 << (8 - nbits)
    value, shift, carry = cols[0:4], cols[4:8], cols[8:12]
    moved = [x[(i + nbytes) % 4] if rotate or i + nbytes < 4 else b.const(0) for i in range(4)]
    first_shift = last_carry = None
    for i in (3, 2, 1, 0):
        r.send_byte_pair(B_SHRCARRY, shift[i], carry[i], moved[i], nbits, is_real)
        if i == 3:
            first_shift = shift[i]
        else:
            b.assert_eq(value[i], shift[i] + last_carry * mult)
        last_carry = carry[i]
    b.assert_eq(value[3], first_shift + last_carry * mult if rotate else first_shift)


def _bitwise_op(r: _Rec, op, x, y, value, is_real):
    """XorOperation::eval (xor.rs:39-57) / AndOperation::eval (and.rs:42-60)."""
    for i in range(4):
        r.send_byte(op, value[i], x[i], y[i], is_real)


def _not_op(r: _Rec, x, value, is_real):
    """NotOperation::eval (not.rs:28-52)."""
    for i in (0, 2):
        r.send_byte_pair(B_U8RANGE, 0, 0, x[i], x[i + 1], is_real)
    for i in range(4):
        r.b.when(is_real).assert_eq(value[i] + x[i], 255)


def _add_many(r: _Rec, words, cols, is_real, range_checks_first):
    """Add4Operation::eval (add4.rs:76-145) / Add5Operation::eval (add5.rs:83-153): cols = value(4), is_carry_0..n-1 (4 each), carry(4).
    Add4 range-checks before it asserts is_real boolean, Add5 after."""
    b, n = r.b, len(words)
    value, carry = cols[0:4], cols[4 + 4 * n:8 + 4 * n]
    flags = [cols[4 + 4 * c:8 + 4 * c] for c in range(n)]

    def ranges():
        for w in words:
            r.slice_range_check_u8(w, is_real)
        r.slice_range_check_u8(value, is_real)

    if range_checks_first:
        ranges()
    b.assert_bool(is_real)
    if not range_checks_first:
        ranges()
    real = b.when(is_real)
    for i in range(4):
        total = b.const(0)
        for c in range(n):
            real.assert_bool(flags[c][i])
        for c in range(n):
            total = flags[c][i] if c == 0 else total + flags[c][i]
        real.assert_eq(total, 1)
    for i in range(4):
        acc = flags[1][i] * 1
        for c in range(2, n):
            acc = acc + flags[c][i] * c
        real.assert_eq(carry[i], acc)
    for i in range(4):
        overflow = b.const(0) if n == 5 else None
        for w in words:
            overflow = w[i] if overflow is None else overflow + w[i]
        overflow = overflow - value[i]
        if i > 0:
            overflow = overflow + carry[i - 1]
        real.assert_eq(carry[i] * 256, overflow)


def _add_op(r: _Rec, x, y, cols, is_real):
    """AddOperation::eval (add.rs:59-106): cols = value(4), carry(3)."""
    b = r.b
    value, carry = cols[0:4], cols[4:7]
    real = b.when(is_real)
    ov = [x[0] + y[0] - value[0]] + [x[i] + y[i] - value[i] + carry[i - 1] for i in (1, 2, 3)]
    real.assert_zero(ov[3] * (ov[3] - 256))
    for i in range(3):
        real.assert_zero(carry[i] * (ov[i] - 256))
    for i in range(3):
        real.assert_zero((carry[i] - 1) * ov[i])
    for i in range(3):
        real.assert_bool(carry[i])
    real.assert_bool(is_real)
    r.slice_range_check_u8(x, is_real)
    r.slice_range_check_u8(y, is_real)
    r.slice_range_check_u8(value, is_real)


def _sha_extend(r: _Rec):
    """ShaExtendChip::eval (syscall/precompiles/sha256/extend/air.rs:27-221) with eval_flags (extend/flags.rs:41-116): 48 rows per call,
    row j computes w[16 + j] = w[j] + s0(w[j + 1]) + w[j + 9] + s1(w[j + 14]) at timestamp clk + j. The counters: cycle_16 walks the
    order-16 subgroup, cycle_48 counts its laps."""
    l, n, b = r.local, r.next, r.b
    (SHARD, CLK, W_PTR, I, CYCLE_16, C16_START, C16_END, CYCLE_48, C48_START, C48_END, W15, RR_7, RR_18, RS_3, S0_INT, S0, W2, RR_17, RR_19, RS_10,
     S1_INT, S1, W16, W7, S2, W_I, IS_REAL) = (0, 1, 2, 3, 4, 5, 7, 9, 12, 13, 14, 23, 35, 47, 59, 63, 67, 76, 88, 100, 112, 116, 120, 129, 138, 162, 175)
    is_real = l[IS_REAL]
    g = air.F.two_adic_generator(4)
    # eval_flags
    b.when_first_row().assert_eq(l[CYCLE_16], g)
    b.when_first_row().assert_eq(l[I], 16)
    b.when_transition().assert_eq(l[CYCLE_16] * g, n[CYCLE_16])
    _is_zero(b, l[CYCLE_16] - g, l[C16_START:C16_START + 2], b.const(1))
    _is_zero(b, l[CYCLE_16] - 1, l[C16_END:C16_END + 2], b.const(1))
    start16, end16 = l[C16_START + 1], l[C16_END + 1]
    b.when_first_row().assert_eq(l[CYCLE_48], 1)
    b.when_first_row().assert_eq(l[CYCLE_48 + 1], 0)
    b.when_first_row().assert_eq(l[CYCLE_48 + 2], 0)
    for i in range(3):
        b.when_transition().when(end16).assert_eq(l[CYCLE_48 + i], n[CYCLE_48 + (i + 1) % 3])
        b.when_transition().when(1 - end16).assert_eq(l[CYCLE_48 + i], n[CYCLE_48 + i])
        b.assert_bool(l[CYCLE_48 + i])
    b.assert_eq(start16 * l[CYCLE_48] * is_real, l[C48_START])
    b.assert_eq(end16 * l[CYCLE_48 + 2] * is_real, l[C48_END])
    b.when_transition().when(end16 * l[CYCLE_48 + 2]).assert_eq(n[I], 16)
    b.when_transition().when_not(end16 * l[CYCLE_48 + 2]).assert_eq(l[I] + 1, n[I])
    # the inputs stay until the 48 rows are done
    for c in (SHARD, CLK, W_PTR):
        b.when_transition().when_not(end16 * l[CYCLE_48 + 2]).assert_eq(l[c], n[c])
    clk = l[CLK] + (l[I] - 16)
    reads = {W15: 15, W2: 2, W16: 16, W7: 7}
    for base in (W15, W2, W16, W7):
        mem = l[base:base + 9]
        r.eval_memory_access(l[SHARD], clk, l[W_PTR] + (l[I] - reads[base]) * 4, mem[0:4], mem, is_real)
    w15, w2, w16, w7 = l[W15:W15 + 4], l[W2:W2 + 4], l[W16:W16 + 4], l[W7:W7 + 4]
    _fixed_shift_or_rotate(r, w15, 7, l[RR_7:RR_7 + 12], is_real, True)
    _fixed_shift_or_rotate(r, w15, 18, l[RR_18:RR_18 + 12], is_real, True)
    _fixed_shift_or_rotate(r, w15, 3, l[RS_3:RS_3 + 12], is_real, False)
    _bitwise_op(r, B_XOR, l[RR_7:RR_7 + 4], l[RR_18:RR_18 + 4], l[S0_INT:S0_INT + 4], is_real)
    _bitwise_op(r, B_XOR, l[S0_INT:S0_INT + 4], l[RS_3:RS_3 + 4], l[S0:S0 + 4], is_real)
    _fixed_shift_or_rotate(r, w2, 17, l[RR_17:RR_17 + 12], is_real, True)
    _fixed_shift_or_rotate(r, w2, 19, l[RR_19:RR_19 + 12], is_real, True)
    _fixed_shift_or_rotate(r, w2, 10, l[RS_10:RS_10 + 12], is_real, False)
    _bitwise_op(r, B_XOR, l[RR_17:RR_17 + 4], l[RR_19:RR_19 + 4], l[S1_INT:S1_INT + 4], is_real)
    _bitwise_op(r, B_XOR, l[S1_INT:S1_INT + 4], l[RS_10:RS_10 + 4], l[S1:S1 + 4], is_real)
    _add_many(r, [w16, l[S0:S0 + 4], w7, l[S1:S1 + 4]], l[S2:S2 + 24], is_real, True)
    w_i = l[W_I:W_I + 13]
    r.eval_memory_access(l[SHARD], clk, l[W_PTR] + l[I] * 4, w_i[0:4], w_i[4:13], is_real)
    for k in range(4):
        b.assert_eq(w_i[4 + k], l[S2 + k])
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_SHA_EXTEND & 0xffff), l[W_PTR], b.const(0)]],
                                 air.to_virtual_pair(l[C48_START]), air.KIND_SYSCALL))
    b.assert_bool(is_real)
    b.when_transition().when_not(l[C48_END]).assert_eq(is_real, n[IS_REAL])
    b.when_last_row().assert_zero(is_real)


def _sha_compress(r: _Rec):
    """ShaCompressChip::eval (syscall/precompiles/sha256/compress/air.rs:33-506): 80 rows per call, counted by octet (row within eight) and
    octet_num (which eight): octet_num 0 reads the state words, 1..8 run the 64 rounds (one w[i] read each), 9 adds the result to the words
    read and writes them back at clk + 1."""
    l, n, b = r.local, r.next, r.b
    (SHARD, CLK, W_PTR, H_PTR, START, OCTET, OCTET_NUM, MEM, MEM_ADDR, A, K, E_RR_6, E_RR_11, E_RR_25, S1_INT, S1, E_AND_F, E_NOT, E_NOT_AND_G, CH, TEMP1,
     A_RR_2, A_RR_13, A_RR_22, S0_INT, S0, A_AND_B, A_AND_C, B_AND_C, MAJ_INT, MAJ, TEMP2, D_ADD_TEMP1, TEMP1_ADD_TEMP2, FIN_OPERAND, FIN_ADD, IS_INIT,
     IS_COMP, IS_FIN, IS_LAST, IS_REAL) = (0, 1, 2, 3, 4, 5, 13, 23, 36, 37, 69, 73, 85, 97, 109, 113, 117, 121, 125, 129, 133, 161, 173, 185, 197, 201,
                                           205, 209, 213, 217, 221, 225, 232, 239, 246, 250, 257, 258, 259, 260, 261)
    is_real = l[IS_REAL]
    octet, octet_num = l[OCTET:OCTET + 8], l[OCTET_NUM:OCTET_NUM + 10]
    word_at = lambda row, i: row[A + 4 * i:A + 4 * i + 4]      # noqa: E731   a..h
    mem_prev, mem_value = l[MEM:MEM + 4], l[MEM + 4:MEM + 8]
    # eval_control_flow_flags (:60-208)
    for i in range(8):
        b.assert_bool(octet[i])
    total = b.const(0)
    for i in range(8):
        total = total + octet[i]
    b.assert_one(total)
    b.when_first_row().assert_one(octet[0])
    for i in range(8):
        b.when_transition().when(octet[i]).assert_one(n[OCTET + (i + 1) % 8])
    for i in range(10):
        b.assert_bool(octet_num[i])
    total = b.const(0)
    for i in range(10):
        total = total + octet_num[i]
    b.assert_one(total)
    b.when_first_row().assert_one(octet_num[0])
    for i in range(10):
        b.when_transition().when_not(octet[7]).assert_eq(octet_num[i], n[OCTET_NUM + i])
    for i in range(10):
        b.when_transition().when(octet[7]).assert_eq(octet_num[i], n[OCTET_NUM + (i + 1) % 10])
    for i in range(8):
        for k in range(4):
            b.when_transition().when(octet_num[0] + octet_num[9] * (1 - octet[7])).assert_eq(word_at(l, i)[k], word_at(n, i)[k])
        for k in range(4):
            b.when_transition().when(octet_num[0] * octet[i]).assert_eq(word_at(l, i)[k], mem_value[k])
    b.assert_eq(l[IS_INIT], octet_num[0] * is_real)
    b.assert_eq(l[IS_COMP], (octet_num[1] + octet_num[2] + octet_num[3] + octet_num[4] + octet_num[5] + octet_num[6] + octet_num[7] + octet_num[8]) * is_real)
    b.assert_eq(l[IS_FIN], octet_num[9] * is_real)
    b.assert_eq(l[IS_LAST], octet[7] * octet_num[9])
    for c in (SHARD, CLK, W_PTR, H_PTR):
        b.when_transition().when(is_real).when_not(l[IS_LAST]).assert_eq(l[c], n[c])
    b.assert_bool(is_real)
    b.when_transition().when(is_real).when_not(l[IS_LAST]).assert_one(n[IS_REAL])
    b.when_transition().when_not(is_real).assert_zero(n[IS_REAL])
    b.when_last_row().assert_zero(is_real)
    # eval_memory (:210-271)
    r.eval_memory_access(l[SHARD], l[CLK] + l[IS_FIN], l[MEM_ADDR], mem_prev, l[MEM + 4:MEM + 13], l[IS_INIT] + l[IS_COMP] + l[IS_FIN])
    cycle_num = b.const(0)
    for i in range(10):
        cycle_num = cycle_num + octet_num[i] * i
    cycle_step = b.const(0)
    for i in range(8):
        cycle_step = cycle_step + octet[i] * i
    b.when(l[IS_INIT]).assert_eq(l[MEM_ADDR], l[H_PTR] + cycle_step * 4)
    b.when(l[IS_COMP]).assert_eq(l[MEM_ADDR], l[W_PTR] + ((cycle_num - 1) * 8 + cycle_step) * 4)
    b.when(l[IS_FIN]).assert_eq(l[MEM_ADDR], l[H_PTR] + cycle_step * 4)
    for i in range(8):
        for k in range(4):
            b.when(l[IS_INIT]).when(octet[i]).assert_eq(word_at(l, i)[k], mem_prev[k])
        for k in range(4):
            b.when(l[IS_INIT]).when(octet[i]).assert_eq(word_at(l, i)[k], mem_value[k])
    for k in range(4):
        b.when(l[IS_COMP]).assert_eq(mem_prev[k], mem_value[k])
    for k in range(4):
        b.when(l[IS_FIN]).assert_eq(mem_value[k], l[FIN_ADD + k])
    # eval_compression_ops (:273-469)
    for i in range(64):
        for k in range(4):
            b.when(octet_num[i // 8 + 1] * octet[i % 8]).assert_eq(l[K + k], (E.SHA_COMPRESS_K[i] >> (8 * k)) & 0xff)
    comp = l[IS_COMP]
    a_, b_, c_, d_, e_, f_, g_, h_ = (word_at(l, i) for i in range(8))
    grp = lambda base, w=4: l[base:base + w]      # noqa: E731
    _fixed_shift_or_rotate(r, e_, 6, grp(E_RR_6, 12), comp, True)
    _fixed_shift_or_rotate(r, e_, 11, grp(E_RR_11, 12), comp, True)
    _fixed_shift_or_rotate(r, e_, 25, grp(E_RR_25, 12), comp, True)
    _bitwise_op(r, B_XOR, grp(E_RR_6), grp(E_RR_11), grp(S1_INT), comp)
    _bitwise_op(r, B_XOR, grp(S1_INT), grp(E_RR_25), grp(S1), comp)
    _bitwise_op(r, B_AND, e_, f_, grp(E_AND_F), comp)
    _not_op(r, e_, grp(E_NOT), comp)
    _bitwise_op(r, B_AND, grp(E_NOT), g_, grp(E_NOT_AND_G), comp)
    _bitwise_op(r, B_XOR, grp(E_AND_F), grp(E_NOT_AND_G), grp(CH), comp)
    _add_many(r, [h_, grp(S1), grp(CH), grp(K), mem_value], grp(TEMP1, 28), comp, False)
    _fixed_shift_or_rotate(r, a_, 2, grp(A_RR_2, 12), comp, True)
    _fixed_shift_or_rotate(r, a_, 13, grp(A_RR_13, 12), comp, True)
    _fixed_shift_or_rotate(r, a_, 22, grp(A_RR_22, 12), comp, True)
    _bitwise_op(r, B_XOR, grp(A_RR_2), grp(A_RR_13), grp(S0_INT), comp)
    _bitwise_op(r, B_XOR, grp(S0_INT), grp(A_RR_22), grp(S0), comp)
    _bitwise_op(r, B_AND, a_, b_, grp(A_AND_B), comp)
    _bitwise_op(r, B_AND, a_, c_, grp(A_AND_C), comp)
    _bitwise_op(r, B_AND, b_, c_, grp(B_AND_C), comp)
    _bitwise_op(r, B_XOR, grp(A_AND_B), grp(A_AND_C), grp(MAJ_INT), comp)
    _bitwise_op(r, B_XOR, grp(MAJ_INT), grp(B_AND_C), grp(MAJ), comp)
    _add_op(r, grp(S0), grp(MAJ), grp(TEMP2, 7), comp)
    _add_op(r, d_, grp(TEMP1), grp(D_ADD_TEMP1, 7), comp)
    _add_op(r, grp(TEMP1), grp(TEMP2), grp(TEMP1_ADD_TEMP2, 7), comp)
    moves = [(7, g_), (6, f_), (5, e_), (4, grp(D_ADD_TEMP1)), (3, c_), (2, b_), (1, a_), (0, grp(TEMP1_ADD_TEMP2))]
    for i, src in moves:
        for k in range(4):
            b.when_transition().when(comp).assert_eq(word_at(n, i)[k], src[k])
    # eval_finalize_ops (:471-505)
    for k in range(4):
        picked = b.const(0)
        for i in range(8):
            picked = picked + octet[i] * word_at(l, i)[k]
        b.when(l[IS_FIN]).assert_eq(picked, l[FIN_OPERAND + k])
    _add_op(r, mem_prev, grp(FIN_OPERAND), grp(FIN_ADD, 7), l[IS_FIN])
    b.assert_eq(l[START], is_real * octet[0] * octet_num[0])
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_SHA_COMPRESS & 0xffff), l[W_PTR], l[H_PTR]]],
                                 air.to_virtual_pair(l[START]), air.KIND_SYSCALL))


def record_sha_extend_constraints() -> _Rec:
    r = _Rec(E.SHA_EXTEND_WIDTH)
    _sha_extend(r)
    return r


def record_sha_extend_chip(log_height: int) -> RecordedChip:
    """The ShaExtend precompile (crates/core/machine/src/syscall/precompiles/sha256/extend/): 48 rows per call, 176 columns."""
    return _finish(record_sha_extend_constraints(), "ShaExtend", log_height, E.SHA_EXTEND_WIDTH, False)


def record_sha_compress_constraints() -> _Rec:
    r = _Rec(E.SHA_COMPRESS_WIDTH)
    _sha_compress(r)
    return r


def record_sha_compress_chip(log_height: int) -> RecordedChip:
    """The ShaCompress precompile (crates/core/machine/src/syscall/precompiles/sha256/compress/): 80 rows per call, 262 columns."""
    return _finish(record_sha_compress_constraints(), "ShaCompress", log_height, E.SHA_COMPRESS_WIDTH, False)


# ---- big-field gadgets (crates/core/machine/src/operations/field/): a 256-bit (or 384-bit) field element is a polynomial in x = 2^8 with byte
# coefficients; an identity op(a, b, ...) = result + carry * p over the integers is checked as a polynomial identity: the difference vanishes
# at x = 256, so it is (x - 256) * witness(x), with the witness coefficients shifted by WITNESS_OFFSET and split into two byte columns.

def _poly_add(x, y):
    n = max(len(x), len(y))
    return [(x[i] if i < len(x) else None) if i >= len(y) else (y[i] if i >= len(x) else x[i] + y[i]) for i in range(n)]


def _poly_sub(b, x, y):
    n = max(len(x), len(y))
    return [x[i] if i >= len(y) else (b.const(0) - y[i] if i >= len(x) else x[i] - y[i]) for i in range(n)]


def _poly_mul(x, y):
    out = [None] * (len(x) + len(y) - 1)
    for i, xi in enumerate(x):
        for j, yj in enumerate(y):
            t = xi * yj
            out[i + j] = t if out[i + j] is None else out[i + j] + t
    return out


def _field_gadget(r: _Rec, p_vanishing, cols, n_limbs, witness_offset, is_real, n_witness=None):
    """eval_field_operation (operations/field/util_air.rs:5-29) and the range checks every gadget ends with (field_op.rs:333-336):
    cols = result(N), carry(N), witness_low(NW), witness_high(NW); NW = 2N - 2 for the fields, 2N - 1 for U256Field (curves/src/uint256.rs:31-36)."""
    b = r.b
    nw = 2 * n_limbs - 2 if n_witness is None else n_witness
    result, carry = cols[0:n_limbs], cols[n_limbs:2 * n_limbs]
    w_low, w_high = cols[2 * n_limbs:2 * n_limbs + nw], cols[2 * n_limbs + nw:2 * n_limbs + 2 * nw]
    witness = [w_low[i] + w_high[i] * 256 - witness_offset for i in range(nw)]
    # witness(x) * (x - 256)
    prod = [None] * (nw + 1)
    for i in range(nw):
        lo = witness[i] * (air.F.P - 256)
        prod[i] = lo if prod[i] is None else prod[i] + lo
        prod[i + 1] = witness[i] * 1
    for c in _poly_sub(b, p_vanishing, prod):
        b.assert_zero(c)
    r.slice_range_check_u8(result, is_real)
    r.slice_range_check_u8(carry, is_real)
    r.slice_range_check_u8(w_low, is_real)
    r.slice_range_check_u8(w_high, is_real)


def _limbs_of_const(b, value, n):
    return [b.const((value >> (8 * i)) & 0xff) for i in range(n)]


ED25519_P = (1 << 255) - 19
ED25519_D = int.from_bytes(bytes([163, 120, 89, 19, 202, 77, 235, 117, 171, 216, 65, 65, 77, 10, 112, 0, 152, 232, 121, 119, 121, 64, 199, 140, 115, 254,
                                  111, 43, 238, 108, 3, 82]), "little")      # EdwardsParameters::D of Ed25519Parameters (curves/src/edwards/ed25519.rs:47-50)


def _ed_add(r: _Rec):
    """EdAddAssignChip::eval (syscall/precompiles/edwards/ed_add.rs:247-335): (x3, y3) = ((x1 y2 + x2 y1) / (1 + d f), (y1 y2 + x1 x2) / (1 - d f))
    with f = x1 x2 y1 y2, through two FieldInnerProductCols, four FieldOpCols (Mul) and two FieldDenCols; p is overwritten at clk + 1."""
    l, b = r.local, r.b
    N, G = 32, 188
    IS_REAL, SHARD, CLK, P_PTR, Q_PTR, P_ACCESS, Q_ACCESS, GADGETS = 0, 1, 2, 3, 4, 5, 5 + 16 * 13, 5 + 16 * 13 + 16 * 9
    is_real = l[IS_REAL]
    p_access = [l[P_ACCESS + 13 * i:P_ACCESS + 13 * i + 13] for i in range(16)]      # MemoryWriteCols: prev_value(4), access(9)
    q_access = [l[Q_ACCESS + 9 * i:Q_ACCESS + 9 * i + 9] for i in range(16)]
    gadget = [l[GADGETS + G * k:GADGETS + G * k + G] for k in range(8)]
    x3_num, y3_num, x1_mul_y1, x2_mul_y2, f, d_mul_f, x3_ins, y3_ins = gadget
    prev_limbs = lambda acc, lo, hi: [c for a in acc[lo:hi] for c in a[0:4]]          # noqa: E731   limbs_from_prev_access
    x1, y1 = prev_limbs(p_access, 0, 8), prev_limbs(p_access, 8, 16)
    x2, y2 = prev_limbs(q_access, 0, 8), prev_limbs(q_access, 8, 16)                  # a read's previous value is its value
    modulus = _limbs_of_const(b, ED25519_P, N)

    def inner_product(cols, xs, ys):          # FieldInnerProductCols::eval (field_inner_product.rs:82-120)
        acc = [b.const(0)]
        for x, y in zip(xs, ys):
            acc = _poly_add(acc, _poly_mul(x, y))
        van = _poly_sub(b, _poly_sub(b, acc, cols[0:N]), _poly_mul(cols[N:2 * N], modulus))
        _field_gadget(r, van, cols, N, 1 << 14, is_real)

    def mul(cols, x, y):                      # FieldOpCols::eval, FieldOperation::Mul (field_op.rs:296-331)
        van = _poly_sub(b, _poly_sub(b, _poly_mul(x, y), cols[0:N]), _poly_mul(cols[N:2 * N], modulus))
        _field_gadget(r, van, cols, N, 1 << 14, is_real)

    def den(cols, a, bb, sign):               # FieldDenCols::eval (field_den.rs:84-125): result * (1 +- b) = a
        result = cols[0:N]
        lhs = _poly_add(_poly_mul(bb, result), result if sign else a)
        rhs = a if sign else result
        van = _poly_sub(b, _poly_sub(b, lhs, rhs), _poly_mul(cols[N:2 * N], modulus))
        _field_gadget(r, van, cols, N, 1 << 14, is_real)

    inner_product(x3_num, [x1, x2], [y2, y1])
    inner_product(y3_num, [y1, x1], [y2, x2])
    mul(x1_mul_y1, x1, y1)
    mul(x2_mul_y2, x2, y2)
    mul(f, x1_mul_y1[0:N], x2_mul_y2[0:N])
    mul(d_mul_f, f[0:N], _limbs_of_const(b, ED25519_D, N))
    den(x3_ins, x3_num[0:N], d_mul_f[0:N], True)
    den(y3_ins, y3_num[0:N], d_mul_f[0:N], False)
    value_limbs = [c for a in p_access for c in a[4:8]]                               # value_as_limbs
    for i in range(N):
        b.when(is_real).assert_eq(x3_ins[i], value_limbs[i])
    for i in range(N):
        b.when(is_real).assert_eq(y3_ins[i], value_limbs[N + i])
    for i in range(16):      # eval_memory_access_slice (air/memory.rs:65-82)
        r.eval_memory_access(l[SHARD], l[CLK], l[Q_PTR] + 4 * i, q_access[i][0:4], q_access[i], is_real)
    for i in range(16):
        r.eval_memory_access(l[SHARD], l[CLK] + 1, l[P_PTR] + 4 * i, p_access[i][0:4], p_access[i][4:13], is_real)
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_ED_ADD & 0xffff), l[P_PTR], l[Q_PTR]]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))


def _field_op(r: _Rec, cols, a, bb, op, modulus, n_limbs, witness_offset, is_real):
    """FieldOpCols::eval_with_modulus (operations/field/field_op.rs:296-331): Add / Mul check a op b = result; Sub / Div swap the roles —
    result op' b = a with op' = Add / Mul — so `a` is what the identity's right-hand side holds. Operands may be shorter than a field
    element (the constant polynomials [0] and [1])."""
    b = r.b
    result = cols[0:n_limbs]
    lhs, rhs = (a, result) if op in ("add", "mul") else (result, a)
    p_op = _poly_add(lhs, bb) if op in ("add", "sub") else _poly_mul(lhs, bb)
    van = _poly_sub(b, _poly_sub(b, p_op, rhs), _poly_mul(cols[n_limbs:2 * n_limbs], modulus))
    _field_gadget(r, van, cols, n_limbs, witness_offset, is_real)


def _field_lt(r: _Rec, cols, lhs, rhs, n_limbs, is_real):
    """FieldLtCols::eval (operations/field/range.rs:63-139): cols = byte_flags(N), lhs_comparison_byte, rhs_comparison_byte; the flag marks
    the most significant byte where lhs < rhs, every byte above it is equal."""
    b = r.b
    flags, lhs_byte, rhs_byte = cols[0:n_limbs], cols[n_limbs], cols[n_limbs + 1]
    total = b.const(0)
    for f in flags:
        b.when(is_real).assert_bool(f)
        total = total + f
    b.when(is_real).assert_one(total)
    visited, lhs_cmp, rhs_cmp = b.const(0), b.const(0), b.const(0)
    for i in reversed(range(n_limbs)):
        visited = visited + flags[i]
        lhs_cmp = lhs_cmp + lhs[i] * flags[i]
        rhs_cmp = rhs_cmp + flags[i] * rhs[i]
        b.when(is_real).when_not(visited).assert_eq(lhs[i], rhs[i])
    b.when(is_real).assert_eq(lhs_byte, lhs_cmp)
    b.when(is_real).assert_eq(rhs_byte, rhs_cmp)
    r.send_byte(B_LTU, 1, lhs_byte, rhs_byte, is_real)


def _ed_decompress(r: _Rec):
    """EdDecompressCols::eval (syscall/precompiles/edwards/ed_decompress.rs:104-190): x = sqrt((y^2 - 1) / (d y^2 + 1)), the even root, negated
    when the sign bit asks for the odd one; y is read at ptr + 32, x is written at ptr, both at clk."""
    l, b = r.local, r.b
    N, G = 32, 188
    IS_REAL, SHARD, CLK, PTR, SIGN, X_ACCESS, Y_ACCESS, Y_RANGE, YY, U, DYY, V, U_DIV_V, X_MULT, X_RANGE, X_LSB, NEG_X = (
        0, 1, 2, 3, 4, 5, 109, 181, 215, 403, 591, 779, 967, 1155, 1343, 1377, 1378)
    is_real, sign = l[IS_REAL], l[SIGN]
    x_access = [l[X_ACCESS + 13 * i:X_ACCESS + 13 * i + 13] for i in range(8)]
    y_access = [l[Y_ACCESS + 9 * i:Y_ACCESS + 9 * i + 9] for i in range(8)]
    modulus = _limbs_of_const(b, ED25519_P, N)
    d_const = _limbs_of_const(b, ED25519_D, N)
    one, zero = [b.const(1)], [b.const(0)]
    grp = lambda base: l[base:base + G]      # noqa: E731
    b.assert_bool(sign)
    y = [c for a in y_access for c in a[0:4]]
    _field_lt(r, l[Y_RANGE:Y_RANGE + 34], y, modulus, N, is_real)
    _field_op(r, grp(YY), y, y, "mul", modulus, N, 1 << 14, is_real)
    _field_op(r, grp(U), l[YY:YY + N], one, "sub", modulus, N, 1 << 14, is_real)
    _field_op(r, grp(DYY), d_const, l[YY:YY + N], "mul", modulus, N, 1 << 14, is_real)
    _field_op(r, grp(V), one, l[DYY:DYY + N], "add", modulus, N, 1 << 14, is_real)
    _field_op(r, grp(U_DIV_V), l[U:U + N], l[V:V + N], "div", modulus, N, 1 << 14, is_real)
    # FieldSqrtCols::eval (operations/field/field_sqrt.rs:88-131): the multiplication's result columns hold the root; its product is the input
    sqrt = l[X_MULT:X_MULT + N]
    mult = list(l[U_DIV_V:U_DIV_V + N]) + list(l[X_MULT + N:X_MULT + G])          # `multiplication.result = *a`
    _field_op(r, mult, sqrt, sqrt, "mul", modulus, N, 1 << 14, is_real)
    _field_lt(r, l[X_RANGE:X_RANGE + 34], sqrt, modulus, N, is_real)
    r.slice_range_check_u8(sqrt, is_real)
    b.assert_bool(l[X_LSB])
    b.when(is_real).assert_eq(l[X_LSB], 0)
    r.send_byte(B_AND, l[X_LSB], sqrt[0], 1, is_real)
    _field_op(r, grp(NEG_X), zero, sqrt, "sub", modulus, N, 1 << 14, is_real)
    for i in range(8):      # eval_memory_access_slice
        r.eval_memory_access(l[SHARD], l[CLK], l[PTR] + 4 * i, x_access[i][0:4], x_access[i][4:13], is_real)
    for i in range(8):
        r.eval_memory_access(l[SHARD], l[CLK], l[PTR] + 32 + 4 * i, y_access[i][0:4], y_access[i], is_real)
    x_limbs = [c for a in x_access for c in a[4:8]]
    for i in range(N):
        b.when(is_real).when(sign).assert_eq(l[NEG_X + i], x_limbs[i])
    for i in range(N):
        b.when(is_real).when_not(sign).assert_eq(sqrt[i], x_limbs[i])
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_ED_DECOMPRESS & 0xffff), l[PTR], sign]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))


def record_ed_decompress_constraints() -> _Rec:
    r = _Rec(E.ED_DECOMPRESS_WIDTH)
    _ed_decompress(r)
    return r


def record_ed_decompress_chip(log_height: int) -> RecordedChip:
    """The EdDecompress precompile (crates/core/machine/src/syscall/precompiles/edwards/ed_decompress.rs): one Ed25519 point decompression per
    row, 1566 columns; local_only (:262-264)."""
    return _finish(record_ed_decompress_constraints(), "EdDecompress", log_height, E.ED_DECOMPRESS_WIDTH, True)


def _weierstrass(r: _Rec, curve: str, double: bool):
    """WeierstrassAddAssignChip::eval (syscall/precompiles/weierstrass/weierstrass_add.rs:283-404) / WeierstrassDoubleAssignChip::eval
    (weierstrass_double.rs:318-462) over the curve's base field: slope = (q.y - p.y) / (q.x - p.x), or (3 p.x^2 + a) / (2 p.y) for a doubling;
    x3 = slope^2 - p.x - q.x; y3 = slope (p.x - x3) - p.y; every step a FieldOpCols. p is overwritten (at clk + 1 for an addition, at clk for
    a doubling)."""
    c = E.WEIERSTRASS_CURVES[curve]
    l, b = r.local, r.b
    N, W, off = c["n_limbs"], c["n_limbs"] // 2, c["witness_offset"]
    G = 6 * N - 4
    modulus = _limbs_of_const(b, c["p"], N)
    IS_REAL, SHARD, CLK, P_PTR = 0, 1, 2, 3
    Q_PTR = None if double else 4
    P_ACCESS = 4 if double else 5
    Q_ACCESS = None if double else P_ACCESS + 13 * W
    GADGETS = P_ACCESS + 13 * W + (0 if double else 9 * W)
    is_real = l[IS_REAL]
    p_access = [l[P_ACCESS + 13 * i:P_ACCESS + 13 * i + 13] for i in range(W)]
    px, py = [x for a in p_access[:W // 2] for x in a[0:4]], [x for a in p_access[W // 2:] for x in a[0:4]]
    col = lambda k: l[GADGETS + G * k:GADGETS + G * k + G]      # noqa: E731
    res = lambda k: l[GADGETS + G * k:GADGETS + G * k + N]      # noqa: E731
    op = lambda k, a, bb, kind: _field_op(r, col(k), a, bb, kind, modulus, N, off, is_real)      # noqa: E731
    if not double:
        q_access = [l[Q_ACCESS + 9 * i:Q_ACCESS + 9 * i + 9] for i in range(W)]
        qx, qy = [x for a in q_access[:W // 2] for x in a[0:4]], [x for a in q_access[W // 2:] for x in a[0:4]]
        SLOPE_DEN, SLOPE_NUM, SLOPE, SLOPE_SQ, PX_PLUS_QX, X3, PX_MINUS_X, Y3, SLOPE_TIMES = range(9)
        op(SLOPE_NUM, qy, py, "sub")
        op(SLOPE_DEN, qx, px, "sub")
        op(SLOPE, res(SLOPE_NUM), res(SLOPE_DEN), "div")
        op(SLOPE_SQ, res(SLOPE), res(SLOPE), "mul")
        op(PX_PLUS_QX, px, qx, "add")
        op(X3, res(SLOPE_SQ), res(PX_PLUS_QX), "sub")
    else:
        SLOPE_DEN, SLOPE_NUM, SLOPE, PX_SQ, PX_SQ_3, SLOPE_SQ, PX_PLUS_QX, X3, PX_MINUS_X, Y3, SLOPE_TIMES = range(11)
        op(PX_SQ, px, px, "mul")
        op(PX_SQ_3, res(PX_SQ), _limbs_of_const(b, 3, N), "mul")
        op(SLOPE_NUM, _limbs_of_const(b, c["a"], N), res(PX_SQ_3), "add")
        op(SLOPE_DEN, _limbs_of_const(b, 2, N), py, "mul")
        op(SLOPE, res(SLOPE_NUM), res(SLOPE_DEN), "div")
        op(SLOPE_SQ, res(SLOPE), res(SLOPE), "mul")
        op(PX_PLUS_QX, px, px, "add")
        op(X3, res(SLOPE_SQ), res(PX_PLUS_QX), "sub")
    op(PX_MINUS_X, px, res(X3), "sub")
    op(SLOPE_TIMES, res(SLOPE), res(PX_MINUS_X), "mul")
    op(Y3, res(SLOPE_TIMES), py, "sub")
    for i in range(N):
        b.when(is_real).assert_eq(res(X3)[i], p_access[i // 4][4 + i % 4])
        b.when(is_real).assert_eq(res(Y3)[i], p_access[W // 2 + i // 4][4 + i % 4])
    if not double:
        for i in range(W):
            r.eval_memory_access(l[SHARD], l[CLK], l[Q_PTR] + 4 * i, q_access[i][0:4], q_access[i], is_real)
    for i in range(W):
        r.eval_memory_access(l[SHARD], l[CLK] + (0 if double else 1), l[P_PTR] + 4 * i, p_access[i][0:4], p_access[i][4:13], is_real)
    code = c["double" if double else "add"] & 0xffff
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(code), l[P_PTR], b.const(0) if double else l[Q_PTR]]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))


def record_weierstrass_constraints(curve: str, double: bool) -> _Rec:
    r = _Rec(E.weierstrass_widths(curve)[1 if double else 0])
    _weierstrass(r, curve, double)
    return r


def record_weierstrass_chip(curve: str, double: bool, log_height: int) -> RecordedChip:
    """<Curve>AddAssign / <Curve>DoubleAssign for Secp256k1, Secp256r1, Bn254, Bls12381 (crates/core/machine/src/syscall/precompiles/weierstrass/):
    one point operation per row; local_only as the reference declares them."""
    name = curve + ("DoubleAssign" if double else "AddAssign")
    return _finish(record_weierstrass_constraints(curve, double), name, log_height, E.weierstrass_widths(curve)[1 if double else 0], True)


def _weierstrass_decompress(r: _Rec, curve: str):
    """WeierstrassDecompressChip::eval (syscall/precompiles/weierstrass/weierstrass_decompress.rs:305-527): y = sqrt(x^3 + a x + b) — a FieldLtCols
    of x, two products, an inner product (a, b) . (x, 1), a sum, a FieldSqrtCols, the root's negative — and the choice between the root and its
    negative: by parity of the root against the sign bit (Secp256k1, Secp256r1), or by comparing the two as integers (Bls12381:
    LexicographicChoiceCols). x is read at ptr + N, y written at ptr, both at clk."""
    c = E.WEIERSTRASS_CURVES[curve]
    lexicographic = E.WEIERSTRASS_DECOMPRESS[curve]["lexicographic"]
    l, b = r.local, r.b
    N, W, off = c["n_limbs"], c["n_limbs"] // 4, c["witness_offset"]
    G = 6 * N - 4
    modulus = _limbs_of_const(b, c["p"], N)
    IS_REAL, SHARD, CLK, PTR, SIGN, X_ACCESS = 0, 1, 2, 3, 4, 5
    Y_ACCESS = X_ACCESS + 9 * W
    RANGE_X = Y_ACCESS + 13 * W
    X_2 = RANGE_X + N + 2
    X_3, AX_PLUS_B, X_3_PLUS = X_2 + G, X_2 + 2 * G, X_2 + 3 * G
    Y_MULT = X_2 + 4 * G
    Y_RANGE, Y_LSB, NEG_Y = Y_MULT + G, Y_MULT + G + N + 2, Y_MULT + G + N + 3
    CHOICE = NEG_Y + G
    is_real, sign = l[IS_REAL], l[SIGN]
    x_access = [l[X_ACCESS + 9 * i:X_ACCESS + 9 * i + 9] for i in range(W)]
    y_access = [l[Y_ACCESS + 13 * i:Y_ACCESS + 13 * i + 13] for i in range(W)]
    grp = lambda base: l[base:base + G]      # noqa: E731
    b.assert_bool(sign)
    x = [v for a in x_access for v in a[0:4]]                                   # limbs_from_prev_access of a read: its value
    _field_lt(r, l[RANGE_X:RANGE_X + N + 2], x, modulus, N, is_real)
    _field_op(r, grp(X_2), x, x, "mul", modulus, N, off, is_real)
    _field_op(r, grp(X_3), l[X_2:X_2 + N], x, "mul", modulus, N, off, is_real)
    # FieldInnerProductCols::eval (field_inner_product.rs:82-120) of (a, b) and (x, 1)
    cols = grp(AX_PLUS_B)
    inner = _poly_add(_poly_mul(_limbs_of_const(b, c["a"], N), x), _poly_mul(_limbs_of_const(b, c["b"], N), _limbs_of_const(b, 1, N)))
    _field_gadget(r, _poly_sub(b, _poly_sub(b, inner, cols[0:N]), _poly_mul(cols[N:2 * N], modulus)), cols, N, off, is_real)
    _field_op(r, grp(X_3_PLUS), l[X_3:X_3 + N], l[AX_PLUS_B:AX_PLUS_B + N], "add", modulus, N, off, is_real)
    sqrt, neg_y = l[Y_MULT:Y_MULT + N], l[NEG_Y:NEG_Y + N]
    _field_op(r, grp(NEG_Y), [b.const(0)], sqrt, "sub", modulus, N, off, is_real)
    # FieldSqrtCols::eval (field_sqrt.rs:88-131): the multiplication's result columns hold the root, its product is the input
    mult = list(l[X_3_PLUS:X_3_PLUS + N]) + list(l[Y_MULT + N:Y_MULT + G])
    _field_op(r, mult, sqrt, sqrt, "mul", modulus, N, off, is_real)
    _field_lt(r, l[Y_RANGE:Y_RANGE + N + 2], sqrt, modulus, N, is_real)
    r.slice_range_check_u8(sqrt, is_real)
    b.assert_bool(l[Y_LSB])
    b.when(is_real).assert_eq(l[Y_LSB], l[Y_LSB])                               # `is_odd` is the column itself here (:372)
    r.send_byte(B_AND, l[Y_LSB], sqrt[0], 1, is_real)
    y_limbs = [v for a in y_access for v in a[4:8]]                             # limbs_from_access of the write: the new value
    if not lexicographic:
        for i in range(N):
            b.when(is_real).when(l[Y_LSB] - (b.const(1) - sign)).assert_eq(sqrt[i], y_limbs[i])
        for i in range(N):
            b.when(is_real).when(l[Y_LSB] - sign).assert_eq(neg_y[i], y_limbs[i])
    else:
        CMP, NEG_RANGE = CHOICE, CHOICE + N + 2
        is_eq, when_sqrt_lt, when_neg_lt = l[CHOICE + 2 * (N + 2)], l[CHOICE + 2 * (N + 2) + 1], l[CHOICE + 2 * (N + 2) + 2]
        _field_lt(r, l[NEG_RANGE:NEG_RANGE + N + 2], neg_y, modulus, N, is_real)
        b.assert_bool(is_eq)
        b.assert_bool(when_sqrt_lt)
        b.assert_bool(when_neg_lt)
        b.when(is_real).assert_one(when_sqrt_lt + when_neg_lt)
        for i in range(N):
            b.when(is_real).when(is_eq).assert_eq(sqrt[i], y_limbs[i])
        for i in range(N):
            b.when(is_real).when_not(is_eq).assert_eq(neg_y[i], y_limbs[i])
        b.when_not(is_real).assert_zero(when_sqrt_lt)
        b.when_not(is_real).assert_zero(when_neg_lt)
        b.when(is_real).when(sign).assert_eq(is_eq, when_neg_lt)
        b.when(is_real).when_not(sign).assert_eq(is_eq, when_sqrt_lt)
        _field_lt(r, l[CMP:CMP + N + 2], sqrt, neg_y, N, when_sqrt_lt)
        _field_lt(r, l[CMP:CMP + N + 2], neg_y, sqrt, N, when_neg_lt)
    for i in range(W):
        r.eval_memory_access(l[SHARD], l[CLK], l[PTR] + (4 * i + N), x_access[i][0:4], x_access[i], is_real)
    for i in range(W):
        r.eval_memory_access(l[SHARD], l[CLK], l[PTR] + 4 * i, y_access[i][0:4], y_access[i][4:13], is_real)
    code = E.WEIERSTRASS_DECOMPRESS[curve]["code"] & 0xffff
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(code), l[PTR], sign]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))


def record_weierstrass_decompress_constraints(curve: str) -> _Rec:
    r = _Rec(E.weierstrass_decompress_width(curve))
    _weierstrass_decompress(r, curve)
    return r


def record_weierstrass_decompress_chip(curve: str, log_height: int) -> RecordedChip:
    """Secp256k1Decompress / Secp256r1Decompress / Bls12381Decompress (crates/core/machine/src/syscall/precompiles/weierstrass/weierstrass_decompress.rs):
    one decompression per row; local_only (:299-301)."""
    return _finish(record_weierstrass_decompress_constraints(curve), curve + "Decompress", log_height, E.weierstrass_decompress_width(curve), True)


def _uint256_mul(r: _Rec):
    """Uint256MulChip::eval (syscall/precompiles/uint256/air.rs:216-328): output = x * y mod m through one FieldOpCols over U256Field whose modulus
    polynomial is the limbs read from memory, or x^32 (2^256) when an IsZeroOperation on the sum of those limbs says they are all zero; the
    result is below the modulus (FieldLtCols) when there is one. x is written at clk + 1, y and the modulus (contiguous at y_ptr) are read at clk."""
    l, b = r.local, r.b
    N, NW = 32, 63
    SHARD, CLK, X_PTR, Y_PTR, X_MEM, Y_MEM, M_MEM, IS_ZERO, NOT_ZERO, OUTPUT, RANGE, IS_REAL = 0, 1, 2, 3, 4, 108, 180, 252, 254, 255, 445, 479
    is_real = l[IS_REAL]
    x_mem = [l[X_MEM + 13 * i:X_MEM + 13 * i + 13] for i in range(8)]
    y_mem = [l[Y_MEM + 9 * i:Y_MEM + 9 * i + 9] for i in range(8)]
    m_mem = [l[M_MEM + 9 * i:M_MEM + 9 * i + 9] for i in range(8)]
    x = [v for a in x_mem for v in a[0:4]]              # limbs_from_prev_access
    y = [v for a in y_mem for v in a[0:4]]
    m = [v for a in m_mem for v in a[0:4]]
    byte_sum = b.const(0)
    for limb in m:
        byte_sum = byte_sum + limb
    _is_zero(b, byte_sum, l[IS_ZERO:IS_ZERO + 2], is_real)
    is_zero = l[IS_ZERO + 1]
    p_modulus = [limb * (b.const(1) - is_zero) for limb in m] + [b.const(1) * is_zero]      # the limbs, or x^32
    cols = l[OUTPUT:OUTPUT + 2 * N + 2 * NW]
    van = _poly_sub(b, _poly_sub(b, _poly_mul(x, y), cols[0:N]), _poly_mul(cols[N:2 * N], p_modulus))
    _field_gadget(r, van, cols, N, 1 << 14, is_real, NW)
    _field_lt(r, l[RANGE:RANGE + N + 2], cols[0:N], m, N, l[NOT_ZERO])
    b.assert_eq(l[NOT_ZERO], is_real * (b.const(1) - is_zero))
    value_limbs = [v for a in x_mem for v in a[4:8]]
    for i in range(N):
        b.when(is_real).assert_eq(cols[i], value_limbs[i])
    for i in range(8):
        r.eval_memory_access(l[SHARD], l[CLK] + 1, l[X_PTR] + 4 * i, x_mem[i][0:4], x_mem[i][4:13], is_real)
    for i, acc in enumerate(y_mem + m_mem):
        r.eval_memory_access(l[SHARD], l[CLK], l[Y_PTR] + 4 * i, acc[0:4], acc, is_real)
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_UINT256_MUL & 0xffff), l[X_PTR], l[Y_PTR]]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))
    b.assert_bool(is_real)


def record_uint256_mul_constraints() -> _Rec:
    r = _Rec(E.UINT256_MUL_WIDTH)
    _uint256_mul(r)
    return r


def record_uint256_mul_chip(log_height: int) -> RecordedChip:
    """Uint256MulMod (crates/core/machine/src/syscall/precompiles/uint256/air.rs): one modular multiplication of 256-bit integers per row; local_only (:205-207)."""
    return _finish(record_uint256_mul_constraints(), "Uint256MulMod", log_height, E.UINT256_MUL_WIDTH, True)


def _u256x2048_mul(r: _Rec):
    """U256x2048MulChip::eval (syscall/precompiles/u256x2048_mul/air.rs:250-398): eight FieldOpCols over U256Field with the modulus t^32, chained
    through their carries — a * b_i + carry_{i-1} = result_i + carry_i 2^256 — so that the results are the low 2048 bits and the last carry the
    high 256. lo_ptr and hi_ptr are read from registers $a2 and $a3; a and b are read at clk, lo and hi written at clk + 1."""
    l, b = r.local, r.b
    N, NW, G = 32, 63, 190
    SHARD, CLK, A_PTR, B_PTR, LO_PTR, HI_PTR, LO_PTR_MEM, HI_PTR_MEM, A_MEM, B_MEM, LO_MEM, HI_MEM, GADGETS, IS_REAL = 0, 1, 2, 3, 4, 5, 6, 15, 24, 96, 672, 1504, 1608, 3128
    is_real = l[IS_REAL]
    reads = lambda base, n: [l[base + 9 * i:base + 9 * i + 9] for i in range(n)]          # noqa: E731
    writes = lambda base, n: [l[base + 13 * i:base + 13 * i + 13] for i in range(n)]      # noqa: E731
    lo_ptr_mem, hi_ptr_mem = l[LO_PTR_MEM:LO_PTR_MEM + 9], l[HI_PTR_MEM:HI_PTR_MEM + 9]
    a_mem, b_mem, lo_mem, hi_mem = reads(A_MEM, 8), reads(B_MEM, 64), writes(LO_MEM, 64), writes(HI_MEM, 8)
    b.assert_bool(is_real)
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_U256XU2048_MUL & 0xffff), l[A_PTR], l[B_PTR]]],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL))
    r.eval_memory_access(l[SHARD], l[CLK], b.const(E.REG_A2), lo_ptr_mem[0:4], lo_ptr_mem, is_real)
    r.eval_memory_access(l[SHARD], l[CLK], b.const(E.REG_A3), hi_ptr_mem[0:4], hi_ptr_mem, is_real)
    for i in range(8):
        r.eval_memory_access(l[SHARD], l[CLK], l[A_PTR] + 4 * i, a_mem[i][0:4], a_mem[i], is_real)
    for i in range(64):
        r.eval_memory_access(l[SHARD], l[CLK], l[B_PTR] + 4 * i, b_mem[i][0:4], b_mem[i], is_real)
    for i in range(64):
        r.eval_memory_access(l[SHARD], l[CLK] + 1, l[LO_PTR] + 4 * i, lo_mem[i][0:4], lo_mem[i][4:13], is_real)
    for i in range(8):
        r.eval_memory_access(l[SHARD], l[CLK] + 1, l[HI_PTR] + 4 * i, hi_mem[i][0:4], hi_mem[i][4:13], is_real)
    a = [v for acc in a_mem for v in acc[0:4]]
    modulus = [b.const(0)] * N + [b.const(1)]
    carry = [b.const(0)]
    for g in range(8):      # FieldOpCols::eval_mul_and_carry (field_op.rs:281-301)
        cols = l[GADGETS + G * g:GADGETS + G * g + G]
        b_g = [v for acc in b_mem[8 * g:8 * g + 8] for v in acc[0:4]]
        op = _poly_add(_poly_mul(a, b_g), carry)
        van = _poly_sub(b, _poly_sub(b, op, cols[0:N]), _poly_mul(cols[N:2 * N], modulus))
        _field_gadget(r, van, cols, N, 1 << 14, is_real, NW)
        carry = cols[N:2 * N]
    hi_limbs = [v for acc in hi_mem for v in acc[4:8]]
    for i in range(N):
        b.when(is_real).assert_eq(carry[i], hi_limbs[i])
    for g in range(8):
        lo_limbs = [v for acc in lo_mem[8 * g:8 * g + 8] for v in acc[4:8]]
        for i in range(N):
            b.when(is_real).assert_eq(l[GADGETS + G * g + i], lo_limbs[i])
    reduce = lambda w: w[0] + w[1] * 256 + w[2] * 65536 + w[3] * 16777216      # noqa: E731
    b.when(is_real).assert_eq(l[LO_PTR], reduce(lo_ptr_mem[0:4]))
    b.when(is_real).assert_eq(l[HI_PTR], reduce(hi_ptr_mem[0:4]))


def record_u256x2048_mul_constraints() -> _Rec:
    r = _Rec(E.U256X2048_MUL_WIDTH)
    _u256x2048_mul(r)
    return r


def record_u256x2048_mul_chip(log_height: int) -> RecordedChip:
    """U256XU2048Mul (crates/core/machine/src/syscall/precompiles/u256x2048_mul/air.rs): one 256 x 2048-bit product per row. The reference does not
    override local_only for this chip (:231-240), so it is not."""
    return _finish(record_u256x2048_mul_constraints(), "U256XU2048Mul", log_height, E.U256X2048_MUL_WIDTH, False)


def _boolean_circuit_garble(r: _Rec):
    """BooleanCircuitGarbleChip::eval (syscall/precompiles/boolean_circuit_garble/air.rs:25-231): a call is a header row (the reads of the gate count
    and of delta) followed by one row per gate (seventeen reads; three chained XorOperations per ciphertext word; IsEqualWordOperations against the
    expected ciphertext, selected by the gate type; a running conjunction in checks[3]); the last gate's row writes the result. Transcribed as
    written, including what it leaves open (the running conjunction is not tied to the value written; gates_num and delta are tied to memory
    on the table's first row only)."""
    l, n, b = r.local, r.next, r.b
    (SHARD, CLK, IS_REAL, INPUT, OUTPUT, IS_FIRST_ROW, IS_GATE, IS_FIRST_GATE, IS_LAST_GATE, NOT_LAST_GATE, GATE_TYPE, GATE_ID, GATES_NUM, DELTA, MEM, RESULT_MEM,
     AUX1, AUX2, AUX3, IS_EQ, CHECKS) = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 30, 183, 196, 212, 228, 244, 288)
    mem = [l[MEM + 9 * i:MEM + 9 * i + 9] for i in range(17)]
    delta = [l[DELTA + 4 * i:DELTA + 4 * i + 4] for i in range(4)]
    word = lambda base, i: l[base + 4 * i:base + 4 * i + 4]      # noqa: E731
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], b.const(E.SYS_BOOLEAN_CIRCUIT_GARBLE & 0xffff), l[INPUT], l[OUTPUT]]],
                                 air.to_virtual_pair(l[IS_FIRST_ROW]), air.KIND_SYSCALL))
    # eval_flags (:50-65)
    b.assert_bool(l[IS_REAL])
    b.assert_bool(l[IS_FIRST_GATE])
    b.assert_bool(l[NOT_LAST_GATE])
    b.assert_bool(l[IS_GATE])
    b.assert_zero(l[IS_LAST_GATE] * l[IS_FIRST_GATE])
    b.when(l[IS_GATE]).assert_one(l[IS_LAST_GATE] + l[NOT_LAST_GATE])
    b.assert_bool(l[GATE_TYPE])
    b.assert_bool(l[GATE_TYPE + 1])
    b.assert_eq(l[GATE_TYPE] + l[GATE_TYPE + 1], l[IS_GATE])
    b.when(l[IS_REAL]).assert_one(l[IS_FIRST_ROW] + l[IS_GATE])
    # eval_memory_access (:67-113)
    r.eval_memory_access(l[SHARD], l[CLK], l[INPUT], mem[0][0:4], mem[0], l[IS_FIRST_ROW])
    for i in range(4):
        r.eval_memory_access(l[SHARD], l[CLK], l[INPUT] + (4 + 4 * i), mem[i + 1][0:4], mem[i + 1], l[IS_FIRST_ROW])
    for i in range(17):
        r.eval_memory_access(l[SHARD], l[CLK], l[INPUT] + 4 * i, mem[i][0:4], mem[i], l[IS_GATE])
    res = l[RESULT_MEM:RESULT_MEM + 13]
    r.eval_memory_access(l[SHARD], l[CLK], l[OUTPUT], res[0:4], res[4:13], l[IS_LAST_GATE])
    # eval_logic_check (:115-180)
    for i in range(4):
        _bitwise_op(r, B_XOR, mem[1 + i][0:4], mem[5 + i][0:4], word(AUX1, i), l[IS_GATE])
        _bitwise_op(r, B_XOR, word(AUX1, i), mem[9 + i][0:4], word(AUX2, i), l[IS_GATE])
        _bitwise_op(r, B_XOR, word(AUX2, i), delta[i], word(AUX3, i), l[IS_GATE])
    for i in range(4):
        want, cols = mem[13 + i][0:4], l[IS_EQ + 11 * i:IS_EQ + 11 * i + 11]
        for aux, flag in ((AUX2, l[GATE_TYPE]), (AUX3, l[GATE_TYPE + 1])):
            b.assert_bool(flag)                                            # IsEqualWordOperation::eval (is_equal_word.rs:31-47)
            _is_zero_word(b, [word(aux, i)[k] - want[k] for k in range(4)], cols, flag)
    eq = lambda i: l[IS_EQ + 11 * i + 10]      # noqa: E731
    b.when(l[IS_GATE]).assert_eq(l[CHECKS], eq(0) * eq(1))
    b.when(l[IS_GATE]).assert_eq(l[CHECKS + 1], eq(2) * l[CHECKS])
    b.when(l[IS_GATE]).assert_eq(l[CHECKS + 2], eq(3) * l[CHECKS + 1])
    b.when(l[NOT_LAST_GATE]).assert_eq(n[CHECKS + 3], l[CHECKS + 3] * n[CHECKS + 2])
    # eval_transition (:182-230)
    num_gates = mem[0][0] + mem[0][1] * 256 + mem[0][2] * 65536 + mem[0][3] * 16777216
    b.when_first_row().assert_eq(l[GATES_NUM], num_gates)
    for i in range(4):
        for j in range(4):
            b.when_first_row().assert_eq(delta[i][j], mem[i + 1][j])
    gate_type_value = l[GATE_TYPE] * 0 + l[GATE_TYPE + 1]
    b.when(l[IS_GATE]).assert_eq(gate_type_value * E.GARBLE_OR_GATE, num_gates)
    b.when(l[IS_FIRST_GATE]).assert_zero(l[GATE_ID])
    b.when(l[IS_LAST_GATE]).assert_eq(l[GATES_NUM] - 1, l[GATE_ID])
    b.when(l[NOT_LAST_GATE]).assert_eq(l[GATE_ID] + 1, n[GATE_ID])
    b.when(l[NOT_LAST_GATE] * l[IS_GATE]).assert_eq(l[INPUT] + 68, n[INPUT])
    b.when(l[NOT_LAST_GATE] * l[IS_GATE]).assert_eq(l[GATES_NUM], n[GATES_NUM])
    for i in range(4):
        for j in range(4):
            b.when(l[NOT_LAST_GATE] * l[IS_GATE]).assert_eq(delta[i][j], n[DELTA + 4 * i + j])


def record_boolean_circuit_garble_constraints() -> _Rec:
    r = _Rec(E.GARBLE_WIDTH)
    _boolean_circuit_garble(r)
    return r


def record_boolean_circuit_garble_chip(log_height: int) -> RecordedChip:
    """BooleanCircuitGarble (crates/core/machine/src/syscall/precompiles/boolean_circuit_garble/): 1 + num_gates rows per call, 292 columns; constraints
    between consecutive rows, not local_only (the reference does not override it)."""
    return _finish(record_boolean_circuit_garble_constraints(), "BooleanCircuitGarble", log_height, E.GARBLE_WIDTH, False)


def _gt_bytes(r: _Rec, a, bw, is_real, cols):
    """GtColsBytes::eval (operations/cmp.rs:95-169): cols = byte_flags(4), a_comparison_byte, b_comparison_byte, result, has_comparison."""
    b = r.b
    flags, a_byte, b_byte, result, has = cols[0:4], cols[4], cols[5], cols[6], cols[7]
    r.slice_range_check_u8(a, is_real)
    r.slice_range_check_u8(bw, is_real)
    total = b.const(0)
    for f in flags:
        b.when(is_real).assert_bool(f)
        total = total + f
    b.when(is_real).assert_bool(total)
    b.when(is_real).assert_eq(has, total)
    b.when(is_real).assert_bool(has)
    visited, first_gt, b_cmp = b.const(0), b.const(0), b.const(0)
    for i in reversed(range(4)):
        visited = visited + flags[i]
        first_gt = first_gt + a[i] * flags[i]
        b_cmp = b_cmp + bw[i] * flags[i]
        b.when_not(visited).when(is_real).assert_eq(a[i], bw[i])
    b.when(is_real).assert_eq(a_byte, first_gt)
    b.when(is_real).assert_eq(b_byte, b_cmp)
    r.send_byte(B_LTU, result, b_byte, a_byte, is_real)
    r.send_byte(B_LTU, 1 - result, a_byte, b_byte, has)


def _sys_linux(r: _Rec):
    """SysLinuxChip::eval (syscall/precompiles/sys_linux/air.rs:29-477): the syscall id and a0 / a1 are decoded with IsZeroOperations into branch
    selectors (mmap / mmap2, clone, exit_group, brk, fcntl, read, write; anything else is a no-op); each branch fixes `result` (the new $v0) and
    the value written to $a3; brk, write and mmap(0) share one memory access (`inorout`: register BRK, $a2, register HEAP)."""
    l, b = r.local, r.b
    (SHARD, CLK, ID, A0, A1, RESULT, INOROUT, OUTPUT, D_MMAP, D_MMAP2, D_CLONE, D_EXIT, D_BRK, D_FCNTL, D_READ, D_WRITE, IS_MMAP, D_A0_0, D_A0_1, D_A0_2, D_A1_1, D_A1_3,
     IS_MMAP_A0_0, IS_FCNTL_A1_1, IS_FCNTL_A1_3, LO_BITS, HI_BITS, PAGE_ZERO, MMAP_SIZE, SIZE_CARRY, HEAP_ADD, GT, IS_REAL) = (
        0, 1, 2, 3, 7, 11, 15, 28, 41, 43, 45, 47, 49, 51, 53, 55, 57, 58, 60, 62, 64, 66, 68, 69, 70, 71, 75, 79, 81, 85, 87, 94, 102)
    is_real = l[IS_REAL]
    a0, a1, result = l[A0:A0 + 4], l[A1:A1 + 4], l[RESULT:RESULT + 4]
    inorout, output = l[INOROUT:INOROUT + 13], l[OUTPUT:OUTPUT + 13]
    in_prev, in_value, out_value = inorout[0:4], inorout[4:8], output[4:8]
    sid = l[ID]
    for code, col in ((E.SYS_MMAP, D_MMAP), (E.SYS_MMAP2, D_MMAP2), (E.SYS_CLONE, D_CLONE), (E.SYS_EXT_GROUP, D_EXIT), (E.SYS_BRK, D_BRK), (E.SYS_FCNTL, D_FCNTL),
                      (E.SYS_READ, D_READ), (E.SYS_WRITE_LINUX, D_WRITE)):
        _is_zero(b, sid - code, l[col:col + 2], is_real)
    is_clone, is_exit, is_brk, is_fcntl, is_read, is_write = (l[c + 1] for c in (D_CLONE, D_EXIT, D_BRK, D_FCNTL, D_READ, D_WRITE))
    is_mmap = l[IS_MMAP]
    b.when(is_real).assert_eq(is_mmap, l[D_MMAP + 1] + l[D_MMAP2 + 1])
    b.assert_bool(is_mmap)
    is_nop = is_real - (is_mmap + is_clone + is_exit + is_brk + is_fcntl + is_read + is_write)
    b.when(is_real).assert_bool(is_nop)
    reduce = lambda w: w[0] + w[1] * 256 + w[2] * 65536 + w[3] * 16777216      # noqa: E731
    _is_zero(b, reduce(a0), l[D_A0_0:D_A0_0 + 2], is_real)
    _is_zero(b, reduce(a0) - 1, l[D_A0_1:D_A0_1 + 2], is_real)
    _is_zero(b, reduce(a0) - 2, l[D_A0_2:D_A0_2 + 2], is_real)
    _is_zero(b, reduce(a1) - 1, l[D_A1_1:D_A1_1 + 2], is_real)
    _is_zero(b, reduce(a1) - 3, l[D_A1_3:D_A1_3 + 2], is_real)
    is_a0_0, is_a0_1, is_a0_2, is_a1_1, is_a1_3 = (l[c + 1] for c in (D_A0_0, D_A0_1, D_A0_2, D_A1_1, D_A1_3))
    b.assert_eq(l[IS_MMAP_A0_0], is_mmap * is_a0_0)
    b.assert_eq(l[IS_FCNTL_A1_1], is_fcntl * is_a1_1)
    b.assert_eq(l[IS_FCNTL_A1_3], is_fcntl * is_a1_3)
    word_eq = lambda cond, x, y: [cond.assert_eq(x[i], y[i]) for i in range(4)]      # noqa: E731
    word_zero = lambda cond, x: [cond.assert_zero(x[i]) for i in range(4)]          # noqa: E731
    const_word = lambda v: [b.const((v >> (8 * i)) & 0xff) for i in range(4)]       # noqa: E731
    word_eq(b.when(is_brk + is_write), in_value, in_prev)
    # eval_brk (:213-241)
    r.eval_memory_access(l[SHARD], l[CLK], b.const(E.REG_BRK), in_prev, inorout[4:13], is_brk)
    _gt_bytes(r, a0, in_value, is_brk, l[GT:GT + 8])
    gt = l[GT + 6]
    word_eq(b.when(is_brk).when(gt), result, a0)
    word_eq(b.when(is_brk).when_not(gt), result, in_prev)
    word_zero(b.when(is_brk), out_value)
    # eval_clone (:243-254)
    word_zero(b.when(is_clone), out_value)
    b.when(is_clone).assert_one(result[0])
    for i in (1, 2, 3):
        b.when(is_clone).assert_zero(result[i])
    # eval_exit_group (:373-381)
    word_zero(b.when(is_exit), out_value)
    word_zero(b.when(is_exit), result)
    # eval_fnctl (:383-436)
    any_fd = is_a0_0 + is_a0_1 + is_a0_2
    f11, f13 = l[IS_FCNTL_A1_1], l[IS_FCNTL_A1_3]
    word_zero(b.when(f11).when(is_a0_0), result)
    word_eq(b.when(f11).when(is_a0_1), result, const_word(1))
    word_eq(b.when(f11).when(is_a0_2), result, const_word(2))
    word_eq(b.when(f11).when_not(any_fd), result, const_word(0xffffffff))
    word_zero(b.when(f13).when(is_a0_0), result)
    word_eq(b.when(f13).when(is_a0_1 + is_a0_2), result, const_word(1))
    word_eq(b.when(f13).when_not(any_fd), result, const_word(0xffffffff))
    word_eq(b.when(is_fcntl).when_not(is_a1_3 + is_a1_1), result, const_word(0xffffffff))
    word_zero(b.when(f13 + f11).when(any_fd), out_value)
    word_eq(b.when(f13 + f11).when_not(any_fd), out_value, const_word(9))
    word_eq(b.when(is_fcntl).when_not(is_a1_3 + is_a1_1), out_value, const_word(9))
    # eval_read (:438-455)
    word_zero(b.when(is_read).when(is_a0_0), result)
    word_zero(b.when(is_read).when(is_a0_0), out_value)
    word_eq(b.when(is_read).when_not(is_a0_0), result, const_word(0xffffffff))
    word_eq(b.when(is_read).when_not(is_a0_0), out_value, const_word(9))
    # eval_write (:457-471)
    r.eval_memory_access(l[SHARD], l[CLK], b.const(E.REG_A2), in_prev, inorout[4:13], is_write)
    word_eq(b.when(is_write), result, in_value)
    word_zero(b.when(is_write), out_value)
    # eval_mmap (:256-371)
    r.slice_range_check_u8(a0, is_mmap)
    r.slice_range_check_u8(a1, is_mmap)
    lo = b.const(0)
    for bit in range(4):
        b.when(is_mmap).assert_bool(l[LO_BITS + bit])
        lo = lo + l[LO_BITS + bit] * (1 << bit)
    hi = b.const(0)
    for bit in range(4):
        b.when(is_mmap).assert_bool(l[HI_BITS + bit])
        hi = hi + l[HI_BITS + bit] * (1 << bit)
    b.when(is_mmap).assert_eq(a1[1], lo + hi * 16)
    page_offset = a1[0] + lo * 256
    _is_zero(b, page_offset, l[PAGE_ZERO:PAGE_ZERO + 2], is_mmap)
    is_offset_0 = l[PAGE_ZERO + 1]
    mm = l[IS_MMAP_A0_0]
    not_aligned = mm * (1 - is_offset_0)
    size, carry = l[MMAP_SIZE:MMAP_SIZE + 4], l[SIZE_CARRY:SIZE_CARRY + 2]
    b.when(mm).assert_bool(carry[0])
    b.when(mm).assert_bool(carry[1])
    b.when(mm).assert_zero(size[0])
    b.when(mm).assert_eq(size[1], hi * 16 + not_aligned * 16 - carry[0] * 256)
    b.when(mm).assert_eq(size[2], a1[2] + carry[0] - carry[1] * 256)
    b.when(mm).assert_eq(size[3], a1[3] + carry[1])
    r.slice_range_check_u8(size, mm)
    _add_op(r, in_prev, size, l[HEAP_ADD:HEAP_ADD + 7], mm)
    word_eq(b.when(mm), in_value, l[HEAP_ADD:HEAP_ADD + 4])
    r.eval_memory_access(l[SHARD], l[CLK], b.const(E.REG_HEAP), in_prev, inorout[4:13], mm)
    word_eq(b.when(mm), in_prev, result)
    word_eq(b.when(is_mmap).when_not(is_a0_0), a0, result)
    word_zero(b.when(is_mmap), out_value)
    # eval_nop (:473-476)
    word_zero(b.when(is_nop), out_value)
    word_zero(b.when(is_nop), result)
    # the write of $a3, the syscall and its result
    r.eval_memory_access(l[SHARD], l[CLK], b.const(E.REG_A3), output[0:4], output[4:13], is_real)
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], sid, reduce(a0), reduce(a1)]], air.to_virtual_pair(is_real), air.KIND_SYSCALL))
    halves = lambda w: [w[0] + w[1] * 256, w[2] + w[3] * 256]      # noqa: E731
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK]] + halves(result) + halves(a0) + halves(a1)],
                                 air.to_virtual_pair(is_real), air.KIND_SYSCALL_RESULT))


def record_sys_linux_constraints() -> _Rec:
    r = _Rec(E.SYS_LINUX_WIDTH)
    _sys_linux(r)
    return r


def record_sys_linux_chip(log_height: int) -> RecordedChip:
    """SysLinux (crates/core/machine/src/syscall/precompiles/sys_linux/): one Linux syscall per row, 103 columns; not local_only (mod.rs leaves the default)."""
    return _finish(record_sys_linux_constraints(), "SysLinux", log_height, E.SYS_LINUX_WIDTH, False)


def _field_op_variable(r: _Rec, cols, a, bb, modulus, n_limbs, witness_offset, is_add, is_sub, is_mul, is_real):
    """FieldOpCols::eval_variable (operations/field/field_op.rs:227-261) with is_div = 0: the operation is chosen by flags, so the identity
    is the flag-weighted sum of the three."""
    b = r.b
    result = cols[0:n_limbs]
    scale = lambda poly, k: [x * k for x in poly]      # noqa: E731
    p_result = _poly_add(scale(result, is_add + is_mul), scale(a, is_sub))
    p_op = _poly_add(_poly_add(scale(_poly_add(a, bb), is_add), scale(_poly_add(result, bb), is_sub)), scale(_poly_mul(a, bb), is_mul))
    van = _poly_sub(b, _poly_sub(b, p_op, p_result), _poly_mul(cols[n_limbs:2 * n_limbs], modulus))
    _field_gadget(r, van, cols, n_limbs, witness_offset, is_real)


def _fp_tower(r: _Rec, field: str, kind: str):
    """FpOpChip::eval (syscall/precompiles/fptower/fp.rs:203-289), Fp2AddSubAssignChip::eval (fp2_addsub.rs:227-320), Fp2MulAssignChip::eval
    (fp2_mul.rs:244-360) over the base field of Bn254 or Bls12381; x is overwritten at clk + 1, y read at clk."""
    c = E.WEIERSTRASS_CURVES[field]
    l, b = r.local, r.b
    N, off = c["n_limbs"], c["witness_offset"]
    G = 6 * N - 4
    modulus = _limbs_of_const(b, c["p"], N)
    words = N // 4 if kind == "fp" else N // 2
    head = {"fp": 8, "fp2_addsub": 6, "fp2_mul": 5}[kind]
    X_ACCESS, Y_ACCESS, GADGETS = head, head + 13 * words, head + 22 * words
    is_real = l[0]
    x_access = [l[X_ACCESS + 13 * i:X_ACCESS + 13 * i + 13] for i in range(words)]
    y_access = [l[Y_ACCESS + 9 * i:Y_ACCESS + 9 * i + 9] for i in range(words)]
    prev = lambda acc, lo, hi: [x for a in acc[lo:hi] for x in a[0:4]]      # noqa: E731
    col = lambda k: l[GADGETS + G * k:GADGETS + G * k + G]                  # noqa: E731
    res = lambda k: l[GADGETS + G * k:GADGETS + G * k + N]                  # noqa: E731
    codes = E.FP_TOWER_CODES[field]
    if kind == "fp":
        SHARD, CLK, IS_ADD, IS_SUB, IS_MUL, X_PTR, Y_PTR = 1, 2, 3, 4, 5, 6, 7
        for f in (IS_ADD, IS_SUB, IS_MUL):
            b.assert_bool(l[f])
        b.assert_eq(l[IS_ADD] + l[IS_SUB] + l[IS_MUL], 1)
        _field_op_variable(r, col(0), prev(x_access, 0, words), prev(y_access, 0, words), modulus, N, off, l[IS_ADD], l[IS_SUB], l[IS_MUL], is_real)
        outs = [(0, 0)]
        syscall = l[IS_ADD] * (codes["fp_add"] & 0xffff) + l[IS_SUB] * (codes["fp_sub"] & 0xffff) + l[IS_MUL] * (codes["fp_mul"] & 0xffff)
    elif kind == "fp2_addsub":
        SHARD, CLK, IS_ADD, X_PTR, Y_PTR = 1, 2, 3, 4, 5
        b.assert_bool(l[IS_ADD])
        h = words // 2
        for k, (lo, hi) in enumerate(((0, h), (h, words))):
            _field_op_variable(r, col(k), prev(x_access, lo, hi), prev(y_access, lo, hi), modulus, N, off, l[IS_ADD], 1 - l[IS_ADD], b.const(0), is_real)
        outs = [(0, 0), (1, h)]
        syscall = l[IS_ADD] * (codes["fp2_add"] & 0xffff) + (1 - l[IS_ADD]) * (codes["fp2_sub"] & 0xffff)
    else:
        SHARD, CLK, X_PTR, Y_PTR = 1, 2, 3, 4
        h = words // 2
        px, py, qx, qy = prev(x_access, 0, h), prev(x_access, h, words), prev(y_access, 0, h), prev(y_access, h, words)
        A0B0, A1B1, A0B1, A1B0, C0, C1 = range(6)
        op = lambda k, a, bb, o: _field_op(r, col(k), a, bb, o, modulus, N, off, is_real)      # noqa: E731
        op(A0B0, px, qx, "mul")
        op(A1B1, py, qy, "mul")
        op(C0, res(A0B0), res(A1B1), "sub")
        op(A0B1, px, qy, "mul")
        op(A1B0, py, qx, "mul")
        op(C1, res(A0B1), res(A1B0), "add")
        outs = [(C0, 0), (C1, h)]
        syscall = b.const(codes["fp2_mul"] & 0xffff)
    for k, first_word in outs:       # the result is what is written (assert_all_eq of result and value_as_limbs)
        for i in range(N):
            b.when(is_real).assert_eq(res(k)[i], x_access[first_word + i // 4][4 + i % 4])
    for i in range(words):
        r.eval_memory_access(l[SHARD], l[CLK], l[Y_PTR] + 4 * i, y_access[i][0:4], y_access[i], is_real)
    for i in range(words):
        r.eval_memory_access(l[SHARD], l[CLK] + 1, l[X_PTR] + 4 * i, x_access[i][0:4], x_access[i][4:13], is_real)
    r.receives.append(air.Lookup([air.to_virtual_pair(v) for v in [l[SHARD], l[CLK], syscall, l[X_PTR], l[Y_PTR]]], air.to_virtual_pair(is_real), air.KIND_SYSCALL))


def record_fp_tower_constraints(field: str, kind: str) -> _Rec:
    r = _Rec(E.fp_tower_width(field, kind))
    _fp_tower(r, field, kind)
    return r


def record_fp_tower_chip(field: str, kind: str, log_height: int) -> RecordedChip:
    """<Field>FpOpAssign, <Field>Fp2AddSubAssign, <Field>Fp2MulAssign for Bn254 and Bls12381 (crates/core/machine/src/syscall/precompiles/fptower/);
    local_only as the reference declares them."""
    name = {"Bn254": "Bn254", "Bls12381": "Bls12831" if kind != "fp" else "Bls12381"}[field] + {"fp": "FpOpAssign", "fp2_addsub": "Fp2AddSubAssign", "fp2_mul": "Fp2MulAssign"}[kind]
    return _finish(record_fp_tower_constraints(field, kind), name, log_height, E.fp_tower_width(field, kind), True)


def record_ed_add_constraints() -> _Rec:
    r = _Rec(E.ED_ADD_WIDTH)
    _ed_add(r)
    return r


def record_ed_add_chip(log_height: int) -> RecordedChip:
    """The EdAddAssign precompile (crates/core/machine/src/syscall/precompiles/edwards/ed_add.rs): one Ed25519 point addition per row, 1861
    columns; local_only as the reference declares it (:206-208)."""
    return _finish(record_ed_add_constraints(), "EdAddAssign", log_height, E.ED_ADD_WIDTH, True)


def record_memory_global_constraints(finalize: bool) -> _Rec:
    r = _Rec(E.MEMORY_GLOBAL_WIDTH)
    _memory_global(r, finalize)
    return r


def record_memory_global_chip(finalize: bool, log_height: int) -> RecordedChip:
    """MemoryGlobalInit / MemoryGlobalFinalize (crates/core/machine/src/memory/global.rs): MemoryInitializeFinalizeEvents sorted by address,
    111 columns, constraints between consecutive rows and against the public values."""
    return _finish(record_memory_global_constraints(finalize), "MemoryGlobalFinalize" if finalize else "MemoryGlobalInit", log_height,
                   E.MEMORY_GLOBAL_WIDTH, False)


def record_syscall_table_constraints(precompile: bool) -> _Rec:
    r = _Rec(E.SYSCALL_WIDTH)
    _syscall_table(r, precompile)
    return r


def record_syscall_table_chip(precompile: bool, log_height: int) -> RecordedChip:
    """SyscallCore / SyscallPrecompile (crates/core/machine/src/syscall/chip.rs): SyscallEvents, 11 columns, local_only is not claimed by the
    reference (MachineAir::local_only defaults to false)."""
    return _finish(record_syscall_table_constraints(precompile), "SyscallPrecompile" if precompile else "SyscallCore", log_height, E.SYSCALL_WIDTH, False)


def record_keccak_sponge_constraints() -> _Rec:
    r = _Rec(E.KECCAK_SPONGE_WIDTH)
    _keccak_sponge(r)
    return r


def record_keccak_sponge_chip(log_height: int) -> RecordedChip:
    """The KeccakSponge precompile (crates/core/machine/src/syscall/precompiles/keccak_sponge/): 24 rows per 36-word block of a call, 3531
    columns (2633 of them the round columns of p3-keccak-air), constraints between consecutive rows; receives the syscall the
    SyscallPrecompile table sends."""
    return _finish(record_keccak_sponge_constraints(), "KeccakSponge", log_height, E.KECCAK_SPONGE_WIDTH, False)


def record_poseidon2_permute_constraints() -> _Rec:
    r = _Rec(E.POSEIDON2_PERMUTE_WIDTH)
    _poseidon2_permute(r)
    return r


def record_poseidon2_permute_chip(log_height: int) -> RecordedChip:
    """The Poseidon2Permute precompile (crates/core/machine/src/syscall/precompiles/poseidon2/): one permutation of sixteen memory words per
    row, 973 columns; receives the syscall the SyscallPrecompile table sends."""
    return _finish(record_poseidon2_permute_constraints(), "Poseidon2Permute", log_height, E.POSEIDON2_PERMUTE_WIDTH, False)
