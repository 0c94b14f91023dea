"""A small executor for the instruction classes whose chips are built: enough of crates/core/executor/src/executor.rs to
produce one *coherent* shard record — a program, its CpuEvents with register access records, and the per-chip event
vectors (ALU, Mul, DivRem, Branch, Jump, MovCond) exactly as `emit_events` (executor.rs:1106-1150) files them.

The reference's executor (Rust) cannot run here; the synthetic per-chip streams of events.py are independent of each
other, which is fine for a chip on its own but cannot feed the Cpu chip, whose rows are chained (next.pc = local.next_pc,
clk += 5) and whose instruction lookups must match the receiving chips' rows value for value. This module generates a
random forward-only program *while executing it* (MIPS delay slots included), so every pc is visited at most once and
the record is what the reference's executor would emit for that program:

  execute_operation   executor.rs:1463-1700      alu_rr / alu_rw      :1401-1444
  execute_alu         :1853-1924                  branch_rr / execute_branch  :1446-1458, :2092-2115
  execute_jump*       :2119-2148                  execute_condmov      :1830-1851
  rr_cpu / rw_cpu     :1041-1100 (timestamps clk + MemoryAccessPosition, events/memory.rs:29-40)

Register file: 32 general registers, LO = 32, HI = 33; loads and stores go to a small data region (execute_load / execute_store,
:1925-2088; memory at position Memory = clk + 0). Not modelled (their chips are not built): syscalls other than COMMIT and HALT.
"""
import numpy as np

from . import events as E
from . import field as F

# #[repr(C)] InstructionFfi (crates/core/executor/src/instruction.rs:22-33): opcode u8, op_a u8, op_b, op_c, imm_b, imm_c, raw: OptionU32
INSTRUCTION = np.dtype([("opcode", "u1"), ("op_a", "u1"), ("_p0", "u1", (2,)), ("op_b", "<u4"), ("op_c", "<u4"), ("imm_b", "u1"),
                        ("imm_c", "u1"), ("_p1", "u1", (2,)), ("raw_tag", "u1"), ("_p2", "u1", (3,)), ("raw", "<u4")])
assert INSTRUCTION.itemsize == 24
MEMORY_READ_RECORD = np.dtype([("value", "<u4"), ("shard", "<u4"), ("timestamp", "<u4"), ("prev_shard", "<u4"), ("prev_timestamp", "<u4")])
# #[repr(C)] OptionMemoryRecordEnum (events/cpu.rs:100-106): tag (Read = 0, Write = 1, None = 2), both records side by side
OPTION_MEMORY_RECORD = np.dtype([("tag", "u1"), ("_p", "u1", (3,)), ("read", MEMORY_READ_RECORD), ("write", E.MEMORY_WRITE_RECORD)])
assert OPTION_MEMORY_RECORD.itemsize == 48
OPTION_U32 = np.dtype([("tag", "u1"), ("_p", "u1", (3,)), ("value", "<u4")])     # OptionValTag: Some = 0, None = 1 (lib.rs:40-52)
# #[repr(C)] CpuEventFfi (events/cpu.rs:46-77)
CPU_EVENT = np.dtype([("clk", "<u4"), ("pc", "<u4"), ("next_pc", "<u4"), ("next_next_pc", "<u4"), ("a", "<u4"),
                      ("a_record", OPTION_MEMORY_RECORD), ("b", "<u4"), ("b_record", OPTION_MEMORY_RECORD), ("c", "<u4"),
                      ("c_record", OPTION_MEMORY_RECORD), ("hi", OPTION_U32), ("hi_record", OPTION_MEMORY_RECORD),
                      ("memory_record", OPTION_MEMORY_RECORD), ("exit_code", "<u4")])
assert CPU_EVENT.itemsize == 280
# #[repr(C)] MemoryLocalEvent (events/memory.rs:226-237): addr, initial MemoryRecord {shard, timestamp, value}, final MemoryRecord
MEMORY_RECORD = np.dtype([("shard", "<u4"), ("timestamp", "<u4"), ("value", "<u4")])
MEMORY_LOCAL_EVENT = np.dtype([("addr", "<u4"), ("initial", MEMORY_RECORD), ("final", MEMORY_RECORD)])
assert MEMORY_LOCAL_EVENT.itemsize == 28
MEMORY_LOCAL_ENTRIES_PER_ROW, MEMORY_LOCAL_WIDTH = 4, 56
# GlobalLookupEvent (crates/core/executor/src/events/global.rs:6-15, #[repr(C)]): message, is_receive, kind (LookupKind as u8)
GLOBAL_LOOKUP_EVENT = np.dtype([("message", "<u4", (7,)), ("is_receive", "u1"), ("kind", "u1"), ("pad", "u1", (2,))])
assert GLOBAL_LOOKUP_EVENT.itemsize == 32
GLOBAL_WIDTH = 99
TAG_READ, TAG_WRITE, TAG_NONE = 0, 1, 2
CPU_WIDTH = 67
PROGRAM_PREP_WIDTH, PROGRAM_MULT_WIDTH = 14, 1
POS_C, POS_B, POS_A, POS_HI = 1, 2, 3, 4      # MemoryAccessPosition
REG_LO, REG_HI = 32, 33
# public values (crates/stark/src/air/public_values.rs:22-60): committed_value_digest 8 words, deferred_proofs_digest 8, then
PV_START_PC, PV_NEXT_PC, PV_EXIT_CODE, PV_SHARD, PV_EXECUTION_SHARD = 40, 41, 42, 43, 44

_ALU_RR = [E.ADD, E.SUB, E.SLL, E.SRL, E.SRA, E.ROR, E.SLT, E.SLTU, E.AND, E.OR, E.XOR, E.NOR, E.MUL]
_SHIFT = (E.SLL, E.SRL, E.SRA, E.ROR)
_BRANCH = [E.BEQ, E.BNE, E.BGEZ, E.BLEZ, E.BGTZ, E.BLTZ]
_ONE_OPERAND = (E.BGEZ, E.BLEZ, E.BGTZ, E.BLTZ)


class Record:
    """What ExecutionRecord holds for the chips that are built."""

    def __init__(self):
        self.cpu, self.alu, self.mul, self.divrem, self.branch, self.jump, self.mov_cond = [], {c: [] for c in E.CHIP_NAMES}, [], [], [], [], []
        self.memory_local, self.mem_instr, self.syscall, self.misc = [], [], [], []


def _alu(op, b, c):
    return int(E.alu_result(np.array([op], dtype=np.uint8), np.array([b], dtype=np.uint64), np.array([c], dtype=np.uint64))[0])


_CHIP_OF = {op: chip for chip, ops in E.CHIP_OPCODES.items() for op in ops}


def run(n_cycles: int, seed: int = 1, shard: int = 1, pc_base: int = 0x1000, halt: bool = False):
    """Execute `n_cycles` instructions of a program generated on the way. Returns (program, record, public_values) with
    program: INSTRUCTION array (instructions that were jumped over are `ADD $0, 0, 0` no-ops that never run), record:
    structured event arrays, public_values: dict of the words the Cpu chip checks (start_pc, next_pc, execution_shard).
    With `halt`, the program ends as a real one does: eight COMMIT syscalls publish the words of a value digest, then HALT
    (execute_operation's SYSCALL arm, executor.rs:1591-1668): 35 more cycles, next_pc = 0, and the public values carry the
    committed digest and exit code 0."""
    rng = np.random.default_rng(seed)
    R = [int(x) for x in rng.integers(0, 1 << 32, 34, dtype=np.uint64)]
    R[0] = 0
    for i in range(1, 34, 5):     # small and special values so comparisons, shifts and divisions hit their corners
        R[i] = int(E._CORNERS[i % len(E._CORNERS)])
    last = {i: (0, 0) for i in range(34)}   # (shard, timestamp) of the previous access to a register / memory word; shard 0 = before this shard
    R = dict(enumerate(R))         # registers 0..33 and, keyed by their byte address, the memory words that get touched
    DATA, LOW = 0x00100000, 64     # a data region, and a few words whose address fits one byte (the chip's `addr < 256` branch)
    first = {}                     # register -> (shard, timestamp, value) on entry to this shard (ExecutionRecord::cpu_local_memory_access)
    program = {}
    rec = Record()
    pc, next_pc = pc_base, pc_base + 4
    delay_slot = False
    pending_jump_reg = None        # a register just loaded with a jump target

    def read(reg, clk, pos):
        if reg not in R:
            R[reg], last[reg] = 0, (0, 0)
        first.setdefault(reg, (last[reg][0], last[reg][1], R[reg]))
        r = (TAG_READ, (R[reg], shard, clk + pos, last[reg][0], last[reg][1]), None)
        last[reg] = (shard, clk + pos)
        return r

    def write(reg, value, clk, pos):
        value = 0 if reg == 0 else value & 0xffffffff
        if reg not in R:
            R[reg], last[reg] = 0, (0, 0)
        first.setdefault(reg, (last[reg][0], last[reg][1], R[reg]))
        r = (TAG_WRITE, None, (value, shard, clk + pos, R[reg], last[reg][0], last[reg][1]))
        R[reg] = value
        last[reg] = (shard, clk + pos)
        return r

    digest = [int(x) for x in rng.integers(0, 1 << 32, 8, dtype=np.uint64)]
    epilogue = []
    if halt:   # li $v0, code; li $a0, arg1; li $a1, arg2; syscall — eight commits, then halt with exit code 0
        for i, w in enumerate(digest):
            epilogue += [(E.ADD, E.REG_V0, E.SYS_COMMIT, 0, 1, 1), (E.ADD, E.REG_A0, i, 0, 1, 1), (E.ADD, E.REG_A1, w, 0, 1, 1),
                         (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
        epilogue += [(E.ADD, E.REG_V0, E.SYS_HALT, 0, 1, 1), (E.ADD, E.REG_A0, 0, 0, 1, 1), (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
    cyc = -1
    while cyc + 1 < n_cycles + len(epilogue):
        cyc += 1
        clk = 5 * cyc
        u = rng.random()
        if cyc >= n_cycles and delay_slot:      # a branch's delay slot comes first: one more plain instruction
            epilogue.insert(cyc - n_cycles, (E.ADD, 1, 0, 0, 1, 1))
        reg = lambda: int(rng.integers(0, 32))       # noqa: E731
        dst = lambda: int(rng.integers(1, 32)) if rng.random() > 0.02 else 0   # noqa: E731  (a few writes to $0)
        # ---- pick the instruction at pc (the program is written as it runs)
        if cyc >= n_cycles:
            ins = epilogue[cyc - n_cycles]
        elif pending_jump_reg is not None and not delay_slot:
            ins = (E.JUMP, dst(), pending_jump_reg, 0, 0, 1)
            pending_jump_reg = None
        elif delay_slot or u < 0.36:
            op = _ALU_RR[int(rng.integers(0, len(_ALU_RR)))]
            form = rng.random()
            if form < 0.55:
                ins = (op, dst(), reg(), reg(), 0, 0)
            elif form < 0.95:
                imm = int(rng.integers(0, 32)) if op in _SHIFT else int(rng.integers(0, 1 << 32)) if rng.random() < 0.5 else int(rng.integers(0, 1 << 16))
                ins = (op, dst(), reg(), imm, 0, 1)
            else:
                ins = (op, dst(), int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 16)), 1, 1)
        elif u < 0.50:
            op = int(rng.integers(E.LB, E.SC + 1))
            rs = reg()
            word_addr = (LOW + 4 * int(rng.integers(0, 40))) if rng.random() < 0.1 else DATA + 4 * int(rng.integers(0, 96))
            off = 0 if op in (E.LW, E.LL, E.SW, E.SC) else 2 * int(rng.integers(0, 2)) if op in (E.LH, E.LHU, E.SH) else int(rng.integers(0, 4))
            ins = (op, dst(), rs, (word_addr + off - R[rs]) & 0xffffffff, 0, 1)   # the offset that lands on the chosen address
        elif u < 0.54:
            ins = (E.CLZ if rng.random() < 0.5 else E.CLO, dst(), reg(), 0, 0, 1)
        elif u < 0.58:    # the MiscInstrs chip's opcodes
            k = int(rng.integers(0, 8))
            if k == 0:
                ins = (E.SEXT, dst(), reg(), int(rng.integers(0, 2)), 0, 1) if rng.random() < 0.6 else (E.WSBH, dst(), reg(), 0, 0, 1)
            elif k == 1:
                lsb = int(rng.integers(0, 32))
                ins = (E.EXT, dst(), reg(), lsb | int(rng.integers(0, 32 - lsb)) << 5, 0, 1)
            elif k == 2:
                lsb = int(rng.integers(0, 32))
                ins = (E.INS, dst(), reg(), lsb | int(rng.integers(lsb, 32)) << 5, 0, 1)
            elif k == 3:
                rs, rt = reg(), reg()
                ins = (E.TEQ, rs, rt, 0, 0, 1) if R[rs] != R[rt] else (E.SEXT, dst(), rs, 1, 0, 1)   # equal operands trap
            else:
                ins = ([E.MADDU, E.MSUBU, E.MADD, E.MSUB][k - 4], REG_LO, reg(), reg(), 0, 0)
        elif u < 0.66:
            ins = (E.MULT if rng.random() < 0.5 else E.MULTU, REG_LO, reg(), reg(), 0, 0)
        elif u < 0.76:
            op = [E.DIV, E.DIVU, E.MOD, E.MODU][int(rng.integers(0, 4))]
            rs, rt = reg(), reg()
            a_reg = REG_LO if op in (E.DIV, E.DIVU) else dst()
            if R[rt] == 0:          # the executor traps on a zero divisor (executor.rs:1858-1862): divide by an immediate instead
                ins = (op, a_reg, rs, int(rng.integers(1, 1 << 32)) if rng.random() < 0.5 else int(rng.integers(1, 256)), 0, 1)
            else:
                ins = (op, a_reg, rs, rt, 0, 0)
        elif u < 0.82:
            ins = (E.MEQ if rng.random() < 0.5 else E.MNE, dst(), reg(), reg(), 0, 0)
        elif u < 0.92:
            op = _BRANCH[int(rng.integers(0, 6))]
            off = 4 * int(rng.integers(2, 12))
            ins = (op, reg(), 0, off, 1, 1) if op in _ONE_OPERAND else (op, reg(), reg(), off, 0, 1)
        elif u < 0.95:
            ins = (E.JUMPI, 31 if rng.random() < 0.5 else 0, next_pc + 4 * int(rng.integers(2, 10)), 0, 1, 1)
        elif u < 0.98:
            ins = (E.JUMPDIRECT, 31, 4 * int(rng.integers(2, 10)), 0, 1, 1)
        else:   # load a forward target into a register (ADD of two immediates); the jump through it comes next
            pending_jump_reg = int(rng.integers(1, 32))
            ins = (E.ADD, pending_jump_reg, next_pc + 4 + 4 * int(rng.integers(3, 10)), 0, 1, 1)
        assert pc not in program
        program[pc] = ins
        op, op_a, op_b, op_c, imm_b, imm_c = ins
        # ---- execute it
        next_next_pc = next_pc + 4
        a_rec = b_rec = c_rec = hi_rec = m_rec = None
        hi = syscall_next = None
        was_delay_slot, delay_slot = delay_slot, False
        if op in _CHIP_OF or op in (E.MUL, E.MULT, E.MULTU, E.DIV, E.DIVU, E.MOD, E.MODU):
            if not imm_c:
                c_rec = read(op_c, clk, POS_C)
                c = c_rec[1][0]
                b_rec = read(op_b, clk, POS_B)
                b = b_rec[1][0]
            elif not imm_b:
                b_rec = read(op_b, clk, POS_B)
                b, c = b_rec[1][0], op_c
            else:
                b, c = op_b, op_c
            if op in (E.MUL, E.MULT, E.MULTU):
                lo, h = E.mul_result(np.array([op]), np.array([b], dtype=np.uint32), np.array([c], dtype=np.uint32))
                a, hv = int(lo[0]), int(h[0])
            elif op in (E.DIV, E.DIVU, E.MOD, E.MODU):
                q, r = E.quotient_and_remainder(np.array([op]), np.array([b], dtype=np.uint32), np.array([c], dtype=np.uint32))
                a, hv = (int(q[0]), int(r[0])) if op in (E.DIV, E.DIVU) else (int(r[0]), 0)
            else:
                a, hv = _alu(op, b, c), 0
            if op in (E.MULT, E.MULTU, E.DIV, E.DIVU):   # is_use_lo_hi_alu: LO at position A, HI at position HI
                a_rec = write(REG_LO, a, clk, POS_A)
                hi_rec = write(REG_HI, hv, clk, POS_HI)
                hi = hv
            else:
                a_rec = write(op_a, a, clk, POS_A)
            if op in (E.MUL, E.MULT, E.MULTU, E.DIV, E.DIVU, E.MOD, E.MODU):
                w = hi_rec[2] if hi_rec else (0, 0, 0, 0, 0, 0)
                ev = (shard, clk, pc, next_pc, op, [0, 0, 0], hi or 0, a, b, c, w, 1 if hi_rec else 0, [0, 0, 0])
                (rec.mul if op in (E.MUL, E.MULT, E.MULTU) else rec.divrem).append(ev)
            else:
                rec.alu[_CHIP_OF[op]].append((pc, next_pc, op, [0, 0, 0], 0, a, b, c))
        elif op in E.LOADS:
            b_rec = read(op_b, clk, POS_B)
            b, c = b_rec[1][0], op_c
            rt = R[op_a]                                   # peeked, no record: the write record's prev_value carries it
            addr = (b + c) & 0xffffffff
            m_rec = read(addr & 0xfffffffc, clk, 0)       # MemoryAccessPosition::Memory
            a = E.load_value(op, m_rec[1][0], addr, rt)
            a_rec = write(op_a, a, clk, POS_A)
            hi = rt
            rec.mem_instr.append((shard, clk, pc, next_pc, op, [0, 0, 0], a, b, c, TAG_READ, list(m_rec[1]) + [0], rt))
        elif op in E.STORES:
            b_rec = read(op_b, clk, POS_B)
            b, c = b_rec[1][0], op_c
            if op == E.SC:
                rt = R[op_a]
            else:
                a_rec = read(op_a, clk, POS_A)
                rt = a_rec[1][0]
            addr = (b + c) & 0xffffffff
            aligned = addr & 0xfffffffc
            m_rec = write(aligned, E.store_value(op, R.get(aligned, 0), addr, rt), clk, 0)
            if op == E.SC:
                a_rec = write(op_a, 1, clk, POS_A)
                a = 1
            else:
                a = rt
            hi = rt
            rec.mem_instr.append((shard, clk, pc, next_pc, op, [0, 0, 0], a, b, c, TAG_WRITE, list(m_rec[2]), rt))
        elif op in (E.SEXT, E.EXT, E.INS):
            b_rec = read(op_b, clk, POS_B)
            b, c = b_rec[1][0], op_c
            prev_a = R[op_a] if op == E.INS else 0
            a = E.misc_result(op, b, c, prev_a)
            a_rec = write(op_a, a, clk, POS_A)
            hi = prev_a if op == E.INS else None
            rec.misc.append((shard, clk, pc, next_pc, op, [0, 0, 0], a, b, c, prev_a, (0, 0, 0, 0, 0, 0)))
        elif op == E.WSBH:      # execute_wsbh (executor.rs:1776-1785): swap the bytes within each halfword; a MovCond chip event
            b_rec = read(op_b, clk, POS_B)
            b, c = b_rec[1][0], 0
            a = ((b >> 16) & 0xff) << 24 | ((b >> 24) & 0xff) << 16 | (b & 0xff) << 8 | (b >> 8) & 0xff
            a_rec = write(op_a, a, clk, POS_A)
            rec.mov_cond.append((pc, next_pc, op, [0, 0, 0], a, b, c, 0))
        elif op == E.TEQ:
            b_rec = read(op_b, clk, POS_B)
            a_rec = read(op_a, clk, POS_A)
            a, b, c = a_rec[1][0], b_rec[1][0], 0
            assert a != b
            rec.misc.append((shard, clk, pc, next_pc, op, [0, 0, 0], a, b, c, 0, (0, 0, 0, 0, 0, 0)))
        elif op in (E.MADDU, E.MSUBU, E.MADD, E.MSUB):
            c_rec = read(op_c, clk, POS_C)
            c = c_rec[1][0]
            b_rec = read(op_b, clk, POS_B)
            b = b_rec[1][0]
            lo_val = R[REG_LO]
            a, hv = E.misc_result(op, b, c, hi_lo=(R[REG_HI], lo_val))
            a_rec = write(REG_LO, a, clk, POS_A)
            hi_rec = write(REG_HI, hv, clk, POS_HI)
            hi = lo_val
            rec.misc.append((shard, clk, pc, next_pc, op, [0, 0, 0], a, b, c, lo_val, hi_rec[2]))
        elif op == E.SYSCALL:
            code = R[op_a]                                 # peeked: the write record's prev_value carries it
            c_rec = read(op_c, clk, POS_C)
            c = c_rec[1][0]
            b_rec = read(op_b, clk, POS_B)
            b = b_rec[1][0]
            sid = code & 0xffff
            assert sid in (E.SYS_HALT, E.SYS_COMMIT), sid
            a = code                                       # neither returns a value: V0 keeps the code
            a_rec = write(op_a, a, clk, POS_A)
            hi = code
            next_pc_after = 0 if sid == E.SYS_HALT else next_pc
            rec.syscall.append((pc, next_pc_after, shard, clk, a_rec[2], 1, [0, 0, 0], sid, b, c))
            syscall_next = next_pc_after
        elif op in _BRANCH:
            if op in _ONE_OPERAND:
                b = 0
            else:
                b_rec = read(op_b, clk, POS_B)
                b = b_rec[1][0]
            a_rec = read(op_a, clk, POS_A)
            a, c = a_rec[1][0], op_c
            sa = a - (1 << 32) if a >> 31 else a
            taken = {E.BEQ: a == b, E.BNE: a != b, E.BGEZ: sa >= 0, E.BLEZ: sa <= 0, E.BGTZ: sa > 0, E.BLTZ: sa < 0}[op]
            if taken:
                next_next_pc = (c + next_pc) & 0xffffffff
            delay_slot = True
            rec.branch.append((pc, next_pc, next_next_pc, op, [0, 0, 0], a, b, c))
        elif op in (E.JUMP, E.JUMPI, E.JUMPDIRECT):
            if op == E.JUMP:
                b_rec = read(op_b, clk, POS_B)
                target = b = b_rec[1][0]
            elif op == E.JUMPI:
                target = b = op_b
            else:
                b, target = op_b, (op_b + next_pc) & 0xffffffff
            a = (next_pc + 4) & 0xffffffff
            a_rec = write(op_a, a, clk, POS_A)
            c, next_next_pc = 0, target
            delay_slot = True
            rec.jump.append((pc, next_pc, next_next_pc, op, [0, 0, 0], a, b, c))
        elif op in (E.MEQ, E.MNE):
            prev_a = R[op_a]
            c_rec = read(op_c, clk, POS_C)
            c = c_rec[1][0]
            b_rec = read(op_b, clk, POS_B)
            b = b_rec[1][0]
            a = b if ((c == 0) == (op == E.MEQ)) else prev_a
            a_rec = write(op_a, a, clk, POS_A)
            hi = prev_a
            rec.mov_cond.append((pc, next_pc, op, [0, 0, 0], a, b, c, prev_a))
        else:
            raise AssertionError(op)
        del was_delay_slot
        # op_a_value of the Cpu row is `a` as computed (a write to $0 stores 0 but the event keeps the result)
        if syscall_next is not None:       # next_pc = precompile_next_pc, next_next_pc = precompile_next_pc + 4 (executor.rs:1660-1661)
            next_pc, next_next_pc = syscall_next, syscall_next + 4
        rec.cpu.append((clk, pc, next_pc, next_next_pc, a, a_rec, b, b_rec, c, c_rec, hi, hi_rec, m_rec))
        pc, next_pc = next_pc, next_next_pc

    # ---- pack
    top = max(program) if program else pc_base
    prog = np.zeros((top - pc_base) // 4 + 1, dtype=INSTRUCTION)
    prog["opcode"], prog["imm_b"], prog["imm_c"], prog["raw_tag"] = E.ADD, 1, 1, 1     # the filler: ADD $0, 0, 0
    for p, (op, op_a, op_b, op_c, imm_b, imm_c) in program.items():
        prog[(p - pc_base) // 4] = (op, op_a, [0, 0], op_b, op_c, imm_b, imm_c, [0, 0], 1, [0, 0, 0], 0)

    def opt(r):
        o = np.zeros((), dtype=OPTION_MEMORY_RECORD)
        o["tag"] = TAG_NONE
        if r is not None:
            o["tag"] = r[0]
            if r[0] == TAG_READ:
                o["read"] = r[1]
            else:
                o["write"] = r[2]
        return o

    cpu = np.zeros(len(rec.cpu), dtype=CPU_EVENT)
    none = opt(None)
    for i, (clk, p, np_, nnp, a, a_rec, b, b_rec, c, c_rec, hi, hi_rec, m_rec) in enumerate(rec.cpu):
        e = cpu[i]
        e["clk"], e["pc"], e["next_pc"], e["next_next_pc"], e["a"], e["b"], e["c"] = clk, p, np_, nnp, a, b, c
        e["a_record"], e["b_record"], e["c_record"], e["hi_record"] = opt(a_rec), opt(b_rec), opt(c_rec), opt(hi_rec)
        e["memory_record"] = opt(m_rec) if m_rec is not None else none
        e["hi"]["tag"], e["hi"]["value"] = (0, hi) if hi is not None else (1, 0)
    out = Record()
    out.cpu = cpu
    out.alu = {chip: np.array(v, dtype=E.ALU_EVENT) if v else np.zeros(0, dtype=E.ALU_EVENT) for chip, v in rec.alu.items()}
    out.mul = np.array(rec.mul, dtype=E.COMP_ALU_EVENT) if rec.mul else np.zeros(0, dtype=E.COMP_ALU_EVENT)
    out.divrem = np.array(rec.divrem, dtype=E.COMP_ALU_EVENT) if rec.divrem else np.zeros(0, dtype=E.COMP_ALU_EVENT)
    out.branch = np.array(rec.branch, dtype=E.BRANCH_EVENT) if rec.branch else np.zeros(0, dtype=E.BRANCH_EVENT)
    out.jump = np.array(rec.jump, dtype=E.JUMP_EVENT) if rec.jump else np.zeros(0, dtype=E.JUMP_EVENT)
    out.mov_cond = np.array(rec.mov_cond, dtype=E.MOV_COND_EVENT) if rec.mov_cond else np.zeros(0, dtype=E.MOV_COND_EVENT)
    out.mem_instr = np.array(rec.mem_instr, dtype=E.MEM_INSTR_EVENT) if rec.mem_instr else np.zeros(0, dtype=E.MEM_INSTR_EVENT)
    out.syscall = np.array(rec.syscall, dtype=E.SYSCALL_EVENT) if rec.syscall else np.zeros(0, dtype=E.SYSCALL_EVENT)
    out.misc = np.array(rec.misc, dtype=E.MISC_EVENT) if rec.misc else np.zeros(0, dtype=E.MISC_EVENT)
    out.memory_local = np.array([(reg, first[reg], (last[reg][0], last[reg][1], R[reg])) for reg in sorted(first)], dtype=MEMORY_LOCAL_EVENT) \
        if first else np.zeros(0, dtype=MEMORY_LOCAL_EVENT)
    pv = {"start_pc": pc_base, "next_pc": int(cpu["next_pc"][-1]) if len(cpu) else pc_base, "execution_shard": shard, "shard": shard,
          "exit_code": 0, "committed_value_digest": digest if halt else [0] * 8}
    return prog, out, pv


def add_dependencies(rec: Record) -> Record:
    """The events the executor derives while it runs (crates/core/executor/src/dependencies.rs): CLO/CLZ -> SRL, JumpDirect ->
    ADD, branches -> SLT x2 (+ ADD when taken), divisions -> ADD / MULT(U) / SLTU. Appended to the receiving chips' vectors."""
    alu = dict(rec.alu)
    lt_dep, add_dep = E.branch_dependencies(rec.branch)
    div_add, div_mul, div_lt = E.divrem_dependencies(rec.divrem)
    misc_mul, misc_sll, misc_sr, misc_add = E.misc_dependencies(rec.misc)
    alu[E.CHIP_SHIFT_RIGHT] = np.concatenate([alu[E.CHIP_SHIFT_RIGHT], E.cloclz_dependencies(alu[E.CHIP_CLO_CLZ]), misc_sr])
    alu[E.CHIP_SHIFT_LEFT] = np.concatenate([alu[E.CHIP_SHIFT_LEFT], misc_sll])
    alu[E.CHIP_ADD_SUB] = np.concatenate([alu[E.CHIP_ADD_SUB], E.jump_dependencies(rec.jump), add_dep, div_add, E.memory_dependencies(rec.mem_instr),
                                          misc_add])
    alu[E.CHIP_LT] = np.concatenate([alu[E.CHIP_LT], lt_dep, div_lt])
    out = Record()
    out.cpu, out.alu, out.branch, out.jump, out.mov_cond, out.divrem = rec.cpu, alu, rec.branch, rec.jump, rec.mov_cond, rec.divrem
    out.memory_local, out.mem_instr, out.syscall, out.misc = rec.memory_local, rec.mem_instr, rec.syscall, rec.misc
    out.mul = np.concatenate([rec.mul, div_mul, misc_mul])
    return out


def global_lookup_events(memory_local: np.ndarray) -> np.ndarray:
    """MemoryLocalChip::generate_dependencies (crates/core/machine/src/memory/local.rs:98-135): per touched address, the access the shard
    starts from is *received* from the global table and the one it ends with is *sent* to it; message = (shard, timestamp, addr,
    the value's four bytes), kind Memory."""
    out = np.zeros(2 * len(memory_local), dtype=GLOBAL_LOOKUP_EVENT)
    for half, rec, is_receive in ((0, "initial", 1), (1, "final", 0)):
        m = out["message"][half::2]
        m[:, 0], m[:, 1], m[:, 2] = memory_local[rec]["shard"], memory_local[rec]["timestamp"], memory_local["addr"]
        for k in range(4):
            m[:, 3 + k] = (memory_local[rec]["value"] >> (8 * k)) & 0xff
        out["is_receive"][half::2] = is_receive
    out["kind"] = 1     # LookupKind::Memory
    return out


def public_values(pv: dict) -> np.ndarray:
    """The public-values vector (Montgomery words) with the fields the Cpu chip constrains filled in."""
    from . import synth
    v = np.zeros(synth.PROOF_MAX_NUM_PVS, dtype=np.uint64)
    v[PV_START_PC], v[PV_NEXT_PC], v[PV_SHARD], v[PV_EXECUTION_SHARD] = pv["start_pc"], pv["next_pc"], pv["shard"], pv["execution_shard"]
    v[PV_EXIT_CODE] = pv.get("exit_code", 0)
    for i, w in enumerate(pv.get("committed_value_digest", [0] * 8)):     # [Word; 8]: one field element per byte
        v[4 * i:4 * i + 4] = [(w >> (8 * k)) & 0xff for k in range(4)]
    return F.to_monty(v)
