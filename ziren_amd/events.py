"""Executor events of the ALU chips, as the reference lays them out in memory, and synthetic event streams.

`ALU_EVENT` is the `#[repr(C)] AluEvent` of crates/core/executor/src/events/instr.rs:10-26 (28 bytes); opcode
numbers are crates/core/executor/src/opcode.rs:26-48. The executor cannot run here, so shards are
filled with synthetic instruction streams: operands from SplitMix64 with the corner cases mixed in, results
computed with the MIPS semantics of the executor's ALU (`a = b op c`).
"""
import numpy as np

from . import field as F

ALU_EVENT = np.dtype([("pc", "<u4"), ("next_pc", "<u4"), ("opcode", "u1"), ("_pad", "u1", (3,)), ("hi", "<u4"),
                      ("a", "<u4"), ("b", "<u4"), ("c", "<u4")])
assert ALU_EVENT.itemsize == 28

ADD, SUB = 0, 1
SLL, SRL, SRA, ROR = 9, 10, 11, 12
SLT, SLTU = 13, 14
AND, OR, XOR, NOR = 15, 16, 17, 18
CLZ, CLO = 19, 20
JUMP, JUMPI, JUMPDIRECT = 27, 28, 29
MEQ, MNE, WSBH = 50, 51, 52
BEQ, BGEZ, BGTZ, BLEZ, BLTZ, BNE = 21, 22, 23, 24, 25, 26
UNUSED_PC, DEFAULT_PC_INC = 1, 4  # crates/core/executor/src/executor.rs:44-47

# #[repr(C)] JumpEvent, crates/core/executor/src/events/instr.rs:200-217 (28 bytes as well, different fields)
JUMP_EVENT = np.dtype([("pc", "<u4"), ("next_pc", "<u4"), ("next_next_pc", "<u4"), ("opcode", "u1"), ("_pad", "u1", (3,)),
                       ("a", "<u4"), ("b", "<u4"), ("c", "<u4")])
assert JUMP_EVENT.itemsize == 28
JUMP_WIDTH = 66
BRANCH_EVENT = JUMP_EVENT    # #[repr(C)] BranchEvent, events/instr.rs:161-178: the same seven fields
BRANCH_WIDTH = 62
# #[repr(C)] MovCondEvent, crates/core/executor/src/events/instr.rs:286-302
MOV_COND_EVENT = np.dtype([("pc", "<u4"), ("next_pc", "<u4"), ("opcode", "u1"), ("_pad", "u1", (3,)), ("a", "<u4"), ("b", "<u4"),
                           ("c", "<u4"), ("prev_a", "<u4")])
assert MOV_COND_EVENT.itemsize == 28
MOV_COND_WIDTH = 32

# zkm_alu_chip
CHIP_ADD_SUB, CHIP_BITWISE, CHIP_LT, CHIP_SHIFT_LEFT, CHIP_SHIFT_RIGHT, CHIP_CLO_CLZ = range(6)
CHIP_NAMES = {CHIP_ADD_SUB: "AddSub", CHIP_BITWISE: "Bitwise", CHIP_LT: "Lt", CHIP_SHIFT_LEFT: "ShiftLeft",
              CHIP_SHIFT_RIGHT: "ShiftRight", CHIP_CLO_CLZ: "CloClz"}
CHIP_WIDTH = {CHIP_ADD_SUB: 19, CHIP_BITWISE: 18, CHIP_LT: 32, CHIP_SHIFT_LEFT: 44, CHIP_SHIFT_RIGHT: 67, CHIP_CLO_CLZ: 17}
CHIP_OPCODES = {CHIP_ADD_SUB: (ADD, SUB), CHIP_BITWISE: (AND, OR, XOR, NOR), CHIP_LT: (SLT, SLTU),
                CHIP_SHIFT_LEFT: (SLL,), CHIP_SHIFT_RIGHT: (SRL, SRA, ROR), CHIP_CLO_CLZ: (CLZ, CLO)}

_CORNERS = np.array([0, 1, 2, 0x7f, 0x80, 0xff, 0x100, 0xffff, 0x10000, 0x7fffffff, 0x80000000, 0x80000001,
                     0xfffffffe, 0xffffffff, 0x00ff00ff, 0xff00ff00, 0x01010101], dtype=np.uint64)


def alu_result(opcode: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """`a` of `a = b op c` for arrays of opcodes and u32 operands."""
    b = b.astype(np.uint64)
    c = c.astype(np.uint64)
    sh = c & 31
    sb = b.astype(np.uint32).astype(np.int32).astype(np.int64)
    sc = c.astype(np.uint32).astype(np.int32).astype(np.int64)
    M = np.uint64(0xffffffff)
    out = np.zeros_like(b)
    sel = lambda op: opcode == op  # noqa: E731
    out = np.where(sel(ADD), (b + c) & M, out)
    out = np.where(sel(SUB), (b - c) & M, out)
    out = np.where(sel(AND), b & c, out)
    out = np.where(sel(OR), b | c, out)
    out = np.where(sel(XOR), b ^ c, out)
    out = np.where(sel(NOR), ~(b | c) & M, out)
    out = np.where(sel(SLT), (sb < sc).astype(np.uint64), out)
    out = np.where(sel(SLTU), (b < c).astype(np.uint64), out)
    out = np.where(sel(SLL), (b << sh) & M, out)
    out = np.where(sel(SRL), b >> sh, out)
    out = np.where(sel(SRA), (sb >> sh.astype(np.int64)).astype(np.uint64) & M, out)
    out = np.where(sel(ROR), ((b >> sh) | (b << (np.uint64(32) - sh))) & M, out)
    if np.any(sel(CLZ) | sel(CLO)):  # leading zeros of b / of !b (crates/core/executor/src/executor.rs:1916-1917)
        x = np.where(sel(CLO), ~b & M, b)
        lz = np.full(b.shape, 32, dtype=np.uint64)
        nz = x != 0
        lz[nz] = 31 - np.floor(np.log2(x[nz].astype(np.float64))).astype(np.uint64)
        out = np.where(sel(CLZ) | sel(CLO), lz, out)
    return out.astype(np.uint32)


def make_alu_events(opcode, b, c, pc0: int = 0x1000) -> np.ndarray:
    opcode = np.asarray(opcode, dtype=np.uint8)
    n = len(opcode)
    ev = np.zeros(n, dtype=ALU_EVENT)
    ev["pc"] = (pc0 + 4 * np.arange(n, dtype=np.uint64)) & 0x7ffffffc
    ev["next_pc"] = ev["pc"] + 4
    ev["opcode"] = opcode
    ev["b"] = np.asarray(b, dtype=np.uint32)
    ev["c"] = np.asarray(c, dtype=np.uint32)
    ev["a"] = alu_result(opcode, ev["b"], ev["c"])
    return ev


def synthetic_alu_events(chip: int, n: int, seed: int = 1) -> np.ndarray:
    """n events for `chip`: uniform operands, with about a quarter of them replaced by corner values, equal
    operands and (for the shift chips) every shift amount, so that each branch of the row builder is taken."""
    rng = F.SplitMix64(0x414c5500 + 97 * chip + seed)
    ops = np.array(CHIP_OPCODES[chip], dtype=np.uint8)
    raw = rng.next_u64(4 * n)
    r0, r1, r2, r3 = raw[:n], raw[n:2 * n], raw[2 * n:3 * n], raw[3 * n:]
    opcode = ops[(r0 % np.uint64(len(ops))).astype(np.int64)]
    b = (r1 & np.uint64(0xffffffff)).astype(np.uint64)
    c = (r2 & np.uint64(0xffffffff)).astype(np.uint64)
    kind = (r3 >> np.uint64(8)) % np.uint64(16)
    pick = lambda r: _CORNERS[(r % np.uint64(len(_CORNERS))).astype(np.int64)]  # noqa: E731
    b = np.where(kind == 0, pick(r3 >> np.uint64(16)), b)
    c = np.where(kind == 1, pick(r3 >> np.uint64(24)), c)
    c = np.where(kind == 2, b, c)                                   # equal operands
    c = np.where(kind == 3, b ^ (np.uint64(1) << ((r3 >> np.uint64(32)) % np.uint64(32))), c)  # one differing bit
    if chip in (CHIP_SHIFT_LEFT, CHIP_SHIFT_RIGHT):
        c = np.where(kind < 8, (r3 >> np.uint64(40)) % np.uint64(32), c)  # small shift amounts, all 32 of them
    if chip == CHIP_CLO_CLZ:  # every count 0..32: shift the operand down (CLZ) or fill it with ones from the top (CLO)
        k = (r3 >> np.uint64(40)) % np.uint64(33)
        low = np.where(k >= 32, np.uint64(0), (b & np.uint64(0xffffffff)) >> np.minimum(k, np.uint64(31)))
        b = np.where(opcode == CLZ, low, ~low & np.uint64(0xffffffff))
        c = np.zeros_like(c)
    return make_alu_events(opcode, b, c)


def cloclz_dependencies(cloclz_events: np.ndarray) -> np.ndarray:
    """The SRL events the executor adds to `shift_right_events` for CLO/CLZ instructions
    (emit_cloclz_dependencies, crates/core/executor/src/dependencies.rs:105-122): they carry the CloClz chip's
    `send_alu` to the ShiftRight chip."""
    ev = cloclz_events
    b = np.where(ev["opcode"] == CLZ, ev["b"], ~ev["b"]).astype(np.uint32)
    keep = b != 0
    b, a = b[keep], ev["a"][keep]
    out = np.zeros(len(b), dtype=ALU_EVENT)
    out["pc"] = UNUSED_PC
    out["next_pc"] = UNUSED_PC + DEFAULT_PC_INC
    out["opcode"] = SRL
    out["b"] = b
    out["c"] = 31 - a
    out["a"] = b >> (31 - a)
    return out


def synthetic_jump_events(n: int, seed: int = 1) -> np.ndarray:
    """n jump instructions with the executor's semantics: a = next_pc + 4 (the link value); Jump / Jumpi go to b;
    JumpDirect goes to next_pc + b (b may be negative). All program counters stay below p, as the chip range-checks."""
    raw = F.SplitMix64(0x4a4d5000 + seed).next_u64(3 * n)
    r0, r1, r2 = raw[:n], raw[n:2 * n], raw[2 * n:]
    ev = np.zeros(n, dtype=JUMP_EVENT)
    pc = ((r0 % np.uint64(0x3fffff00)) & np.uint64(0xfffffffc)).astype(np.uint32)
    ev["pc"] = pc
    ev["next_pc"] = pc + 4
    ev["opcode"] = np.array([JUMP, JUMPI, JUMPDIRECT], dtype=np.uint8)[(r1 % np.uint64(3)).astype(np.int64)]
    ev["a"] = ev["next_pc"] + 4
    target = ((r2 % np.uint64(0x7effff00)) & np.uint64(0xfffffffc)).astype(np.uint32)
    corner = (r2 >> np.uint64(40)) % np.uint64(16)
    target = np.where(corner == 0, np.uint32(0x7f000000), target)       # the largest value the range checker accepts
    target = np.where(corner == 1, np.uint32(0), target)
    direct = ev["opcode"] == JUMPDIRECT
    ev["b"] = np.where(direct, target - ev["next_pc"], target)           # offset (wrapping) or absolute target
    ev["next_next_pc"] = target
    ev["c"] = (r1 >> np.uint64(32)).astype(np.uint32) & np.uint32(0xffff)
    return ev


def jump_dependencies(jump_events: np.ndarray) -> np.ndarray:
    """The ADD events the executor adds to `add_sub_events` for JumpDirect (emit_jump_dependencies,
    crates/core/executor/src/dependencies.rs:230-248): they carry the Jump chip's `send_alu` to the AddSub chip."""
    ev = jump_events[jump_events["opcode"] == JUMPDIRECT]
    out = np.zeros(len(ev), dtype=ALU_EVENT)
    out["pc"] = UNUSED_PC
    out["next_pc"] = UNUSED_PC + DEFAULT_PC_INC
    out["opcode"] = ADD
    out["b"] = ev["next_pc"]
    out["c"] = ev["b"]
    out["a"] = ev["next_pc"] + ev["b"]
    return out


def synthetic_mov_cond_events(n: int, seed: int = 1) -> np.ndarray:
    """n conditional-move / byte-swap instructions with the executor's semantics: MEQ a = (c == 0 ? b : prev_a),
    MNE a = (c != 0 ? b : prev_a), WSBH a = bytes of b swapped within each halfword (prev_a = 0)."""
    raw = F.SplitMix64(0x4d4f5600 + seed).next_u64(4 * n)
    r0, r1, r2, r3 = raw[:n], raw[n:2 * n], raw[2 * n:3 * n], raw[3 * n:]
    ev = np.zeros(n, dtype=MOV_COND_EVENT)
    ev["pc"] = ((r0 % np.uint64(0x3fffff00)) & np.uint64(0xfffffffc)).astype(np.uint32)
    ev["next_pc"] = ev["pc"] + 4
    op = np.array([MEQ, MNE, WSBH], dtype=np.uint8)[(r0 >> np.uint64(40)) % np.uint64(3)]
    ev["opcode"] = op
    b = (r1 & np.uint64(0xffffffff)).astype(np.uint32)
    c = (r2 & np.uint64(0xffffffff)).astype(np.uint32)
    kind = (r3 >> np.uint64(8)) % np.uint64(8)
    c = np.where(kind < 3, np.uint32(0), c)                                  # c = 0 often
    c = np.where(kind == 3, c & np.uint32(0x00ff0000), c)                    # a single non-zero byte (or zero)
    c = np.where(kind == 4, c & np.uint32(0xff000000), c)
    prev_a = (r3 >> np.uint64(32)).astype(np.uint32)
    swapped = ((b & np.uint32(0x00ff00ff)) << np.uint32(8)) | ((b & np.uint32(0xff00ff00)) >> np.uint32(8))
    cz = c == 0
    a = np.where(op == MEQ, np.where(cz, b, prev_a), np.where(op == MNE, np.where(cz, prev_a, b), swapped))
    ev["a"], ev["b"], ev["c"] = a, b, np.where(op == WSBH, np.uint32(0), c)
    ev["prev_a"] = np.where(op == WSBH, np.uint32(0), prev_a)
    return ev


def branch_taken(ev: np.ndarray) -> np.ndarray:
    """`branching` of crates/core/machine/src/control_flow/branch/trace.rs:119-127 (a and b compared as signed words)."""
    a, b, op = ev["a"].astype(np.int32), ev["b"].astype(np.int32), ev["opcode"]
    eq, lt, gt = a == b, a < b, a > b
    return np.select([op == BEQ, op == BNE, op == BLTZ, op == BLEZ, op == BGTZ, op == BGEZ], [eq, ~eq, lt, lt | eq, gt, eq | gt],
                     default=False).astype(bool)


def synthetic_branch_events(n: int, seed: int = 1) -> np.ndarray:
    """n branch instructions: next_next_pc = next_pc + c when the branch is taken, next_pc + 4 otherwise."""
    raw = F.SplitMix64(0x42524100 + seed).next_u64(4 * n)
    r0, r1, r2, r3 = raw[:n], raw[n:2 * n], raw[2 * n:3 * n], raw[3 * n:]
    ev = np.zeros(n, dtype=BRANCH_EVENT)
    pc = ((r0 % np.uint64(0x3fffff00)) & np.uint64(0xfffffffc)).astype(np.uint32)
    ev["pc"] = pc
    ev["next_pc"] = pc + 4
    ev["opcode"] = np.array([BEQ, BGEZ, BGTZ, BLEZ, BLTZ, BNE], dtype=np.uint8)[(r0 >> np.uint64(40)) % np.uint64(6)]
    a = (r1 & np.uint64(0xffffffff)).astype(np.uint32)
    b = (r2 & np.uint64(0xffffffff)).astype(np.uint32)
    kind = (r3 >> np.uint64(8)) % np.uint64(8)
    b = np.where(kind < 2, a, b)                            # equal operands
    b = np.where(kind == 2, np.uint32(0), b)                # the *Z forms compare with register zero
    a = np.where(kind == 3, np.uint32(0x80000000), a)
    ev["a"], ev["b"] = a, b
    off = (((r3 >> np.uint64(32)) % np.uint64(0x20000)).astype(np.int64) - 0x10000) * 4        # +-256 KiB, word aligned
    ev["c"] = (off & 0xffffffff).astype(np.uint32)
    taken = branch_taken(ev)
    ev["next_next_pc"] = np.where(taken, ev["next_pc"] + ev["c"], ev["next_pc"] + 4)
    return ev


def branch_dependencies(branch_events: np.ndarray):
    """emit_branch_dependencies (crates/core/executor/src/dependencies.rs:181-227): two SLT events per branch for the Lt
    chip (a < b and b < a, signed) and, for taken branches, one ADD event for the AddSub chip. Returns (lt, add_sub)."""
    ev = branch_events
    n = len(ev)
    a, b = ev["a"].astype(np.int32), ev["b"].astype(np.int32)
    lt = np.zeros(2 * n, dtype=ALU_EVENT)
    lt["pc"] = UNUSED_PC
    lt["next_pc"] = UNUSED_PC + DEFAULT_PC_INC
    lt["opcode"] = SLT
    lt["a"][0::2], lt["b"][0::2], lt["c"][0::2] = (a < b), ev["a"], ev["b"]
    lt["a"][1::2], lt["b"][1::2], lt["c"][1::2] = (a > b), ev["b"], ev["a"]
    t = ev[branch_taken(ev)]
    add = np.zeros(len(t), dtype=ALU_EVENT)
    add["pc"] = UNUSED_PC
    add["next_pc"] = UNUSED_PC + DEFAULT_PC_INC
    add["opcode"] = ADD
    add["a"], add["b"], add["c"] = t["next_next_pc"], t["next_pc"], t["c"]
    return lt, add


# ---- Mul chip: CompAluEvents (crates/core/executor/src/events/instr.rs:50-73, 64 bytes) ------------------------------------------
MUL, MULT, MULTU = 2, 3, 4
MEMORY_WRITE_RECORD = np.dtype([("value", "<u4"), ("shard", "<u4"), ("timestamp", "<u4"), ("prev_value", "<u4"),
                                ("prev_shard", "<u4"), ("prev_timestamp", "<u4")])   # events/memory.rs:69-82
COMP_ALU_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("pc", "<u4"), ("next_pc", "<u4"), ("opcode", "u1"), ("_pad", "u1", (3,)),
                           ("hi", "<u4"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("hi_record", MEMORY_WRITE_RECORD),
                           ("hi_record_is_real", "u1"), ("_pad2", "u1", (3,))])
assert COMP_ALU_EVENT.itemsize == 64
MUL_WIDTH = 58
REGISTER_HI = 33              # the only register op_hi_access writes (alu/mul/mod.rs:478-485)
MEMORY_ACCESS_POSITION_HI = 4  # events/memory.rs:29-40


def mul_result(opcode, b, c):
    """(lo, hi) of the executor's MUL / MULT / MULTU (executor.rs:1890-1904): MUL keeps the low word and no hi."""
    b64, c64 = np.asarray(b, dtype=np.uint32).astype(np.uint64), np.asarray(c, dtype=np.uint32).astype(np.uint64)
    sb = np.asarray(b, dtype=np.uint32).astype(np.int32).astype(np.int64)
    sc = np.asarray(c, dtype=np.uint32).astype(np.int32).astype(np.int64)
    unsigned = b64 * c64
    signed = (sb * sc).astype(np.uint64)
    prod = np.where(np.asarray(opcode) == MULT, signed, unsigned)
    lo = (prod & np.uint64(0xffffffff)).astype(np.uint32)
    hi = np.where(np.asarray(opcode) == MUL, np.uint64(0), prod >> np.uint64(32)).astype(np.uint32)
    return lo, hi


def make_mul_events(opcode, b, c, pc0: int = 0x1000) -> np.ndarray:
    """`CompAluEvent::new` (instr.rs:78-93): no shard / clk, no HI-register record."""
    opcode = np.asarray(opcode, dtype=np.uint8)
    n = len(opcode)
    ev = np.zeros(n, dtype=COMP_ALU_EVENT)
    ev["pc"] = (pc0 + 4 * np.arange(n, dtype=np.uint64)) & 0x7ffffffc
    ev["next_pc"] = ev["pc"] + 4
    ev["opcode"] = opcode
    ev["b"], ev["c"] = np.asarray(b, dtype=np.uint32), np.asarray(c, dtype=np.uint32)
    ev["a"], ev["hi"] = mul_result(opcode, ev["b"], ev["c"])
    return ev


def synthetic_mul_events(n: int, seed: int = 1, shard: int = 3) -> np.ndarray:
    """n multiplications: MUL (no HI write), and MULT / MULTU of which three quarters write the HI register (the rest
    are the DivRem chip's dependency events, which carry no record: dependencies.rs). A record's previous access is in an
    earlier shard or earlier in this one, 1 .. 2^24 ticks back, so every branch of populate_access is taken."""
    raw = F.SplitMix64(0x4d554c00 + seed).next_u64(5 * n)
    r0, r1, r2, r3, r4 = (raw[i * n:(i + 1) * n] for i in range(5))
    opcode = np.array([MUL, MULT, MULTU], dtype=np.uint8)[(r0 % np.uint64(3)).astype(np.int64)]
    b = (r1 & np.uint64(0xffffffff)).astype(np.uint64)
    c = (r2 & np.uint64(0xffffffff)).astype(np.uint64)
    kind = (r3 >> np.uint64(8)) % np.uint64(16)
    pick = lambda r: _CORNERS[(r % np.uint64(len(_CORNERS))).astype(np.int64)]  # noqa: E731
    b = np.where(kind == 0, pick(r3 >> np.uint64(16)), b)
    c = np.where(kind == 1, pick(r3 >> np.uint64(24)), c)
    c = np.where(kind == 2, b, c)
    b = np.where(kind == 3, b | np.uint64(0x80000000), b)   # negative operands for MULT
    c = np.where(kind == 4, c | np.uint64(0x80000000), c)
    ev = make_mul_events(opcode, b, c)
    real = (opcode != MUL) & ((r4 & np.uint64(3)) != 0)
    clk = (np.uint64(1 << 24) + np.uint64(5) * np.arange(n, dtype=np.uint64)).astype(np.uint32)   # room for 2^24 ticks back
    ev["shard"] = np.where(real, shard, 0)
    ev["clk"] = np.where(real, clk, 0)
    rec = ev["hi_record"]
    same = ((r4 >> np.uint64(2)) & np.uint64(1)) == 1
    back = 1 + ((r4 >> np.uint64(8)) % np.uint64(1 << 24))                     # diff_minus_one spans all 24 bits
    back = np.where((r4 >> np.uint64(40)) % np.uint64(8) == 0, np.uint64(1), back)
    back = np.where((r4 >> np.uint64(40)) % np.uint64(8) == 1, np.uint64(1 << 24), back)
    ts = clk.astype(np.uint64) + MEMORY_ACCESS_POSITION_HI
    rec["value"] = np.where(real, ev["hi"], 0)
    rec["shard"] = np.where(real, shard, 0)
    rec["timestamp"] = np.where(real, ts, 0)
    rec["prev_value"] = np.where(real, (r4 >> np.uint64(16)) & np.uint64(0xffffffff), 0)
    rec["prev_shard"] = np.where(real, np.where(same, shard, (r4 >> np.uint64(48)) % np.uint64(shard)), 0)
    rec["prev_timestamp"] = np.where(real, np.where(same, ts - back, (r4 >> np.uint64(4)) % np.uint64(1 << 24)), 0)
    ev["hi_record_is_real"] = real
    return ev


# ---- DivRem chip: CompAluEvents as well (alu/divrem/mod.rs) --------------------------------------------------------------------------
DIV, DIVU, MOD, MODU = 5, 6, 7, 8
DIVREM_WIDTH = 106


def quotient_and_remainder(opcode, b, c):
    """get_quotient_and_remainder (crates/core/executor/src/utils.rs:33-43): division by zero gives (2^32 - 1, b); the signed
    forms wrap (i32::MIN / -1 = i32::MIN, remainder 0) and truncate towards zero."""
    opcode = np.asarray(opcode)
    b = np.asarray(b, dtype=np.uint32)
    c = np.asarray(c, dtype=np.uint32)
    signed = (opcode == DIV) | (opcode == MOD)
    c1 = np.where(c == 0, np.uint32(1), c)
    uq, ur = b // c1, b % c1
    sb, sc = b.astype(np.int32).astype(np.int64), c1.astype(np.int32).astype(np.int64)
    aq = np.abs(sb) // np.abs(sc)
    sq = np.where((sb < 0) != (sc < 0), -aq, aq)
    sr = sb - sq * sc
    q = np.where(signed, sq.astype(np.uint64) & np.uint64(0xffffffff), uq).astype(np.uint32)
    r = np.where(signed, sr.astype(np.uint64) & np.uint64(0xffffffff), ur).astype(np.uint32)
    return np.where(c == 0, np.uint32(0xffffffff), q).astype(np.uint32), np.where(c == 0, b, r).astype(np.uint32)


def make_divrem_events(opcode, b, c, pc0: int = 0x1000, shard: int = 3, seed: int = 1) -> np.ndarray:
    """DIV / DIVU: a = quotient, hi = remainder, written to the HI register (the record is always real, alu/divrem/mod.rs:246-255);
    MOD / MODU: a = remainder, no HI write, shard = clk = 0."""
    opcode = np.asarray(opcode, dtype=np.uint8)
    n = len(opcode)
    ev = np.zeros(n, dtype=COMP_ALU_EVENT)
    ev["pc"] = (pc0 + 4 * np.arange(n, dtype=np.uint64)) & 0x7ffffffc
    ev["next_pc"] = ev["pc"] + 4
    ev["opcode"] = opcode
    ev["b"], ev["c"] = np.asarray(b, dtype=np.uint32), np.asarray(c, dtype=np.uint32)
    q, r = quotient_and_remainder(opcode, ev["b"], ev["c"])
    div = (opcode == DIV) | (opcode == DIVU)
    ev["a"] = np.where(div, q, r)
    ev["hi"] = np.where(div, r, 0)
    r4 = F.SplitMix64(0x44495600 + seed).next_u64(n)
    clk = (np.uint64(1 << 24) + np.uint64(5) * np.arange(n, dtype=np.uint64)).astype(np.uint32)
    ts = clk.astype(np.uint64) + MEMORY_ACCESS_POSITION_HI
    same = ((r4 >> np.uint64(2)) & np.uint64(1)) == 1
    back = 1 + ((r4 >> np.uint64(8)) % np.uint64(1 << 24))
    ev["shard"] = np.where(div, shard, 0)
    ev["clk"] = np.where(div, clk, 0)
    rec = ev["hi_record"]
    rec["value"] = np.where(div, r, 0)
    rec["shard"] = np.where(div, shard, 0)
    rec["timestamp"] = np.where(div, ts, 0)
    rec["prev_value"] = np.where(div, (r4 >> np.uint64(16)) & np.uint64(0xffffffff), 0)
    rec["prev_shard"] = np.where(div, np.where(same, shard, (r4 >> np.uint64(48)) % np.uint64(shard)), 0)
    rec["prev_timestamp"] = np.where(div, np.where(same, ts - back, (r4 >> np.uint64(4)) % np.uint64(1 << 24)), 0)
    ev["hi_record_is_real"] = div
    return ev


def synthetic_divrem_events(n: int, seed: int = 1) -> np.ndarray:
    """n divisions of all four opcodes: uniform operands with the corner cases mixed in — division by zero, i32::MIN / -1,
    equal operands, small divisors (long quotients), dividends smaller than the divisor, negative operands."""
    raw = F.SplitMix64(0x44495200 + seed).next_u64(4 * n)
    r0, r1, r2, r3 = (raw[i * n:(i + 1) * n] for i in range(4))
    opcode = np.array([DIV, DIVU, MOD, MODU], dtype=np.uint8)[(r0 % np.uint64(4)).astype(np.int64)]
    b = (r1 & np.uint64(0xffffffff)).astype(np.uint64)
    c = (r2 & np.uint64(0xffffffff)).astype(np.uint64)
    kind = (r3 >> np.uint64(8)) % np.uint64(16)
    pick = lambda r: _CORNERS[(r % np.uint64(len(_CORNERS))).astype(np.int64)]  # noqa: E731
    b = np.where(kind == 0, pick(r3 >> np.uint64(16)), b)
    c = np.where(kind == 1, pick(r3 >> np.uint64(24)), c)
    c = np.where(kind == 2, b, c)
    c = np.where(kind == 3, np.uint64(0), c)
    b = np.where(kind == 4, np.uint64(0x80000000), b)
    c = np.where(kind == 4, np.uint64(0xffffffff), c)
    c = np.where((kind == 5) | (kind == 6), c & np.uint64(0xff), c)                                   # small divisors
    c = np.where(kind == 7, (np.uint64(0x100000000) - (c & np.uint64(0xffff))) & np.uint64(0xffffffff), c)   # small negative divisors
    b = np.where(kind == 8, b & np.uint64(0xffff), b)
    return make_divrem_events(opcode, b, c, seed=seed)


def divrem_dependencies(divrem_events: np.ndarray):
    """emit_divrem_dependencies (crates/core/executor/src/dependencies.rs:12-103): ADD events proving abs(c) and
    abs(remainder) for negative values (AddSub chip), one MULT / MULTU event c * quotient without a HI record (Mul chip), one
    SLTU event abs(remainder) < max(abs(c), 1) unless c = 0 (Lt chip). Returns (add_sub, mul, lt)."""
    ev = divrem_events
    n = len(ev)
    op = ev["opcode"]
    signed = (op == DIV) | (op == MOD)
    q, r = quotient_and_remainder(op, ev["b"], ev["c"])
    c = ev["c"]
    c_neg = signed & ((c >> 31) == 1)
    r_neg = signed & ((r >> 31) == 1)
    abs_c = np.where(c_neg, (~c + np.uint32(1)), c).astype(np.uint32)
    abs_r = np.where(r_neg, (~r + np.uint32(1)), r).astype(np.uint32)
    adds = []
    for i in range(n):   # order of the pushes: c first, then the remainder, event by event
        if c_neg[i]:
            adds.append((c[i], abs_c[i]))
        if r_neg[i]:
            adds.append((r[i], abs_r[i]))
    add = np.zeros(len(adds), dtype=ALU_EVENT)
    add["pc"], add["next_pc"], add["opcode"] = UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC, ADD
    if adds:
        add["b"] = [x for x, _ in adds]
        add["c"] = [y for _, y in adds]
    mul = np.zeros(n, dtype=COMP_ALU_EVENT)
    mul["pc"], mul["next_pc"] = UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC
    mul["opcode"] = np.where(signed, MULT, MULTU)
    mul["b"], mul["c"] = q, c
    mul["a"], mul["hi"] = mul_result(mul["opcode"], q, c)
    keep = c != 0
    lt = np.zeros(int(keep.sum()), dtype=ALU_EVENT)
    lt["pc"], lt["next_pc"], lt["opcode"] = UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC, SLTU
    lt["a"] = 1
    lt["b"] = abs_r[keep]
    lt["c"] = np.maximum(abs_c[keep], 1)
    return add, mul, lt


# ---- memory instructions: MemInstrEvent (crates/core/executor/src/events/instr.rs:108-136, #[repr(C)], 64 bytes) -----------------------
LB, LBU, LH, LHU, LW, LWL, LWR, LL, SB, SH, SW, SWL, SWR, SC = range(31, 45)
LOADS, STORES = (LB, LBU, LH, LHU, LW, LWL, LWR, LL), (SB, SH, SW, SWL, SWR, SC)
# mem_access is the #[repr(C)] enum MemoryRecordEnum (events/memory.rs:88-95): a 4-byte tag (Read = 0, Write = 1) and the union of
# MemoryReadRecord (value, shard, timestamp, prev_shard, prev_timestamp) and MemoryWriteRecord (value, shard, timestamp, prev_value,
# prev_shard, prev_timestamp)
MEM_INSTR_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("pc", "<u4"), ("next_pc", "<u4"), ("opcode", "u1"), ("_pad", "u1", (3,)),
                            ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("mem_tag", "<u4"), ("mem", "<u4", (6,)), ("prev_a_val", "<u4")])
assert MEM_INSTR_EVENT.itemsize == 64
MEMORY_INSTRS_WIDTH = 79
NUM_REGISTERS = 36      # crates/core/executor/src/register.rs:3


def load_value(opcode: int, mem: int, addr: int, rt: int) -> int:
    """The value a load writes to rt (execute_load, executor.rs:1925-2000)."""
    i = addr & 3
    if opcode == LB:
        v = (mem >> (8 * i)) & 0xff
        return v | 0xffffff00 if v & 0x80 else v
    if opcode == LBU:
        return (mem >> (8 * i)) & 0xff
    if opcode in (LH, LHU):
        v = (mem >> (8 * (i & 2))) & 0xffff
        return v | 0xffff0000 if opcode == LH and v & 0x8000 else v
    if opcode in (LW, LL):
        return mem
    if opcode == LWL:
        sh = 24 - 8 * i
        return ((rt & ~(0xffffffff << sh)) | (mem << sh)) & 0xffffffff
    if opcode == LWR:
        sh = 8 * i
        return ((rt & ~(0xffffffff >> sh)) | (mem >> sh)) & 0xffffffff
    raise ValueError(opcode)


def store_value(opcode: int, mem: int, addr: int, rt: int) -> int:
    """The word a store leaves in memory (execute_store, executor.rs:2002-2088)."""
    i = addr & 3
    if opcode == SB:
        return (mem & ~(0xff << (8 * i)) | ((rt & 0xff) << (8 * i))) & 0xffffffff
    if opcode == SH:
        j = i & 2
        return (mem & ~(0xffff << (8 * j)) | ((rt & 0xffff) << (8 * j))) & 0xffffffff
    if opcode in (SW, SC):
        return rt
    if opcode == SWL:
        sh = 24 - 8 * i
        return ((mem & ~(0xffffffff >> sh)) | (rt >> sh)) & 0xffffffff
    if opcode == SWR:
        sh = 8 * i
        return ((mem & ~((0xffffffff << sh) & 0xffffffff)) | ((rt << sh) & 0xffffffff)) & 0xffffffff
    raise ValueError(opcode)


def memory_dependencies(mem_events: np.ndarray) -> np.ndarray:
    """emit_memory_dependencies (crates/core/executor/src/dependencies.rs:125-178): per event an ADD proving addr = b + c, and
    for LB / LH of a negative value a SUB proving the sign extension (a = unsigned - 2^8 or 2^16); AddSub chip events in order."""
    out = []
    for e in mem_events:
        addr = (int(e["b"]) + int(e["c"])) & 0xffffffff
        out.append((UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC, ADD, [0, 0, 0], 0, addr, int(e["b"]), int(e["c"])))
        op = int(e["opcode"])
        if op in (LB, LH):
            mem = int(e["mem"][0])
            off = addr & 3
            if op == LB:
                unsigned, msb, sign = (mem >> (8 * off)) & 0xff, (mem >> (8 * off)) & 0xff, 256
            else:
                unsigned = (mem >> (8 * (off & 2))) & 0xffff
                msb, sign = unsigned >> 8, 65536
            if msb >> 7:
                out.append((UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC, SUB, [0, 0, 0], 0, int(e["a"]), unsigned, sign))
    return np.array(out, dtype=ALU_EVENT) if out else np.zeros(0, dtype=ALU_EVENT)


# ---- syscall instructions: SyscallEvent (crates/core/executor/src/events/syscall.rs:7-29, #[repr(C)], 56 bytes) -----------------------
SYSCALL = 30
SYSCALL_EVENT = np.dtype([("pc", "<u4"), ("next_pc", "<u4"), ("shard", "<u4"), ("clk", "<u4"), ("a_record", MEMORY_WRITE_RECORD),
                          ("a_record_is_real", "u1"), ("_pad", "u1", (3,)), ("syscall_id", "<u4"), ("arg1", "<u4"), ("arg2", "<u4")])
assert SYSCALL_EVENT.itemsize == 56
# the syscall tables: SyscallCore / SyscallPrecompile (crates/core/machine/src/syscall/chip.rs:71-107), eleven columns
SYSCALL_WIDTH = 11
SYS_POSEIDON2_PERMUTE = 0x00010030      # SyscallCode::POSEIDON2_PERMUTE (syscalls/code.rs:182): id 0x30, send-to-table byte set, no extra cycles
# MemoryInitializeFinalizeEvent (crates/core/executor/src/events/memory.rs:180-209, #[repr(C)]): addr, value, shard, timestamp
MEMORY_INIT_FINALIZE_EVENT = np.dtype([("addr", "<u4"), ("value", "<u4"), ("shard", "<u4"), ("timestamp", "<u4")])
MEMORY_GLOBAL_WIDTH = 111               # MemoryInitCols (crates/core/machine/src/memory/global.rs:221-259)
# Poseidon2PermuteEvent (events/precompiles/poseidon2_permute.rs:9-27) flattened for the C ABI: shard, clk, state_addr and the sixteen
# MemoryWriteRecords of the state words (pre_state = prev_value, post_state = value)
POSEIDON2_PERMUTE_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("state_addr", "<u4"), ("state_records", MEMORY_WRITE_RECORD, (16,))])
assert POSEIDON2_PERMUTE_EVENT.itemsize == 4 * 99
POSEIDON2_PERMUTE_WIDTH = 973           # Poseidon2MemCols (syscall/precompiles/poseidon2/columns.rs:9-27)
# KeccakSpongeEvent (events/precompiles/keccak_sponge.rs:15-46) holds Vecs; across the C ABI it is cut into its 36-word blocks, one record
# per block (24 trace rows each): the state after the block is xored in (xored_state_list[block_index], as u32 pairs), the block's
# MemoryReadRecords (their values are the input words) and, used by the first / last block only, the record of the input length (read at
# output_addr + 64) and the sixteen output MemoryWriteRecords.
SYS_KECCAK_SPONGE = 0x01010009          # SyscallCode::KECCAK_SPONGE (syscalls/code.rs:57): id 9, own table, one extra cycle
MEMORY_READ_RECORD = np.dtype([("value", "<u4"), ("shard", "<u4"), ("timestamp", "<u4"), ("prev_shard", "<u4"), ("prev_timestamp", "<u4")])
KECCAK_SPONGE_BLOCK = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("input_addr", "<u4"), ("output_addr", "<u4"), ("input_len_u32s", "<u4"),
                                ("block_index", "<u4"), ("xored_state", "<u4", (50,)), ("input_read_records", MEMORY_READ_RECORD, (36,)),
                                ("input_length_record", MEMORY_READ_RECORD), ("output_write_records", MEMORY_WRITE_RECORD, (16,))])
assert KECCAK_SPONGE_BLOCK.itemsize == 4 * 337
KECCAK_SPONGE_WIDTH = 3531              # KeccakSpongeCols (syscall/precompiles/keccak_sponge/columns.rs:17-37): 2633 KeccakCols + 898
KECCAK_RATE_U32S, KECCAK_OUTPUT_U32S, NUM_KECCAK_COLS = 36, 16, 2633
KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b, 0x0000000080000001,
             0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088, 0x0000000080008009, 0x000000008000000a,
             0x000000008000808b, 0x800000000000008b, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
             0x000000000000800a, 0x800000008000000a, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
KECCAK_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]   # [x][y]


def keccak_f(state):
    """keccak-f[1600] on 25 u64 lanes state[x + 5 * y] (tiny-keccak's keccakf, which KeccakSpongeSyscall::execute calls)."""
    M = (1 << 64) - 1
    rotl = lambda v, r: ((v << r) | (v >> (64 - r))) & M if r else v   # noqa: E731
    a = list(state)
    for rc in KECCAK_RC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], KECCAK_ROT[x][y])
        a = [b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & M & b[(x + 2) % 5 + 5 * y]) for y in range(5) for x in range(5)]
        a[0] ^= rc
    return a


def keccak256_words(data: bytes):
    """The guest library's keccak256 (crates/zkvm/lib/src/keccak256.rs:3-57) up to the syscall: pad10*1 to the 136-byte rate, little-endian
    words, every 34-word block extended by two zero words to the precompile's general 36-word rate."""
    n = len(data)
    padded = bytearray(data) + bytearray(136 - n % 136)
    if n % 136 == 135:
        padded[-1] = 0x81
    else:
        padded[n] = 1
        padded[-1] = 0x80
    words = []
    for k in range(0, len(padded), 136):
        words += [int.from_bytes(padded[k + 4 * i:k + 4 * i + 4], "little") for i in range(34)] + [0, 0]
    return words


# The SHA-256 precompiles (syscall/precompiles/sha256/): SHA_EXTEND fills w[16..64] of a message schedule in place (48 extra cycles, one
# per word), SHA_COMPRESS runs the 64 rounds on the state at h_ptr (one extra cycle). Their events hold Vecs of fixed length: flattened.
SYS_SHA_EXTEND, SYS_SHA_COMPRESS = 0x30010005, 0x01010006     # syscalls/code.rs:45,48
SHA_EXTEND_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("w_ptr", "<u4"), ("w_i_minus_15_reads", MEMORY_READ_RECORD, (48,)),
                             ("w_i_minus_2_reads", MEMORY_READ_RECORD, (48,)), ("w_i_minus_16_reads", MEMORY_READ_RECORD, (48,)),
                             ("w_i_minus_7_reads", MEMORY_READ_RECORD, (48,)), ("w_i_writes", MEMORY_WRITE_RECORD, (48,))])
assert SHA_EXTEND_EVENT.itemsize == 4 * 1251
SHA_COMPRESS_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("w_ptr", "<u4"), ("h_ptr", "<u4"), ("h_read_records", MEMORY_READ_RECORD, (8,)),
                               ("w_i_read_records", MEMORY_READ_RECORD, (64,)), ("h_write_records", MEMORY_WRITE_RECORD, (8,))])
assert SHA_COMPRESS_EVENT.itemsize == 4 * 412
SHA_EXTEND_WIDTH, SHA_COMPRESS_WIDTH = 176, 262         # ShaExtendCols (extend/columns.rs:17-73), ShaCompressCols (compress/columns.rs:17-108)
SHA_COMPRESS_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
SHA256_IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def sha_extend(w):
    """sha_extend (extend/mod.rs:19-25): w[16..64] of the message schedule from w[0..16]."""
    M = 0xffffffff
    rr = lambda x, n: ((x >> n) | (x << (32 - n))) & M   # noqa: E731
    w = list(w[:16]) + [0] * 48
    for i in range(16, 64):
        s0 = rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3)
        s1 = rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10)
        w[i] = (w[i - 16] + s0 + w[i - 7] + s1) & M
    return w


def sha_compress(h, w):
    """Sha256CompressSyscall::execute's arithmetic (syscalls/precompiles/sha256/compress.rs:52-91): the state after 64 rounds, added to h."""
    M = 0xffffffff
    rr = lambda x, n: ((x >> n) | (x << (32 - n))) & M   # noqa: E731
    a, b, c, d, e, f, g, hh = h
    for i in range(64):
        s1 = rr(e, 6) ^ rr(e, 11) ^ rr(e, 25)
        ch = (e & f) ^ (~e & M & g)
        temp1 = (hh + s1 + ch + SHA_COMPRESS_K[i] + w[i]) & M
        s0 = rr(a, 2) ^ rr(a, 13) ^ rr(a, 22)
        maj = (a & b) ^ (a & c) ^ (b & c)
        temp2 = (s0 + maj) & M
        hh, g, f, e, d, c, b, a = g, f, e, (d + temp1) & M, c, b, a, (temp1 + temp2) & M
    return [(x + y) & M for x, y in zip(h, [a, b, c, d, e, f, g, hh])]


# EdAddAssign (syscall/precompiles/edwards/ed_add.rs): EllipticCurveAddEvent (events/precompiles/ec.rs:24-47) flattened — p and q are the previous
# values of the p write records and the values of the q read records
SYS_ED_ADD = 0x01010007                 # syscalls/code.rs:51: one extra cycle (p is written at clk + 1)
ED_ADD_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("p_ptr", "<u4"), ("q_ptr", "<u4"), ("p_memory_records", MEMORY_WRITE_RECORD, (16,)),
                         ("q_memory_records", MEMORY_READ_RECORD, (16,))])
assert ED_ADD_EVENT.itemsize == 4 * 180
ED_ADD_WIDTH = 5 + 16 * 13 + 16 * 9 + 8 * 188       # EdAddAssignCols (ed_add.rs:41-57): 1861
ED25519_P = (1 << 255) - 19
ED25519_D = 37095705934669439343138083508754565189542113879843219016388785533085940283555


def ed25519_add(p, q):
    """AffinePoint<Ed25519> + AffinePoint (curves/src/edwards/mod.rs ed_add): the complete twisted-Edwards addition, a = -1."""
    P, D = ED25519_P, ED25519_D
    (x1, y1), (x2, y2) = p, q
    f = x1 * x2 * y1 * y2 % P
    x3 = (x1 * y2 + x2 * y1) * pow(1 + D * f, P - 2, P) % P
    y3 = (y1 * y2 + x1 * x2) * pow(1 - D * f, P - 2, P) % P
    return x3, y3


# EdDecompress (syscall/precompiles/edwards/ed_decompress.rs): EdDecompressEvent (events/precompiles/edwards.rs:13-32) flattened — y_bytes and
# decompressed_x_bytes are the values of the y read records and of the x write records
SYS_ED_DECOMPRESS = 0x00010008          # syscalls/code.rs:54: no extra cycle
ED_DECOMPRESS_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("ptr", "<u4"), ("sign", "<u4"), ("x_memory_records", MEMORY_WRITE_RECORD, (8,)),
                                ("y_memory_records", MEMORY_READ_RECORD, (8,))])
assert ED_DECOMPRESS_EVENT.itemsize == 4 * 92
ED_DECOMPRESS_WIDTH = 1566              # EdDecompressCols (ed_decompress.rs:39-57)
ED25519_SQRT_M1 = 19681161376707505956807079304988542015446066515923890162744021073123829784752


def ed25519_sqrt(a):
    """ed25519_sqrt (curves/src/edwards/ed25519.rs:75-113): the even square root of a, None when there is none."""
    P = ED25519_P
    beta = pow(a, (P + 3) // 8, P)
    sq = beta * beta % P
    if sq == (P - a) % P:
        beta = beta * ED25519_SQRT_M1 % P
    elif sq != a:
        return None
    return (P - beta) % P if beta & 1 else beta


def ed25519_decompress(y, sign):
    """decompress (curves/src/edwards/ed25519.rs:115-141): x from y and the sign bit."""
    P, D = ED25519_P, ED25519_D
    yy = y * y % P
    x = ed25519_sqrt((yy - 1) * pow(D * yy + 1, P - 2, P) % P)
    if x is None:
        return None
    return (P - x) % P if sign else x


# The short-Weierstrass precompiles (syscall/precompiles/weierstrass/weierstrass_add.rs, weierstrass_double.rs), one chip per curve and operation
# (MipsAir variants Secp256k1Add .. Bls12381Double, mips/mod.rs:137-158): parameters as crates/curves/src/weierstrass/{secp256k1,secp256r1,bn254,
# bls12_381}.rs give them; `index` is the curve number across the C ABI (ZKM_CURVE_*).
WEIERSTRASS_CURVES = {
    "Secp256k1": dict(index=0, p=(1 << 256) - (1 << 32) - 977, a=0, b=7, n_limbs=32, witness_offset=1 << 14, add=0x0101000A, double=0x0001000B,
                      generator=(55066263022277343669578718895168534326250603453777594175500187360389116729240,
                                 32670510020758816978083085130507043184471273380659243275938904335757337482424)),
    "Secp256r1": dict(index=1, p=(1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 1, a=(1 << 256) - (1 << 224) + (1 << 192) + (1 << 96) - 4,
                      b=0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B, n_limbs=32, witness_offset=1 << 14, add=0x0101002C, double=0x0001002D,
                      generator=(48439561293906451759052585252797914202762949526041747995844080717082404635286,
                                 36134250956749795798585127919587881956611106672985015071877198253568414405109)),
    "Bn254": dict(index=2, p=21888242871839275222246405745257275088696311157297823662689037894645226208583, a=0, b=3, n_limbs=32, witness_offset=1 << 14,
                  add=0x0101000E, double=0x0001000F, generator=(1, 2)),
    "Bls12381": dict(index=3, p=0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, a=0, b=4, n_limbs=48,
                     witness_offset=1 << 15, add=0x0101001E, double=0x0001001F,
                     generator=(3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
                                1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569)),
}


def weierstrass_event_dtypes(curve):
    """EllipticCurveAddEvent / EllipticCurveDoubleEvent (events/precompiles/ec.rs:24-72) flattened for a curve: p (and q) are the previous values of
    the p write records (the values of the q read records)."""
    w = WEIERSTRASS_CURVES[curve]["n_limbs"] // 2
    add = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("p_ptr", "<u4"), ("q_ptr", "<u4"), ("p_memory_records", MEMORY_WRITE_RECORD, (w,)),
                    ("q_memory_records", MEMORY_READ_RECORD, (w,))])
    double = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("p_ptr", "<u4"), ("p_memory_records", MEMORY_WRITE_RECORD, (w,))])
    return add, double


def weierstrass_widths(curve):
    """WeierstrassAddAssignCols (weierstrass_add.rs:43-62): 5 + W * 13 + W * 9 + 9 gadgets; WeierstrassDoubleAssignCols (weierstrass_double.rs:43-62):
    4 + W * 13 + 11 gadgets; a gadget is 2 N + 2 (2 N - 2) columns."""
    n = WEIERSTRASS_CURVES[curve]["n_limbs"]
    w, g = n // 2, 6 * n - 4
    return 5 + 22 * w + 9 * g, 4 + 13 * w + 11 * g


def weierstrass_add(curve, p, q):
    """AffinePoint + AffinePoint on a short Weierstrass curve (curves/src/weierstrass/mod.rs sw_add): distinct x coordinates."""
    P = WEIERSTRASS_CURVES[curve]["p"]
    (x1, y1), (x2, y2) = p, q
    slope = (y2 - y1) * pow(x2 - x1, P - 2, P) % P
    x3 = (slope * slope - x1 - x2) % P
    return x3, (slope * (x1 - x3) - y1) % P


def weierstrass_double(curve, p):
    c = WEIERSTRASS_CURVES[curve]
    P, (x1, y1) = c["p"], p
    slope = (3 * x1 * x1 + c["a"]) * pow(2 * y1, P - 2, P) % P
    x3 = (slope * slope - 2 * x1) % P
    return x3, (slope * (x1 - x3) - y1) % P


# <Curve>Decompress (syscall/precompiles/weierstrass/weierstrass_decompress.rs): y from x and a sign bit on Secp256k1, Secp256r1 (the bit is y's
# parity: SignChoiceRule::LeastSignificantBit) and Bls12381 (the bit says y > p - y: Lexicographic, mips/mod.rs:240,259,347). The syscall reads x at
# ptr + N and writes y at ptr, both at clk (create_ec_decompress_event, events/precompiles/ec.rs:181-228).
WEIERSTRASS_DECOMPRESS = {"Secp256k1": dict(code=0x0001000C, lexicographic=False), "Secp256r1": dict(code=0x0001002E, lexicographic=False),
                          "Bls12381": dict(code=0x0001001C, lexicographic=True)}


def weierstrass_decompress_event_dtype(curve):
    """EllipticCurveDecompressEvent (events/precompiles/ec.rs:74-94) flattened: x is the values of the x read records, the decompressed y the
    values of the y write records."""
    w = WEIERSTRASS_CURVES[curve]["n_limbs"] // 4
    return np.dtype([("shard", "<u4"), ("clk", "<u4"), ("ptr", "<u4"), ("sign_bit", "<u4"), ("x_memory_records", MEMORY_READ_RECORD, (w,)),
                     ("y_memory_records", MEMORY_WRITE_RECORD, (w,))])


def weierstrass_decompress_width(curve):
    """WeierstrassDecompressCols (weierstrass_decompress.rs:52-68): 5 + 9 W + 13 W, range_x (N + 2), x_2, x_3, ax_plus_b, x_3_plus_b_plus_ax (G each),
    y (FieldSqrtCols: G + N + 2 + 1), neg_y (G); the lexicographic rule adds LexicographicChoiceCols (:73-79): two FieldLtCols and three flags."""
    n = WEIERSTRASS_CURVES[curve]["n_limbs"]
    w, g = n // 4, 6 * n - 4
    return 5 + 22 * w + (n + 2) + 4 * g + (g + n + 3) + g + (2 * (n + 2) + 3 if WEIERSTRASS_DECOMPRESS[curve]["lexicographic"] else 0)


def weierstrass_sqrt(curve, a):
    """secp256k1_sqrt / secp256r1_sqrt / bls12381_sqrt (curves/src/weierstrass/*.rs): the root the k256 / p256 / amcl field types return. Those
    crates are not in the tree; for all three moduli p = 3 (mod 4) and the root is a^((p + 1) / 4) (unpinned: either root satisfies the AIR)."""
    P = WEIERSTRASS_CURVES[curve]["p"]
    r = pow(a, (P + 1) // 4, P)
    if r * r % P != a % P:
        raise ValueError("not a square")
    return r


def weierstrass_decompress(curve, x, sign_bit):
    """The y the syscall writes: the root whose parity is the bit (Secp256k1 / Secp256r1), or the larger of y, p - y when the bit is set (Bls12381)."""
    c = WEIERSTRASS_CURVES[curve]
    P = c["p"]
    r = weierstrass_sqrt(curve, (x * x * x + c["a"] * x + c["b"]) % P)
    if WEIERSTRASS_DECOMPRESS[curve]["lexicographic"]:
        return max(r, P - r) if sign_bit else min(r, P - r)
    return r if (r & 1) == sign_bit else P - r


# Uint256MulMod (syscall/precompiles/uint256/air.rs): x <- x * y mod m for 256-bit integers, m = 0 meaning 2^256. The syscall reads y and, right
# after it, the modulus at clk and writes x at clk + 1 (syscalls/precompiles/uint256.rs:14-97); Uint256MulEvent (events/precompiles/uint256.rs:12-35)
# flattened: x is the previous values of the x write records, y and the modulus the values of their read records.
SYS_UINT256_MUL = 0x0101001D
UINT256_MUL_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("x_ptr", "<u4"), ("y_ptr", "<u4"), ("x_memory_records", MEMORY_WRITE_RECORD, (8,)),
                              ("y_memory_records", MEMORY_READ_RECORD, (8,)), ("modulus_memory_records", MEMORY_READ_RECORD, (8,))])
assert UINT256_MUL_EVENT.itemsize == 4 * 132
UINT256_MUL_WIDTH = 480      # Uint256MulCols (uint256/air.rs:57-91): 4 + 8 * 13 + 2 * 8 * 9 + IsZero 2 + 1 + FieldOpCols<U256Field> (32 + 32 + 63 + 63) + FieldLtCols 34 + 1


def uint256_mulmod(x, y, modulus):
    return x * y % (modulus if modulus else 1 << 256)


# U256XU2048Mul (syscall/precompiles/u256x2048_mul/air.rs): a 256-bit a times a 2048-bit b; the low 2048 bits go to the address in $a2, the high 256
# to the address in $a3. The syscall reads the two registers, a and b at clk and writes lo and hi at clk + 1
# (syscalls/precompiles/u256x2048_mul.rs:20-93); U256xU2048MulEvent (events/precompiles/u256x2048_mul.rs:9-44) flattened.
SYS_U256XU2048_MUL = 0x0101002F
REG_A2, REG_A3 = 6, 7
U256X2048_MUL_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("a_ptr", "<u4"), ("b_ptr", "<u4"), ("lo_ptr", "<u4"), ("hi_ptr", "<u4"),
                                ("lo_ptr_memory", MEMORY_READ_RECORD), ("hi_ptr_memory", MEMORY_READ_RECORD), ("a_memory_records", MEMORY_READ_RECORD, (8,)),
                                ("b_memory_records", MEMORY_READ_RECORD, (64,)), ("lo_memory_records", MEMORY_WRITE_RECORD, (64,)),
                                ("hi_memory_records", MEMORY_WRITE_RECORD, (8,))])
assert U256X2048_MUL_EVENT.itemsize == 4 * 808
U256X2048_MUL_WIDTH = 3129   # U256x2048MulCols (u256x2048_mul/air.rs:52-87): 6 + 2 * 9 + 8 * 9 + 64 * 9 + 64 * 13 + 8 * 13 + 8 FieldOpCols<U256Field> (190) + 1


# BooleanCircuitGarble (syscall/precompiles/boolean_circuit_garble/): checks the ciphertexts of the non-free gates of a garbled circuit. At
# input_ptr: the number of gates, delta (4 words), then 17 words per gate (gate type — 0 AND, 7 OR —, h0, h1, label_b, expected, 4 words each);
# 1 is written to output_ptr when every gate's h0 ^ h1 ^ label_b (^ delta for OR) equals its expected ciphertext, else 0; everything at clk
# (syscalls/precompiles/boolean_circuit/garble.rs:10-95). A call takes 1 + num_gates rows. Across the ABI the BooleanCircuitGarbleEvent
# (events/precompiles/boolean_circuit_garble.rs:12-45, Vecs inside) is cut into its rows: the header row (is_gate 0: the reads of num_gates and
# delta in the first five records) and one per gate (its seventeen reads; the write record counts on the last gate). `pre_check` is the
# conjunction of the gates before this one.
SYS_BOOLEAN_CIRCUIT_GARBLE = 0x00010031
GARBLE_OR_GATE = 7
GARBLE_ROW = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("input_address", "<u4"), ("output_address", "<u4"), ("is_gate", "<u4"), ("gate_id", "<u4"),
                       ("gates_num", "<u4"), ("pre_check", "<u4"), ("delta", "<u4", (4,)), ("reads", MEMORY_READ_RECORD, (17,)), ("write", MEMORY_WRITE_RECORD)])
assert GARBLE_ROW.itemsize == 4 * 103
GARBLE_WIDTH = 292      # BooleanCircuitGarbleCols (boolean_circuit_garble/columns.rs:10-35)


def garble_gate_ok(gate_words, delta):
    """One gate's check: gate_words = [type, h0 x 4, h1 x 4, label_b x 4, expected x 4]."""
    t, h0, h1, lb, want = gate_words[0], gate_words[1:5], gate_words[5:9], gate_words[9:13], gate_words[13:17]
    return all((h0[i] ^ h1[i] ^ lb[i] ^ (delta[i] if t else 0)) == want[i] for i in range(4))


# SysLinux (syscall/precompiles/sys_linux/): the Linux syscalls a MIPS guest's runtime makes — the codes with a non-zero second byte (4003 read, 4004
# write, 4045 brk, 4055 fcntl, 4090 mmap2, 4120 clone, 4210 mmap, 4246 exit_group; every other one is a no-op returning 0). Each returns a value
# in $v0 and writes $a3 (0, or 9 = EBADF); brk reads register BRK (34), write reads $a2, mmap with a0 = 0 bumps register HEAP (35)
# (syscalls/precompiles/sys_linux/*.rs). LinuxEvent (events/precompiles/linux.rs:9-28) flattened: its one read record (brk, write) and its
# write records (the $a3 write, and HEAP's for mmap with a0 = 0); the ones an event does not have are zero.
SYS_READ, SYS_WRITE_LINUX, SYS_BRK, SYS_FCNTL, SYS_MMAP2, SYS_CLONE, SYS_MMAP, SYS_OPEN = 4003, 4004, 4045, 4055, 4090, 4120, 4210, 4005
REG_BRK, REG_HEAP = 34, 35
LINUX_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("a0", "<u4"), ("a1", "<u4"), ("v0", "<u4"), ("syscall_code", "<u4"), ("read_record", MEMORY_READ_RECORD),
                        ("a3_record", MEMORY_WRITE_RECORD), ("heap_record", MEMORY_WRITE_RECORD)])
assert LINUX_EVENT.itemsize == 4 * 23
SYS_LINUX_WIDTH = 103      # SysLinuxCols (sys_linux/columns.rs:20-82)


def linux_syscall(code, a0, a1, brk=0, heap=0, a2=0):
    """What a Linux syscall does (syscalls/precompiles/sys_linux/): returns (v0, the value written to $a3, the new HEAP or None)."""
    if code == SYS_BRK:
        return max(a0, brk), 0, None
    if code in (SYS_MMAP, SYS_MMAP2):
        size = a1 if a1 & 0xfff == 0 else (a1 + 0x1000 - (a1 & 0xfff)) & 0xffffffff
        return (heap, 0, (heap + size) & 0xffffffff) if a0 == 0 else (a0, 0, None)
    if code == SYS_CLONE:
        return 1, 0, None
    if code == SYS_EXT_GROUP:
        return 0, 0, None
    if code == SYS_FCNTL:
        if a1 == 3:
            return (0, 0, None) if a0 == 0 else (1, 0, None) if a0 in (1, 2) else (0xffffffff, 9, None)
        if a1 == 1:
            return (a0, 0, None) if a0 in (0, 1, 2) else (0xffffffff, 9, None)
        return 0xffffffff, 9, None
    if code == SYS_READ:
        return (0, 0, None) if a0 == 0 else (0xffffffff, 9, None)
    if code == SYS_WRITE_LINUX:
        return a2, 0, None
    return 0, 0, None


# The field-tower precompiles (syscall/precompiles/fptower/fp.rs, fp2_addsub.rs, fp2_mul.rs) over the base fields of Bn254 and Bls12381: FpOpEvent,
# Fp2AddSubEvent, Fp2MulEvent (events/precompiles/fptower.rs:23-94) flattened; `op` is FieldOperation as a word (Add 0, Mul 1, Sub 2).
FP_TOWER_CODES = {"Bn254": dict(fp_add=0x01010026, fp_sub=0x01010027, fp_mul=0x01010028, fp2_add=0x01010029, fp2_sub=0x0101002A, fp2_mul=0x0101002B),
                  "Bls12381": dict(fp_add=0x01010020, fp_sub=0x01010021, fp_mul=0x01010022, fp2_add=0x01010023, fp2_sub=0x01010024, fp2_mul=0x01010025)}
FIELD_OP_ADD, FIELD_OP_MUL, FIELD_OP_SUB = 0, 1, 2


def fp_tower_event_dtype(field, kind):
    n = WEIERSTRASS_CURVES[field]["n_limbs"]
    w = n // 4 if kind == "fp" else n // 2
    head = [("shard", "<u4"), ("clk", "<u4"), ("x_ptr", "<u4"), ("y_ptr", "<u4")] + ([] if kind == "fp2_mul" else [("op", "<u4")])
    return np.dtype(head + [("x_memory_records", MEMORY_WRITE_RECORD, (w,)), ("y_memory_records", MEMORY_READ_RECORD, (w,))])


def fp_tower_width(field, kind):
    """FpOpCols (fp.rs:40-54): 8 + 22 W + G with W = N / 4 words; Fp2AddSubAssignCols (fp2_addsub.rs:40-51): 6 + 22 W + 2 G, Fp2MulAssignCols
    (fp2_mul.rs:40-54): 5 + 22 W + 6 G with W = N / 2."""
    n = WEIERSTRASS_CURVES[field]["n_limbs"]
    g = 6 * n - 4
    return {"fp": 8 + 22 * (n // 4) + g, "fp2_addsub": 6 + 22 * (n // 2) + 2 * g, "fp2_mul": 5 + 22 * (n // 2) + 6 * g}[kind]


def fp_tower_result(field, kind, op, x, y):
    """What the syscalls write (syscalls/precompiles/fptower/fp.rs:52-59, fp2_addsub.rs:52-70, fp2_mul.rs:52-75): x and y are ints (fp) or pairs."""
    P = WEIERSTRASS_CURVES[field]["p"]
    f = {FIELD_OP_ADD: lambda a, b: (a + b) % P, FIELD_OP_SUB: lambda a, b: (a - b) % P, FIELD_OP_MUL: lambda a, b: a * b % P}
    if kind == "fp":
        return f[op](x, y)
    if kind == "fp2_addsub":
        return f[op](x[0], y[0]), f[op](x[1], y[1])
    return (x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P


SYSCALL_INSTRS_WIDTH = 77
# SyscallCode (crates/core/executor/src/syscalls/code.rs): byte 0-1 id, byte 2 "has its own table", byte 3 extra cycles
SYS_HALT, SYS_WRITE, SYS_ENTER_UNCONSTRAINED, SYS_EXIT_UNCONSTRAINED, SYS_COMMIT, SYS_COMMIT_DEFERRED_PROOFS, SYS_HINT_LEN = 0, 2, 3, 4, 0x10, 0x1a, 0xf0
SYS_EXT_GROUP = 4246
REG_V0, REG_A0, REG_A1 = 2, 4, 5


# ---- misc instructions: MiscEvent (crates/core/executor/src/events/instr.rs:239-261, #[repr(C)], 60 bytes) ---------------------------
INS, MADDU, MSUBU, MADD, MSUB = 45, 46, 47, 48, 49
EXT, TEQ, SEXT = 53, 54, 55
MISC_EVENT = np.dtype([("shard", "<u4"), ("clk", "<u4"), ("pc", "<u4"), ("next_pc", "<u4"), ("opcode", "u1"), ("_pad", "u1", (3,)),
                       ("a", "<u4"), ("b", "<u4"), ("c", "<u4"), ("prev_a", "<u4"), ("hi_record", MEMORY_WRITE_RECORD)])
assert MISC_EVENT.itemsize == 60
MISC_INSTRS_WIDTH = 72


def misc_result(opcode: int, b: int, c: int, prev_a: int = 0, hi_lo=(0, 0)):
    """The executor's results for the MiscInstrs opcodes (executor.rs:1684-1826): SEXT / EXT / INS return `a`; MADDU / MSUBU / MADD /
    MSUB return (lo, hi) of HI:LO +- b * c."""
    M32 = 0xffffffff
    if opcode == SEXT:
        v = b & 0xffff if c > 0 else b & 0xff
        sign = v >> (15 if c > 0 else 7)
        return (v | (M32 ^ (0xffff if c > 0 else 0xff))) & M32 if sign else v
    if opcode == EXT:
        msbd, lsb = c >> 5, c & 0x1f
        return (b & ((1 << (msbd + lsb + 1)) - 1)) >> lsb
    if opcode == INS:
        msb, lsb = c >> 5, c & 0x1f
        field = (((1 << (msb - lsb + 1)) - 1) << lsb) & M32
        return (prev_a & ~field & M32) | ((b << lsb) & field)
    signed = opcode in (MADD, MSUB)
    sb, sc = (b - (1 << 32) if signed and b >> 31 else b), (c - (1 << 32) if signed and c >> 31 else c)
    addend = (hi_lo[0] << 32) + hi_lo[1]
    out = (addend + sb * sc if opcode in (MADDU, MADD) else addend - sb * sc) & ((1 << 64) - 1)
    return out & M32, out >> 32


def misc_dependencies(misc_events: np.ndarray):
    """emit_misc_dependencies (crates/core/executor/src/dependencies.rs:251-388): MADD* -> one MULT / MULTU event (Mul chip); EXT -> SLL
    (ShiftLeft) + SRL (ShiftRight); INS -> ROR, SRL, SRL (ShiftRight), SLL (ShiftLeft), ADD (AddSub), ROR (ShiftRight). Returns
    (mul, shift_left, shift_right, add_sub) event arrays in emission order."""
    mul, sll, sr, add = [], [], [], []
    alu = lambda op, a, b, c: (UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC, op, [0, 0, 0], 0, a & 0xffffffff, b, c)   # noqa: E731
    for e in misc_events:
        op, a, b, c, prev_a = int(e["opcode"]), int(e["a"]), int(e["b"]), int(e["c"]), int(e["prev_a"])
        if op in (MADDU, MSUBU, MADD, MSUB):
            mop = MULTU if op in (MADDU, MSUBU) else MULT
            lo, hi = mul_result(np.array([mop]), np.array([b], dtype=np.uint32), np.array([c], dtype=np.uint32))
            mul.append((0, 0, UNUSED_PC, UNUSED_PC + DEFAULT_PC_INC, mop, [0, 0, 0], int(hi[0]), int(lo[0]), b, c, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0]))
        elif op == EXT:
            lsb, msbd = c & 0x1f, c >> 5
            v = (b << (31 - lsb - msbd)) & 0xffffffff
            sll.append(alu(SLL, v, b, 31 - lsb - msbd))
            sr.append(alu(SRL, a, v, 31 - msbd))
        elif op == INS:
            lsb, msb = c & 0x1f, c >> 5
            ror = ((prev_a >> lsb) | (prev_a << (32 - lsb))) & 0xffffffff if lsb else prev_a
            srl1 = ror >> 1
            srl = srl1 >> (msb - lsb)
            v = (b << (31 - msb + lsb)) & 0xffffffff
            sr += [alu(ROR, ror, prev_a, lsb), alu(SRL, srl1, ror, 1), alu(SRL, srl, srl1, msb - lsb)]
            sll.append(alu(SLL, v, b, 31 - msb + lsb))
            add.append(alu(ADD, srl + v, srl, v))
            sr.append(alu(ROR, a, (srl + v) & 0xffffffff, 31 - msb))
    arr = lambda rows, dt: np.array(rows, dtype=dt) if rows else np.zeros(0, dtype=dt)   # noqa: E731
    return arr(mul, COMP_ALU_EVENT), arr(sll, ALU_EVENT), arr(sr, ALU_EVENT), arr(add, ALU_EVENT)
