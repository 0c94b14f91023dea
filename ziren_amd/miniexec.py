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
    """One shard of a generated program: (program, record, public values). See _execute."""
    m = _execute(n_cycles, seed, shard, pc_base, halt)
    return m.program, m.shards[0].record, m.shards[0].pv


class Shard:
    """One ExecutionRecord of a run: kind "cpu" (instructions), "precompile" (deferred precompile events) or "memory" (global
    memory initialisation / finalisation), its events and the public values its proof carries."""

    def __init__(self, kind, record, pv):
        self.kind, self.record, self.pv = kind, record, pv


class Machine:
    def __init__(self, program, shards, pc_base):
        self.program, self.shards, self.pc_base = program, shards, pc_base


def run_machine(n_cycles: int = 0, seed: int = 1, pc_base: int = 0x1000, shard_cycles: int = 1 << 30, program=None, poseidon2_calls: int = 0,
                memory_chunk: int = 1 << 30, keccak_calls: int = 0, sha_calls: int = 0, ed_calls: int = 0,
                curve_calls=None, fp_calls=None, decompress_calls=None, uint256_calls: int = 0, u2048_calls: int = 0,
                garble_calls=(), linux_calls=()) -> Machine:
    """A whole run as the reference's prover sees it (crates/core/machine/src/utils/prove.rs:255-400): CPU shards of at most `shard_cycles`
    cycles (never split between a branch and its delay slot, executor.rs:2352-2356), then one shard with the deferred precompile events
    (ExecutionRecord::split, record.rs:130-218), then the shards that initialise and finalise every touched address
    (executor.rs:2554-2618, record.rs:220-277; `memory_chunk` events per shard). `program`: a list of (opcode, op_a, op_b, op_c, imm_b, imm_c)
    to execute until it halts instead of generating one; `poseidon2_calls` / `keccak_calls`: POSEIDON2_PERMUTE / KECCAK_SPONGE precompile calls
    spread over a generated run (each kind is deferred to a precompile shard of its own, record.rs:150-185)."""
    return _execute(n_cycles, seed, 1, pc_base, True, shard_cycles=shard_cycles, given=program, poseidon2_calls=poseidon2_calls, memory_chunk=memory_chunk,
                    machine=True, keccak_calls=keccak_calls, sha_calls=sha_calls, ed_calls=ed_calls, curve_calls=curve_calls, fp_calls=fp_calls, decompress_calls=decompress_calls,
                    uint256_calls=uint256_calls, u2048_calls=u2048_calls, garble_calls=garble_calls,
                    linux_calls=linux_calls)


def _execute(n_cycles: int, seed: int = 1, shard: int = 1, pc_base: int = 0x1000, halt: bool = False, shard_cycles: int = 1 << 30, given=None,
             poseidon2_calls: int = 0, memory_chunk: int = 1 << 30, machine: bool = False, keccak_calls: int = 0, sha_calls: int = 0, ed_calls: int = 0,
             curve_calls=None, fp_calls=None, decompress_calls=None, uint256_calls: int = 0, u2048_calls: int = 0,
             garble_calls=(), linux_calls=()) -> Machine:
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
    born = {}                      # address -> the value it held before its first access of the whole run (what MemoryGlobalInit sends)
    program = {}
    rec = Record()
    done_shards = []               # (Record lists, first pc, last next_pc, shard number) of the CPU shards already closed
    precompile = []                # (kind, syscall event, the precompile's event(s), its MemoryLocalEvents) in execution order
    if given is not None:
        for i, ins in enumerate(given):
            program[pc_base + 4 * i] = tuple(ins)
    pc, next_pc = pc_base, pc_base + 4
    delay_slot = False
    pending_jump_reg = None        # a register just loaded with a jump target

    def read(reg, clk, pos):
        if reg not in R:
            R[reg], last[reg] = 0, (0, 0)
        born.setdefault(reg, R[reg])
        first.setdefault(reg, (last[reg][0], last[reg][1], R[reg]))
        r = (TAG_READ, (R[reg], shard, clk + pos, last[reg][0], last[reg][1]), None)
        last[reg] = (shard, clk + pos)
        return r

    def write(reg, value, clk, pos):
        value = 0 if reg == 0 else value & 0xffffffff
        if reg not in R:
            R[reg], last[reg] = 0, (0, 0)
        born.setdefault(reg, R[reg])
        first.setdefault(reg, (last[reg][0], last[reg][1], R[reg]))
        r = (TAG_WRITE, None, (value, shard, clk + pos, R[reg], last[reg][0], last[reg][1]))
        R[reg] = value
        last[reg] = (shard, clk + pos)
        return r

    digest = [int(x) for x in rng.integers(0, 1 << 32, 8, dtype=np.uint64)]
    committed = {}                 # digest index -> word, as the COMMIT syscalls publish them
    epilogue = []
    if halt:   # li $v0, code; li $a0, arg1; li $a1, arg2; syscall — eight commits, then halt with exit code 0
        for i, w in enumerate(digest):
            epilogue += [(E.ADD, E.REG_V0, E.SYS_COMMIT, 0, 1, 1), (E.ADD, E.REG_A0, i, 0, 1, 1), (E.ADD, E.REG_A1, w, 0, 1, 1),
                         (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
        epilogue += [(E.ADD, E.REG_V0, E.SYS_HALT, 0, 1, 1), (E.ADD, E.REG_A0, 0, 0, 1, 1), (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
    cyc = -1
    shard_start = 0                # global cycle at which the current shard began (its clk is 0 there)
    halted = False
    queued = []                    # instructions that must come next (a precompile call's set-up), generated runs only
    p2_at = set(int(x) for x in np.linspace(n_cycles // 8, max(n_cycles - 40, n_cycles // 8), poseidon2_calls)) if poseidon2_calls else set()
    p2_seq = 0
    k_at = set(int(x) for x in np.linspace(n_cycles // 6, max(n_cycles - 60, n_cycles // 6), keccak_calls)) - p2_at if keccak_calls else set()
    k_seq = 0
    s_at = set(int(x) for x in np.linspace(n_cycles // 5, max(n_cycles - 80, n_cycles // 5), sha_calls)) - p2_at - k_at if sha_calls else set()
    s_seq = 0
    e_at = {n_cycles // 4} - p2_at - k_at - s_at if ed_calls else set()
    w_at = {n_cycles // 3} - p2_at - k_at - s_at - e_at if curve_calls or decompress_calls or uint256_calls or u2048_calls or garble_calls or linux_calls else set()
    f_at = {n_cycles // 2} - p2_at - k_at - s_at - e_at - w_at if fp_calls else set()
    clk_extra = 0                  # the extra cycles of the shard's syscalls so far (Syscall::num_extra_cycles, executor.rs:1641)

    def close_shard():
        """bump_record (executor.rs:2186-2200): the open access chains become the shard's MemoryLocal events."""
        nonlocal rec, shard, shard_start, clk_extra
        rec.memory_local = [(reg, first[reg], (last[reg][0], last[reg][1], R[reg])) for reg in sorted(first)] + rec.memory_local
        done_shards.append((rec, shard))
        first.clear()
        rec = Record()
        shard += 1
        shard_start = cyc
        clk_extra = 0

    while (not halted) if given is not None else (cyc + 1 < n_cycles + len(epilogue)):
        cyc += 1
        if machine and cyc - shard_start >= shard_cycles and not delay_slot and not queued and (given is not None or cyc < n_cycles):
            close_shard()
        clk = 5 * (cyc - shard_start) + clk_extra
        u = rng.random()
        if given is None and cyc >= n_cycles and delay_slot:      # a branch's delay slot comes first: one more plain instruction
            epilogue.insert(cyc - n_cycles, (E.ADD, 1, 0, 0, 1, 1))
        reg = lambda: int(rng.integers(0, 32))       # noqa: E731
        dst = lambda: int(rng.integers(1, 32)) if rng.random() > 0.02 else 0   # noqa: E731  (a few writes to $0)
        if given is None and cyc in p2_at and cyc < n_cycles:
            # a POSEIDON2_PERMUTE call as the reference's test program makes it (syscall/precompiles/poseidon2/mod.rs:24-43): sixteen field
            # elements stored to the state, the code in $v0, the state's address in $a0, zero in $a1, SYSCALL
            had = len(queued)
            ptr = 0x00200000 + 64 * p2_seq
            p2_seq += 1
            for i in range(16):
                queued += [(E.ADD, 30, int(rng.integers(0, F.P)), 0, 1, 1), (E.SW, 30, 0, ptr + 4 * i, 0, 1)]
            queued += [(E.ADD, E.REG_V0, E.SYS_POSEIDON2_PERMUTE, 0, 1, 1), (E.ADD, E.REG_A0, ptr, 0, 1, 1), (E.ADD, E.REG_A1, 0, 0, 1, 1),
                       (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            n_cycles += len(queued) - had  # the body grows by the call; the epilogue still follows it
            p2_at = set(x + len(queued) - had if x > cyc else x for x in p2_at)
            k_at = set(x + len(queued) - had if x > cyc else x for x in k_at)
            s_at = set(x + len(queued) - had if x > cyc else x for x in s_at)
            e_at = set(x + len(queued) - had if x > cyc else x for x in e_at)
            w_at = set(x + len(queued) - had if x > cyc else x for x in w_at)
            f_at = set(x + len(queued) - had if x > cyc else x for x in f_at)
        if given is None and cyc in k_at and cyc < n_cycles:
            # a KECCAK_SPONGE call as the guest library's keccak256 makes it (crates/zkvm/lib/src/keccak256.rs:3-57): the padded message as
            # 36-word blocks, its length in words at result + 64, the code in $v0, input and result pointers in $a0 / $a1
            had = len(queued)
            msg = bytes(rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8))
            words = E.keccak256_words(msg)
            in_ptr, out_ptr = 0x00300000 + 0x1000 * k_seq, 0x00380000 + 0x100 * k_seq
            k_seq += 1
            for i, w in enumerate(words):
                queued += [(E.ADD, 30, w, 0, 1, 1), (E.SW, 30, 0, in_ptr + 4 * i, 0, 1)]
            queued += [(E.ADD, 30, len(words), 0, 1, 1), (E.SW, 30, 0, out_ptr + 64, 0, 1),
                       (E.ADD, E.REG_V0, E.SYS_KECCAK_SPONGE, 0, 1, 1), (E.ADD, E.REG_A0, in_ptr, 0, 1, 1), (E.ADD, E.REG_A1, out_ptr, 0, 1, 1),
                       (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            n_cycles += len(queued) - had
            p2_at = set(x + len(queued) - had if x > cyc else x for x in p2_at)
            k_at = set(x + len(queued) - had if x > cyc else x for x in k_at)
            s_at = set(x + len(queued) - had if x > cyc else x for x in s_at)
            e_at = set(x + len(queued) - had if x > cyc else x for x in e_at)
            w_at = set(x + len(queued) - had if x > cyc else x for x in w_at)
            f_at = set(x + len(queued) - had if x > cyc else x for x in f_at)
        if given is None and cyc in s_at and cyc < n_cycles:
            # one SHA-256 block as the reference's test programs lay the calls out (sha256/extend/mod.rs:44-61, compress/mod.rs:52-78): sixteen
            # message words at w_ptr, SHA_EXTEND(w_ptr, 0), the eight state words at h_ptr, SHA_COMPRESS(w_ptr, h_ptr)
            had = len(queued)
            w_ptr, h_ptr = 0x00400000 + 0x200 * s_seq, 0x00480000 + 0x40 * s_seq
            s_seq += 1
            for i in range(16):
                queued += [(E.ADD, 30, int(rng.integers(0, 1 << 32)), 0, 1, 1), (E.SW, 30, 0, w_ptr + 4 * i, 0, 1)]
            queued += [(E.ADD, E.REG_V0, E.SYS_SHA_EXTEND, 0, 1, 1), (E.ADD, E.REG_A0, w_ptr, 0, 1, 1), (E.ADD, E.REG_A1, 0, 0, 1, 1),
                       (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            for i, word in enumerate(E.SHA256_IV):
                queued += [(E.ADD, 30, word, 0, 1, 1), (E.SW, 30, 0, h_ptr + 4 * i, 0, 1)]
            queued += [(E.ADD, E.REG_V0, E.SYS_SHA_COMPRESS, 0, 1, 1), (E.ADD, E.REG_A0, w_ptr, 0, 1, 1), (E.ADD, E.REG_A1, h_ptr, 0, 1, 1),
                       (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            n_cycles += len(queued) - had
            p2_at = set(x + len(queued) - had if x > cyc else x for x in p2_at)
            k_at = set(x + len(queued) - had if x > cyc else x for x in k_at)
            s_at = set(x + len(queued) - had if x > cyc else x for x in s_at)
            e_at = set(x + len(queued) - had if x > cyc else x for x in e_at)
            w_at = set(x + len(queued) - had if x > cyc else x for x in w_at)
            f_at = set(x + len(queued) - had if x > cyc else x for x in f_at)
        if given is None and cyc in e_at and cyc < n_cycles:
            # Ed25519 additions the way a scalar multiplication makes them: p = B and q = 2B stored once, then `ed_calls` times p <- p + q
            # (ED_ADD(p_ptr, q_ptr), the reference's ed_add test program: syscall/precompiles/edwards/ed_add.rs tests)
            had = len(queued)
            p_ptr, q_ptr = 0x00500000, 0x00500100
            base = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
                    46316835694926478169428394003475163141307993866256225615783033603165251855960)
            dbl = E.ed25519_add(base, base)
            for ptr, pt in ((p_ptr, base), (q_ptr, dbl)):      # only y is stored; x comes from ED_DECOMPRESS(ptr, sign) as a verifier gets A and R
                for i in range(8, 16):
                    queued += [(E.ADD, 30, (pt[1] >> (32 * (i % 8))) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, ptr + 4 * i, 0, 1)]
                queued += [(E.ADD, E.REG_V0, E.SYS_ED_DECOMPRESS, 0, 1, 1), (E.ADD, E.REG_A0, ptr, 0, 1, 1), (E.ADD, E.REG_A1, pt[0] & 1, 0, 1, 1),
                           (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            for _ in range(ed_calls):
                queued += [(E.ADD, E.REG_V0, E.SYS_ED_ADD, 0, 1, 1), (E.ADD, E.REG_A0, p_ptr, 0, 1, 1), (E.ADD, E.REG_A1, q_ptr, 0, 1, 1),
                           (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            n_cycles += len(queued) - had
            p2_at = set(x + len(queued) - had if x > cyc else x for x in p2_at)
            k_at = set(x + len(queued) - had if x > cyc else x for x in k_at)
            s_at = set(x + len(queued) - had if x > cyc else x for x in s_at)
            w_at = set(x + len(queued) - had if x > cyc else x for x in w_at)
            f_at = set(x + len(queued) - had if x > cyc else x for x in f_at)
        if given is None and cyc in w_at and cyc < n_cycles:
            # per curve in `curve_calls`: the generator stored at p and at q, q doubled with <CURVE>_DOUBLE(q), then `count` times p <- p + q
            # with <CURVE>_ADD(p, q): p runs through G, 3G, 5G, ... and never meets q = 2G
            had = len(queued)
            for k, (curve, calls) in enumerate((curve_calls or {}).items()):
                cv = E.WEIERSTRASS_CURVES[curve]
                W = cv["n_limbs"] // 2
                p_ptr, q_ptr = 0x00600000 + 0x400 * k, 0x00600200 + 0x400 * k
                for ptr in (p_ptr, q_ptr):
                    for i in range(W):
                        queued += [(E.ADD, 30, (cv["generator"][i // (W // 2)] >> (32 * (i % (W // 2)))) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, ptr + 4 * i, 0, 1)]
                queued += [(E.ADD, E.REG_V0, cv["double"], 0, 1, 1), (E.ADD, E.REG_A0, q_ptr, 0, 1, 1), (E.ADD, E.REG_A1, 0, 0, 1, 1),
                           (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
                for _ in range(calls):
                    queued += [(E.ADD, E.REG_V0, cv["add"], 0, 1, 1), (E.ADD, E.REG_A0, p_ptr, 0, 1, 1), (E.ADD, E.REG_A1, q_ptr, 0, 1, 1),
                               (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            # per curve in `decompress_calls`: the generator's x stored at ptr + N, then `count` times <CURVE>_DECOMPRESS(ptr, sign) with the sign
            # bit alternating: y at ptr is the generator's y or its negative
            for k, (curve, calls) in enumerate((decompress_calls or {}).items()):
                cv = E.WEIERSTRASS_CURVES[curve]
                ptr = 0x00680000 + 0x400 * k
                for i in range(cv["n_limbs"] // 4):
                    queued += [(E.ADD, 30, (cv["generator"][0] >> (32 * i)) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, ptr + cv["n_limbs"] + 4 * i, 0, 1)]
                for j in range(calls):
                    queued += [(E.ADD, E.REG_V0, E.WEIERSTRASS_DECOMPRESS[curve]["code"], 0, 1, 1), (E.ADD, E.REG_A0, ptr, 0, 1, 1), (E.ADD, E.REG_A1, j & 1, 0, 1, 1),
                               (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            # `uint256_calls` times x <- x * y mod m with UINT256_MUL(x_ptr, y_ptr): y, and after it the modulus, at y_ptr; the modulus alternates
            # between 2^256 - 189 and zero (which stands for 2^256); y is below 2^256 - 189 so that the quotient always fits 256 bits
            if uint256_calls:
                big = (1 << 256) - 189
                x_ptr, y_ptr = 0x006c0000, 0x006c0100
                for ptr, v in ((x_ptr, int.from_bytes(rng.bytes(32), "little") % big), (y_ptr, int.from_bytes(rng.bytes(32), "little") % big)):
                    for i in range(8):
                        queued += [(E.ADD, 30, (v >> (32 * i)) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, ptr + 4 * i, 0, 1)]
                for j in range(uint256_calls):
                    for i in range(8):
                        queued += [(E.ADD, 30, ((0 if j & 1 else big) >> (32 * i)) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, y_ptr + 32 + 4 * i, 0, 1)]
                    queued += [(E.ADD, E.REG_V0, E.SYS_UINT256_MUL, 0, 1, 1), (E.ADD, E.REG_A0, x_ptr, 0, 1, 1), (E.ADD, E.REG_A1, y_ptr, 0, 1, 1),
                               (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            # `u2048_calls` times U256XU2048_MUL(a_ptr, b_ptr) with $a2 = lo_ptr, $a3 = hi_ptr: a random a (256 bits) and b (2048 bits) each time
            for j in range(u2048_calls):
                a_ptr, b_ptr, lo_ptr, hi_ptr = 0x006d0000, 0x006d0100, 0x006d0400, 0x006d0800
                for ptr, n_words in ((a_ptr, 8), (b_ptr, 64)):
                    v = int.from_bytes(rng.bytes(4 * n_words), "little")
                    for i in range(n_words):
                        queued += [(E.ADD, 30, (v >> (32 * i)) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, ptr + 4 * i, 0, 1)]
                queued += [(E.ADD, E.REG_A2, lo_ptr, 0, 1, 1), (E.ADD, E.REG_A3, hi_ptr, 0, 1, 1), (E.ADD, E.REG_V0, E.SYS_U256XU2048_MUL, 0, 1, 1),
                           (E.ADD, E.REG_A0, a_ptr, 0, 1, 1), (E.ADD, E.REG_A1, b_ptr, 0, 1, 1), (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            # per entry of `garble_calls` (a gate count; negative: one expected ciphertext of that many gates is wrong): the gate count, a random
            # delta and the gates (type, h0, h1, label_b, expected) stored at input_ptr, then BOOLEAN_CIRCUIT_GARBLE(input_ptr, output_ptr)
            for j, n_gates in enumerate(garble_calls):
                input_ptr, output_ptr = 0x006e0000 + 0x1000 * j, 0x006f0000 + 4 * j
                delta = [int(rng.integers(0, 1 << 32)) for _ in range(4)]
                words = [abs(n_gates)] + delta
                for g in range(abs(n_gates)):
                    t = E.GARBLE_OR_GATE if rng.random() < 0.5 else 0
                    h0, h1, lb = ([int(rng.integers(0, 1 << 32)) for _ in range(4)] for _ in range(3))
                    want = [h0[i] ^ h1[i] ^ lb[i] ^ (delta[i] if t else 0) for i in range(4)]
                    if n_gates < 0 and g == abs(n_gates) // 2:
                        want[1] ^= 0x00010000
                    words += [t] + h0 + h1 + lb + want
                for i, w in enumerate(words):
                    queued += [(E.ADD, 30, w, 0, 1, 1), (E.SW, 30, 0, input_ptr + 4 * i, 0, 1)]
                queued += [(E.ADD, E.REG_V0, E.SYS_BOOLEAN_CIRCUIT_GARBLE, 0, 1, 1), (E.ADD, E.REG_A0, input_ptr, 0, 1, 1), (E.ADD, E.REG_A1, output_ptr, 0, 1, 1),
                           (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            # per entry of `linux_calls` — (code, a0, a1) or (code, a0, a1, a2) — a Linux syscall: $v0 = code, $a0, $a1 (and $a2) set, SYSCALL
            for call in linux_calls:
                if len(call) > 3:
                    queued += [(E.ADD, E.REG_A2, call[3], 0, 1, 1)]
                queued += [(E.ADD, E.REG_V0, call[0], 0, 1, 1), (E.ADD, E.REG_A0, call[1], 0, 1, 1), (E.ADD, E.REG_A1, call[2], 0, 1, 1),
                           (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            n_cycles += len(queued) - had
            p2_at = set(x + len(queued) - had if x > cyc else x for x in p2_at)
            k_at = set(x + len(queued) - had if x > cyc else x for x in k_at)
            s_at = set(x + len(queued) - had if x > cyc else x for x in s_at)
            e_at = set(x + len(queued) - had if x > cyc else x for x in e_at)
            f_at = set(x + len(queued) - had if x > cyc else x for x in f_at)
        if given is None and cyc in f_at and cyc < n_cycles:
            # per field in `fp_calls`: two random Fp2 elements x, y stored, then `count` calls cycling through FP_ADD, FP_SUB, FP_MUL (on the first
            # components), FP2_ADD, FP2_SUB, FP2_MUL; every call overwrites x with its result
            had = len(queued)
            for k, (field, calls) in enumerate(fp_calls.items()):
                cv = E.WEIERSTRASS_CURVES[field]
                per = cv["n_limbs"] // 4
                x_ptr, y_ptr = 0x00700000 + 0x400 * k, 0x00700200 + 0x400 * k
                for ptr in (x_ptr, y_ptr):
                    for comp in range(2):
                        v = int.from_bytes(rng.bytes(cv["n_limbs"]), "little") % cv["p"]
                        for i in range(per):
                            queued += [(E.ADD, 30, (v >> (32 * i)) & 0xffffffff, 0, 1, 1), (E.SW, 30, 0, ptr + 4 * (comp * per + i), 0, 1)]
                names = ["fp_add", "fp_sub", "fp_mul", "fp2_add", "fp2_sub", "fp2_mul"]
                for j in range(calls):
                    queued += [(E.ADD, E.REG_V0, E.FP_TOWER_CODES[field][names[j % 6]], 0, 1, 1), (E.ADD, E.REG_A0, x_ptr, 0, 1, 1),
                               (E.ADD, E.REG_A1, y_ptr, 0, 1, 1), (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
            n_cycles += len(queued) - had
            p2_at = set(x + len(queued) - had if x > cyc else x for x in p2_at)
            k_at = set(x + len(queued) - had if x > cyc else x for x in k_at)
            s_at = set(x + len(queued) - had if x > cyc else x for x in s_at)
            e_at = set(x + len(queued) - had if x > cyc else x for x in e_at)
            w_at = set(x + len(queued) - had if x > cyc else x for x in w_at)
        # ---- pick the instruction at pc (the program is written as it runs)
        if given is not None:
            if pc not in program:
                if len(rec.cpu) or done_shards:
                    break           # the run is done when the pc leaves the program (executor.rs:2172-2176), HALT or not: simple_program() ends so
                raise RuntimeError(f"miniexec: pc {pc:#x} is outside the program")
            ins = program[pc]
        elif queued and not delay_slot and pending_jump_reg is None:
            ins = queued.pop(0)
        elif cyc >= n_cycles:
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
        assert given is not None or pc not in program
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
            w_curve = {c[k]: (name, k == "double") for name, c in E.WEIERSTRASS_CURVES.items() for k in ("add", "double")}.get(code)
            fp_call = {c: (field, name) for field, codes in E.FP_TOWER_CODES.items() for name, c in codes.items()}.get(code)
            w_decompress = {d["code"]: name for name, d in E.WEIERSTRASS_DECOMPRESS.items()}.get(code)
            is_linux = (code >> 8) & 0xff != 0 and code < 0x10000
            assert is_linux or w_curve or fp_call or w_decompress or code in (E.SYS_UINT256_MUL, E.SYS_U256XU2048_MUL, E.SYS_BOOLEAN_CIRCUIT_GARBLE) or code in (E.SYS_HALT, E.SYS_COMMIT, E.SYS_POSEIDON2_PERMUTE, E.SYS_KECCAK_SPONGE, E.SYS_SHA_EXTEND, E.SYS_SHA_COMPRESS, E.SYS_ED_ADD,
                                        E.SYS_ED_DECOMPRESS), code
            touched = {}                               # address -> [initial (shard, timestamp, value), final]: SyscallContext's local map

            def mem(addr, ts, value=None):
                if addr not in R:
                    R[addr], last[addr] = 0, (0, 0)
                born.setdefault(addr, R[addr])
                if addr in first:                      # an open CPU access chain is closed first (SyscallContext::postprocess)
                    rec.memory_local.append((addr, first.pop(addr), (last[addr][0], last[addr][1], R[addr])))
                prev = (last[addr][0], last[addr][1], R[addr])
                touched.setdefault(addr, [prev, None])
                if value is None:
                    out = (R[addr], shard, ts, prev[0], prev[1])
                else:
                    out = (value, shard, ts, prev[2], prev[0], prev[1])
                    R[addr] = value
                last[addr] = (shard, ts)
                touched[addr][1] = (shard, ts, R[addr])
                return out

            if code == E.SYS_POSEIDON2_PERMUTE:
                # Poseidon2PermuteSyscall::execute (syscalls/precompiles/poseidon2/permute.rs:14-71): the sixteen words at $a0 are
                # replaced by their permutation, written at timestamp clk through the syscall's own local-access map; a CPU access
                # chain that is open on one of these words is closed first (SyscallContext::postprocess, context.rs:128-147)
                from .recursion import poseidon2_permute
                assert c == 0 and b % 4 == 0
                pre = [R.get(b + 4 * i, 0) for i in range(16)]
                assert max(pre) < F.P
                post = poseidon2_permute(pre)
                records, local = [], []
                for i in range(16):
                    addr = b + 4 * i
                    if addr not in R:
                        R[addr], last[addr] = 0, (0, 0)
                    born.setdefault(addr, R[addr])
                    if addr in first:
                        rec.memory_local.append((addr, first.pop(addr), (last[addr][0], last[addr][1], R[addr])))
                    records.append((post[i], shard, clk, pre[i], last[addr][0], last[addr][1]))
                    local.append((addr, (last[addr][0], last[addr][1], pre[i]), (shard, clk, post[i])))
                    R[addr], last[addr] = post[i], (shard, clk)
                precompile.append(("poseidon2", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, records)], local))
            if code == E.SYS_KECCAK_SPONGE:
                # KeccakSpongeSyscall::execute (syscalls/precompiles/keccak/sponge.rs:20-104): the input length is read from result + 64 and
                # the input from $a0 at timestamp clk, every 36-word block is xored into the state and permuted, the first sixteen words of
                # the state are written to the result at clk + 1; the call takes one extra cycle
                assert b % 4 == 0 and c % 4 == 0
                len_rec = mem(c + 64, clk)
                n_words = len_rec[0]
                assert n_words and n_words % E.KECCAK_RATE_U32S == 0
                reads = [mem(b + 4 * j, clk) for j in range(n_words)]
                state, xored = [0] * 25, []
                for k in range(0, n_words, 36):
                    for i in range(18):
                        state[i] ^= reads[k + 2 * i][0] | (reads[k + 2 * i + 1][0] << 32)
                    xored.append([w for lane in state for w in (lane & 0xffffffff, lane >> 32)])
                    state = E.keccak_f(state)
                writes = [mem(c + 4 * j, clk + 1, (state[j // 2] >> (32 * (j & 1))) & 0xffffffff) for j in range(16)]
                nb = n_words // 36
                no_read, no_write = (0, 0, 0, 0, 0), (0, 0, 0, 0, 0, 0)
                blocks = [(shard, clk, b, c, n_words, k, xored[k], reads[36 * k:36 * k + 36], len_rec if k == 0 else no_read,
                           writes if k == nb - 1 else [no_write] * 16) for k in range(nb)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("keccak", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), blocks, local))
                clk_extra += 1
            if code == E.SYS_SHA_EXTEND:
                # Sha256ExtendSyscall::execute (syscalls/precompiles/sha256/extend.rs:16-74): w[16..64] of the schedule at $a0, one word per
                # cycle — four reads and the write at timestamp clk + (i - 16); 48 extra cycles
                assert c == 0 and b % 4 == 0
                r15, r2, r16, r7, wr = [], [], [], [], []
                M32 = 0xffffffff
                rr = lambda x, k: ((x >> k) | (x << (32 - k))) & M32      # noqa: E731
                for i in range(16, 64):
                    ts = clk + i - 16
                    r15.append(mem(b + 4 * (i - 15), ts)); w15 = r15[-1][0]
                    r2.append(mem(b + 4 * (i - 2), ts)); w2 = r2[-1][0]
                    r16.append(mem(b + 4 * (i - 16), ts))
                    r7.append(mem(b + 4 * (i - 7), ts))
                    s0 = rr(w15, 7) ^ rr(w15, 18) ^ (w15 >> 3)
                    s1 = rr(w2, 17) ^ rr(w2, 19) ^ (w2 >> 10)
                    wr.append(mem(b + 4 * i, ts, (s1 + r16[-1][0] + s0 + r7[-1][0]) & M32))
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("sha_extend", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, r15, r2, r16, r7, wr)], local))
                clk_extra += 48
            if code == E.SYS_SHA_COMPRESS:
                # Sha256CompressSyscall::execute (syscalls/precompiles/sha256/compress.rs:34-118): the state at $a1 and the 64 schedule words
                # at $a0 are read at clk, the compressed state is added and written back at clk + 1; one extra cycle
                assert b != c and b % 4 == 0 and c % 4 == 0
                hr = [mem(c + 4 * i, clk) for i in range(8)]
                wr_ = [mem(b + 4 * i, clk) for i in range(64)]
                out = E.sha_compress([x[0] for x in hr], [x[0] for x in wr_])
                hw = [mem(c + 4 * i, clk + 1, out[i]) for i in range(8)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("sha_compress", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, c, hr, wr_, hw)], local))
                clk_extra += 1
            if code == E.SYS_ED_ADD:
                # create_ec_add_event (events/precompiles/ec.rs:96-139) for Ed25519: p is peeked, q is read at clk, p + q is written over p at
                # clk + 1 (p and q may be the same words); one extra cycle
                assert b % 4 == 0 and c % 4 == 0
                pw = [R.get(b + 4 * i, 0) for i in range(16)]
                qr = [mem(c + 4 * i, clk) for i in range(16)]
                as_int = lambda ws: sum(w << (32 * i) for i, w in enumerate(ws))      # noqa: E731
                x3, y3 = E.ed25519_add((as_int(pw[:8]), as_int(pw[8:])), (as_int([x[0] for x in qr[:8]]), as_int([x[0] for x in qr[8:]])))
                out = [(x3 >> (32 * i)) & 0xffffffff for i in range(8)] + [(y3 >> (32 * i)) & 0xffffffff for i in range(8)]
                pwr = [mem(b + 4 * i, clk + 1, out[i]) for i in range(16)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("ed_add", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, c, pwr, qr)], local))
                clk_extra += 1
            if code == E.SYS_ED_DECOMPRESS:
                # EdwardsDecompressSyscall::execute (syscalls/precompiles/edwards/decompress.rs:33-83): y is read at ptr + 32, the x its sign
                # bit ($a1) selects is written at ptr, both at clk; no extra cycle
                assert b % 4 == 0 and c <= 1
                yr = [mem(b + 32 + 4 * i, clk) for i in range(8)]
                x = E.ed25519_decompress(sum(rec_[0] << (32 * i) for i, rec_ in enumerate(yr)), c)
                assert x is not None
                xw = [mem(b + 4 * i, clk, (x >> (32 * i)) & 0xffffffff) for i in range(8)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("ed_decompress", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, c, xw, yr)], local))
            if w_curve:
                # create_ec_add_event / create_ec_double_event (events/precompiles/ec.rs:96-176) for a short-Weierstrass curve: p is peeked; an
                # addition reads q at clk and writes p + q over p at clk + 1 (one extra cycle), a doubling writes 2 p over p at clk
                curve, dbl = w_curve
                W = E.WEIERSTRASS_CURVES[curve]["n_limbs"] // 2
                as_pt = lambda ws: (sum(w << (32 * i) for i, w in enumerate(ws[:W // 2])), sum(w << (32 * i) for i, w in enumerate(ws[W // 2:])))      # noqa: E731
                pw = [R.get(b + 4 * i, 0) for i in range(W)]
                if dbl:
                    out_pt, qr = E.weierstrass_double(curve, as_pt(pw)), []
                else:
                    qr = [mem(c + 4 * i, clk) for i in range(W)]
                    out_pt = E.weierstrass_add(curve, as_pt(pw), as_pt([x[0] for x in qr]))
                out = [(out_pt[i // (W // 2)] >> (32 * (i % (W // 2)))) & 0xffffffff for i in range(W)]
                pwr = [mem(b + 4 * i, clk + (0 if dbl else 1), out[i]) for i in range(W)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                event = (shard, clk, b, pwr) if dbl else (shard, clk, b, c, pwr, qr)
                precompile.append((curve + ("_double" if dbl else "_add"), (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [event], local))
                clk_extra += 0 if dbl else 1
            if code == E.SYS_UINT256_MUL:
                # Uint256MulSyscall::execute (syscalls/precompiles/uint256.rs:14-97): x is peeked, y and the modulus right after it are read at clk,
                # x * y mod modulus (mod 2^256 for a zero modulus) is written over x at clk + 1; one extra cycle
                assert b % 4 == 0 and c % 4 == 0
                as_int = lambda ws: sum(w << (32 * i) for i, w in enumerate(ws))      # noqa: E731
                xw = [R.get(b + 4 * i, 0) for i in range(8)]
                yr = [mem(c + 4 * i, clk) for i in range(8)]
                mr = [mem(c + 32 + 4 * i, clk) for i in range(8)]
                res = E.uint256_mulmod(as_int(xw), as_int([x[0] for x in yr]), as_int([x[0] for x in mr]))
                xwr = [mem(b + 4 * i, clk + 1, (res >> (32 * i)) & 0xffffffff) for i in range(8)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("uint256_mul", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, c, xwr, yr, mr)], local))
                clk_extra += 1
            if code == E.SYS_U256XU2048_MUL:
                # U256xU2048MulSyscall::execute (syscalls/precompiles/u256x2048_mul.rs:20-93): $a2 and $a3 (lo_ptr, hi_ptr), a and b are read at clk; the
                # low 2048 bits of a * b are written at lo_ptr and the high 256 at hi_ptr at clk + 1; one extra cycle
                assert b % 4 == 0 and c % 4 == 0
                as_int = lambda ws: sum(w << (32 * i) for i, w in enumerate(ws))      # noqa: E731
                lo_reg, hi_reg = mem(E.REG_A2, clk), mem(E.REG_A3, clk)
                ar = [mem(b + 4 * i, clk) for i in range(8)]
                br = [mem(c + 4 * i, clk) for i in range(64)]
                prod = as_int([x[0] for x in ar]) * as_int([x[0] for x in br])
                low = [mem(lo_reg[0] + 4 * i, clk + 1, (prod >> (32 * i)) & 0xffffffff) for i in range(64)]
                high = [mem(hi_reg[0] + 4 * i, clk + 1, (prod >> (2048 + 32 * i)) & 0xffffffff) for i in range(8)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("u256x2048_mul", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c),
                                   [(shard, clk, b, c, lo_reg[0], hi_reg[0], lo_reg, hi_reg, ar, br, low, high)], local))
                clk_extra += 1
            if code == E.SYS_BOOLEAN_CIRCUIT_GARBLE:
                # BooleanCircuitGarbleSyscall::execute (syscalls/precompiles/boolean_circuit/garble.rs:10-95): the gate count, delta and 17 words per gate
                # are read from $a0, 1 is written to $a1 when every gate checks, else 0; all at clk, no extra cycle. The event is filed as its rows
                assert b % 4 == 0 and c % 4 == 0
                count_rec = mem(b, clk)
                n_gates = count_rec[0]
                delta_recs = [mem(b + 4 + 4 * i, clk) for i in range(4)]
                delta = tuple(x[0] for x in delta_recs)
                no_read, no_write = (0, 0, 0, 0, 0), (0, 0, 0, 0, 0, 0)
                rows_ = [(shard, clk, b, c, 0, 0, n_gates, 0, delta, [count_rec] + delta_recs + [no_read] * 12, no_write)]
                running = True
                for g in range(n_gates):
                    recs = [mem(b + 20 + 68 * g + 4 * i, clk) for i in range(17)]
                    rows_.append([shard, clk, b + 20 + 68 * g, c, 1, g, n_gates, int(running), delta, recs, no_write])
                    running = running and E.garble_gate_ok([x[0] for x in recs], delta)
                rows_[-1][10] = mem(c, clk, int(running))
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("garble", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [tuple(x) for x in rows_], local))
            returned = None
            if is_linux:
                # the Linux syscalls (syscalls/precompiles/sys_linux/*.rs): brk reads register BRK, write reads $a2, mmap with a0 = 0 moves register
                # HEAP; every one writes $a3 (0, or 9 = EBADF) and returns a value for $v0; all at clk, no extra cycle. Filed under SYS_LINUX; the
                # SyscallPrecompile table learns of the call's code and result through the syscall event's a_record (prev_value, value)
                assert code != E.SYS_EXT_GROUP, "exit_group halts: not generated"
                no_read, no_write = (0, 0, 0, 0, 0), (0, 0, 0, 0, 0, 0)
                read_rec = mem(E.REG_BRK, clk) if code == E.SYS_BRK else mem(E.REG_A2, clk) if code == E.SYS_WRITE_LINUX else no_read
                v0, a3, new_heap = E.linux_syscall(code, b, c, brk=read_rec[0], heap=R.get(E.REG_HEAP, 0), a2=read_rec[0])
                a3_rec = mem(E.REG_A3, clk, a3)
                heap_rec = mem(E.REG_HEAP, clk, new_heap) if new_heap is not None else no_write
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append(("linux", (pc, next_pc, shard, clk, (v0, 0, 0, code, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, c, v0, code, read_rec, a3_rec, heap_rec)], local))
                returned = v0
            if w_decompress:
                # create_ec_decompress_event (events/precompiles/ec.rs:181-228): x is read at ptr + N, the y the sign bit ($a1) selects is written at
                # ptr, both at clk; no extra cycle
                n_words = E.WEIERSTRASS_CURVES[w_decompress]["n_limbs"] // 4
                assert b % 4 == 0 and c <= 1
                xr = [mem(b + 4 * n_words + 4 * i, clk) for i in range(n_words)]
                y = E.weierstrass_decompress(w_decompress, sum(rec_[0] << (32 * i) for i, rec_ in enumerate(xr)), c)
                yw = [mem(b + 4 * i, clk, (y >> (32 * i)) & 0xffffffff) for i in range(n_words)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                precompile.append((w_decompress + "_decompress", (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [(shard, clk, b, c, xr, yw)], local))
            if fp_call:
                # FpOpSyscall / Fp2AddSubSyscall / Fp2MulSyscall::execute (syscalls/precompiles/fptower/fp.rs:30-120, fp2_addsub.rs, fp2_mul.rs): x is
                # peeked, y read at clk, the result written over x at clk + 1; one extra cycle. The three Fp codes of a field file their events
                # together (under the field's FP_ADD), the two Fp2 add / sub codes likewise
                field, name = fp_call
                kind = "fp" if name.startswith("fp_") else "fp2_mul" if name == "fp2_mul" else "fp2_addsub"
                op = {"add": E.FIELD_OP_ADD, "sub": E.FIELD_OP_SUB, "mul": E.FIELD_OP_MUL}[name.split("_")[1]]
                per = E.WEIERSTRASS_CURVES[field]["n_limbs"] // 4
                W = per if kind == "fp" else 2 * per
                as_int = lambda ws: sum(w << (32 * i) for i, w in enumerate(ws))      # noqa: E731
                xw = [R.get(b + 4 * i, 0) for i in range(W)]
                yr = [mem(c + 4 * i, clk) for i in range(W)]
                yw = [x[0] for x in yr]
                if kind == "fp":
                    res = (E.fp_tower_result(field, kind, op, as_int(xw), as_int(yw)),)
                else:
                    res = E.fp_tower_result(field, kind, op, (as_int(xw[:per]), as_int(xw[per:])), (as_int(yw[:per]), as_int(yw[per:])))
                out = [(v >> (32 * i)) & 0xffffffff for v in res for i in range(per)]
                xwr = [mem(b + 4 * i, clk + 1, out[i]) for i in range(W)]
                local = [(addr, v[0], v[1]) for addr, v in sorted(touched.items())]
                event = (shard, clk, b, c, xwr, yr) if kind == "fp2_mul" else (shard, clk, b, c, op, xwr, yr)
                precompile.append((field + "_" + kind, (pc, next_pc, shard, clk, (0, 0, 0, 0, 0, 0), 0, [0, 0, 0], sid, b, c), [event], local))
                clk_extra += 1
            a = code if returned is None else returned     # only the Linux syscalls return a value; otherwise $v0 keeps the code
            a_rec = write(op_a, a, clk, POS_A)
            hi = code
            next_pc_after = 0 if code == E.SYS_HALT else next_pc
            rec.syscall.append((pc, next_pc_after, shard, clk, a_rec[2], 1, [0, 0, 0], sid, b, c))
            syscall_next = next_pc_after
            halted = code == E.SYS_HALT
            if code == E.SYS_COMMIT:
                committed[b] = c
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

    def arr(v, dt):
        return np.array(v, dtype=dt) if len(v) else np.zeros(0, dtype=dt)

    def pack(rec):
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
        out.alu = {chip: arr(v, E.ALU_EVENT) for chip, v in rec.alu.items()}
        out.mul, out.divrem = arr(rec.mul, E.COMP_ALU_EVENT), arr(rec.divrem, E.COMP_ALU_EVENT)
        out.branch, out.jump, out.mov_cond = arr(rec.branch, E.BRANCH_EVENT), arr(rec.jump, E.JUMP_EVENT), arr(rec.mov_cond, E.MOV_COND_EVENT)
        out.mem_instr, out.syscall, out.misc = arr(rec.mem_instr, E.MEM_INSTR_EVENT), arr(rec.syscall, E.SYSCALL_EVENT), arr(rec.misc, E.MISC_EVENT)
        out.memory_local = arr(rec.memory_local, MEMORY_LOCAL_EVENT)
        return out

    rec.memory_local = [(reg, first[reg], (last[reg][0], last[reg][1], R[reg])) for reg in sorted(first)] + rec.memory_local
    done_shards.append((rec, shard))
    final_digest = [committed.get(i, 0) for i in range(8)] if given is not None else (digest if halt else [0] * 8)
    shards = []
    for k, (r, sh) in enumerate(done_shards):
        o = pack(r)
        # the public values every shard of a batch carries are the run's final ones (executor.rs:2381-2400): the committed digest
        # is the same in all of them; start_pc / next_pc are the shard's own
        pv = {"start_pc": int(o.cpu["pc"][0]) if len(o.cpu) else pc_base, "next_pc": int(o.cpu["next_pc"][-1]) if len(o.cpu) else pc_base,
              "execution_shard": sh, "shard": sh, "exit_code": 0, "committed_value_digest": final_digest}
        shards.append(Shard("cpu", o, pv))
    if not machine:
        return Machine(prog, shards, pc_base)
    # ---- the deferred shards (prove.rs:283-400): precompile events first, then memory initialisation / finalisation
    last_pv = shards[-1].pv
    n_shard = shards[-1].pv["shard"]
    for kind, dt in (("poseidon2", E.POSEIDON2_PERMUTE_EVENT), ("keccak", E.KECCAK_SPONGE_BLOCK), ("sha_extend", E.SHA_EXTEND_EVENT),
                     ("sha_compress", E.SHA_COMPRESS_EVENT), ("ed_add", E.ED_ADD_EVENT),
                     ("ed_decompress", E.ED_DECOMPRESS_EVENT), ("uint256_mul", E.UINT256_MUL_EVENT), ("u256x2048_mul", E.U256X2048_MUL_EVENT), ("garble", E.GARBLE_ROW), ("linux", E.LINUX_EVENT)) + tuple(
            (curve + suffix, E.weierstrass_event_dtypes(curve)[k]) for curve in E.WEIERSTRASS_CURVES for k, suffix in ((0, "_add"), (1, "_double"))) + tuple(
            (curve + "_decompress", E.weierstrass_decompress_event_dtype(curve)) for curve in E.WEIERSTRASS_DECOMPRESS) + tuple(
            (field + "_" + kind, E.fp_tower_event_dtype(field, kind)) for field in E.FP_TOWER_CODES for kind in ("fp", "fp2_addsub", "fp2_mul")):
        mine = [e for e in precompile if e[0] == kind]
        if not mine:
            continue
        n_shard += 1
        o = Record()
        o.cpu = np.zeros(0, dtype=CPU_EVENT)
        o.precompile_syscall = arr([e[1] for e in mine], E.SYSCALL_EVENT)
        o.poseidon2_permute = arr([ev for e in mine for ev in e[2]] if kind == "poseidon2" else [], E.POSEIDON2_PERMUTE_EVENT)
        o.keccak_sponge = arr([ev for e in mine for ev in e[2]] if kind == "keccak" else [], E.KECCAK_SPONGE_BLOCK)
        o.sha_extend = arr([ev for e in mine for ev in e[2]] if kind == "sha_extend" else [], E.SHA_EXTEND_EVENT)
        o.sha_compress = arr([ev for e in mine for ev in e[2]] if kind == "sha_compress" else [], E.SHA_COMPRESS_EVENT)
        o.ed_add = arr([ev for e in mine for ev in e[2]] if kind == "ed_add" else [], E.ED_ADD_EVENT)
        o.ed_decompress = arr([ev for e in mine for ev in e[2]] if kind == "ed_decompress" else [], E.ED_DECOMPRESS_EVENT)
        o.uint256_mul = arr([ev for e in mine for ev in e[2]] if kind == "uint256_mul" else [], E.UINT256_MUL_EVENT)
        o.u256x2048_mul = arr([ev for e in mine for ev in e[2]] if kind == "u256x2048_mul" else [], E.U256X2048_MUL_EVENT)
        o.garble = arr([ev for e in mine for ev in e[2]] if kind == "garble" else [], E.GARBLE_ROW)
        o.linux = arr([ev for e in mine for ev in e[2]] if kind == "linux" else [], E.LINUX_EVENT)
        o.weierstrass = (kind, arr([ev for e in mine for ev in e[2]], dt)) if kind.endswith(("_add", "_double")) and kind.split("_")[0] in E.WEIERSTRASS_CURVES else None
        o.weierstrass_decompress = (kind.split("_")[0], arr([ev for e in mine for ev in e[2]], dt)) if kind.endswith("_decompress") and kind != "ed_decompress" else None
        o.fp_tower = (kind, arr([ev for e in mine for ev in e[2]], dt)) if kind.split("_")[0] in E.FP_TOWER_CODES and "_fp" in kind else None
        o.memory_local = arr([ev for e in mine for ev in e[3]], MEMORY_LOCAL_EVENT)
        pv = dict(last_pv, start_pc=last_pv["next_pc"], shard=n_shard)
        shards.append(Shard("precompile", o, pv))
    touched = sorted(a for a in born if last[a] != (0, 0))
    assert touched and touched[0] == 0, "register $0 is always touched (executor.rs:2561-2574 handles the other case separately)"
    init = arr([(a, born[a], 1, 1) for a in touched], E.MEMORY_INIT_FINALIZE_EVENT)
    fin = arr([(a, R[a], last[a][0], last[a][1]) for a in touched], E.MEMORY_INIT_FINALIZE_EVENT)
    prev_addr = 0
    for k in range(0, len(touched), memory_chunk):
        n_shard += 1
        o = Record()
        o.cpu = np.zeros(0, dtype=CPU_EVENT)
        o.memory_init, o.memory_finalize = init[k:k + memory_chunk], fin[k:k + memory_chunk]
        last_addr = int(o.memory_init["addr"][-1])
        pv = dict(last_pv, start_pc=last_pv["next_pc"], shard=n_shard, previous_init_addr=prev_addr, last_init_addr=last_addr,
                  previous_finalize_addr=prev_addr, last_finalize_addr=last_addr)
        prev_addr = last_addr
        shards.append(Shard("memory", o, pv))
    for sh in shards:      # shards before the memory ones: previous = last = 0 (verify.rs:207-245)
        for key in ("previous_init_addr", "last_init_addr", "previous_finalize_addr", "last_finalize_addr"):
            sh.pv.setdefault(key, 0)
    return Machine(prog, shards, pc_base)


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
    for base, key in ((45, "previous_init_addr"), (77, "last_init_addr"), (109, "previous_finalize_addr"), (141, "last_finalize_addr")):
        a = pv.get(key, 0)
        v[base:base + 32] = [(a >> k) & 1 for k in range(32)]
    return F.to_monty(v)
