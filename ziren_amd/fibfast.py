"""The fibonacci guest's shards without stepping through the executor: BASELINE.json's workload ("fibonacci 2^22-row trace") at full size.

`miniexec` executes one instruction per Python iteration (~60 us per cycle: two minutes for the 2^21 cycles of a full shard —
`MAX_SHARD_SIZE`, crates/stark/src/opts.rs:6). The guest's loop (examples/fibonacci/guest/src/main.rs:12-40, hand-assembled below) is six
instructions with a fixed register access pattern, so every event of a shard that lies inside the loop is a closed-form function of the
cycle number and of the Fibonacci values; this module writes those functions down with numpy and produces the same `Record` arrays
(`cpu`, `alu[AddSub]`, `divrem`, `branch`, `memory_local`) and public values that `miniexec.run_machine(program=fibonacci_program(n),
shard_cycles=S)` produces for that shard — event for event, byte for byte: tests/test_fibfast.py compares the two on every loop shard
of several runs. Shards that hold the prologue or the epilogue (the first and the last ones) are left to the executor.

Reference for what a shard's record holds: crates/core/executor/src/executor.rs:2186-2200 (bump_record), :2352-2356 (no shard ends
between a branch and its delay slot), events/cpu.rs:46-77, events/instr.rs, events/memory.rs:226-237; the reference's own sweep of this
guest is crates/prover/scripts/fibonacci_sweep.rs:41-76."""
import numpy as np

from . import abi
from . import events as E
from . import miniexec as M

T0, T1, T2, T3, T4, T5 = 8, 9, 10, 11, 12, 13
MODULUS = 7919
PROLOGUE, LOOP = 5, 6          # instructions before the loop, instructions per iteration
LOOP_INDEX = 5                 # the loop's first instruction


def fibonacci_program(n):
    """examples/fibonacci/guest/src/main.rs:12-40 by hand: a, b = 0, 1; n times (c = (a + b) % 7919; a = b; b = c); the words n, a, b are
    committed (digest words 0..2, the other five words zero), then HALT. The loop branches backwards; `b = c` sits in the delay slot."""
    p = [(E.ADD, T0, 0, n, 0, 1), (E.ADD, T1, 0, 0, 0, 1), (E.ADD, T2, 0, 1, 0, 1), (E.ADD, T4, 0, 0, 0, 1), (E.ADD, T5, 0, MODULUS, 0, 1)]
    loop = len(p)
    p += [(E.ADD, T3, T1, T2, 0, 0), (E.MODU, T3, T3, T5, 0, 0), (E.ADD, T1, T2, 0, 0, 1), (E.ADD, T4, T4, 1, 0, 1)]
    branch = len(p)
    p += [(E.BNE, T4, T0, (4 * loop - 4 * (branch + 1)) & 0xffffffff, 0, 1),      # target = next_pc + offset
          (E.ADD, T2, T3, 0, 0, 1)]                                                 # delay slot: b = c
    for idx, reg in enumerate([T0, T1, T2, 0, 0, 0, 0, 0]):
        p += [(E.ADD, E.REG_V0, 0, E.SYS_COMMIT, 0, 1), (E.ADD, E.REG_A0, 0, idx, 0, 1), (E.ADD, E.REG_A1, reg, 0, 0, 1), (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
    p += [(E.ADD, E.REG_V0, 0, E.SYS_HALT, 0, 1), (E.ADD, E.REG_A0, 0, 0, 0, 1), (E.SYSCALL, E.REG_V0, E.REG_A0, E.REG_A1, 0, 0)]
    return p


def fib(n):
    a, b = 0, 1
    for _ in range(n):
        a, b = b, (a + b) % MODULUS
    return a, b


def program_array(n, pc_base=0x1000):
    """The INSTRUCTION array miniexec packs for the same program."""
    p = fibonacci_program(n)
    prog = np.zeros(len(p), dtype=M.INSTRUCTION)
    for i, (op, op_a, op_b, op_c, imm_b, imm_c) in enumerate(p):
        prog[i] = (op, op_a, [0, 0], op_b, op_c, imm_b, imm_c, [0, 0], 1, [0, 0, 0], 0)
    return prog


def shard_starts(n, shard_cycles):
    """First global cycle of every CPU shard of the run (the last entry: the cycle count). A shard is closed at the first cycle at least
    `shard_cycles` after its start that is not a delay slot (executor.rs:2352-2356)."""
    total = PROLOGUE + LOOP * n + 35         # eight COMMITs and the HALT: 35 cycles
    starts = [0]
    while starts[-1] + shard_cycles < total:
        t = starts[-1] + shard_cycles
        if PROLOGUE <= t < PROLOGUE + LOOP * n and (t - PROLOGUE) % LOOP == 5:
            t += 1
        starts.append(t)
    return starts + [total]


# (register, MemoryAccessPosition, cycles back to the register's previous access, that access's position) for every access of a slot, in
# the order the executor makes them (c, b, a). Derived from the six instructions: every register of the loop is touched once per iteration
# at least, so the previous access is never more than six cycles back.
_ACCESSES = {
    0: [("c", T2, 1, 1, 3), ("b", T1, 2, 4, 3), ("a", T3, 3, 1, 2)],        # ADD  T3 <- T1 + T2
    1: [("c", T5, 1, 6, 1), ("b", T3, 2, 1, 3), ("a", T3, 3, 0, 2)],        # MODU T3 <- T3 % T5
    2: [("b", T2, 2, 2, 1), ("a", T1, 3, 2, 2)],                            # ADD  T1 <- T2 + 0
    3: [("b", T4, 2, 5, 3), ("a", T4, 3, 0, 2)],                            # ADD  T4 <- T4 + 1
    4: [("b", T0, 2, 6, 2), ("a", T4, 3, 1, 3)],                            # BNE  T4, T0   (a is read)
    5: [("b", T3, 2, 4, 3), ("a", T2, 3, 3, 2)],                            # ADD  T2 <- T3 + 0   (delay slot)
}


def fib_shard(n, shard_cycles, shard_no, pc_base=0x1000):
    """The CPU shard number `shard_no` (1-based) of the run of fibonacci_program(n) cut into shards of `shard_cycles` cycles, which must lie
    entirely inside the loop with the branch taken in every iteration it holds. Returns a miniexec.Machine with that one shard."""
    starts = shard_starts(n, shard_cycles)
    g0, g1 = starts[shard_no - 1], starts[shard_no]
    if not (g0 >= PROLOGUE + 2 * LOOP and g1 <= PROLOGUE + LOOP * (n - 1)):
        raise ValueError("the shard holds part of the prologue, the first iterations or the loop's exit: use miniexec for it")
    starts_a = np.array(starts, dtype=np.int64)
    G = np.arange(g0, g1, dtype=np.int64)
    it, sl = (G - PROLOGUE) // LOOP, (G - PROLOGUE) % LOOP
    # Fibonacci state per iteration: a_i = T1, b_i = T2 on entry, sum_i = a_i + b_i, c_i = sum_i % 7919; T3 on entry = c_(i-1)
    i_lo, i_hi = int(it[0]) - 1, int(it[-1]) + 1
    av, bv = np.zeros(i_hi - i_lo + 1, dtype=np.int64), np.zeros(i_hi - i_lo + 1, dtype=np.int64)
    a, b = fib(i_lo)
    for k in range(i_hi - i_lo + 1):
        av[k], bv[k] = a, b
        a, b = b, (a + b) % MODULUS
    k = it - i_lo
    A, B = av[k], bv[k]
    S_ = A + B
    C = S_ % MODULUS
    Cprev = bv[k]                 # T3 on entry to iteration i = c_(i-1) = b_i
    ev_a = np.select([sl == 0, sl == 1, sl == 2, sl == 3, sl == 4, sl == 5], [S_, C, B, it + 1, it + 1, C])
    ev_b = np.select([sl == 0, sl == 1, sl == 2, sl == 3, sl == 4, sl == 5], [A, S_, B, it, np.full_like(it, n), C])
    off = (4 * LOOP_INDEX - 4 * (LOOP_INDEX + 4 + 1)) & 0xffffffff
    ev_c = np.select([sl == 0, sl == 1, sl == 2, sl == 3, sl == 4, sl == 5], [B, np.full_like(it, MODULUS), 0 * it, 0 * it + 1, 0 * it + off, 0 * it])
    # the value a written register held before (prev_value of the write record)
    wr_prev = np.select([sl == 0, sl == 1, sl == 2, sl == 3, sl == 5], [Cprev, S_, A, it, B], default=0)
    P = lambda idx: pc_base + 4 * idx          # noqa: E731
    pc = P(LOOP_INDEX + sl)
    next_pc = np.where(sl == 5, P(LOOP_INDEX), pc + 4)
    next_next_pc = np.where(sl == 4, P(LOOP_INDEX), next_pc + 4)      # the branch is taken: target = the loop's first instruction

    def shard_of(g):
        return np.searchsorted(starts_a, g, side="right")             # 1-based

    sh = shard_of(G)
    assert (sh == shard_no).all()
    clk = 5 * (G - starts_a[sh - 1])

    cpu = np.zeros(len(G), dtype=M.CPU_EVENT)
    cpu["clk"], cpu["pc"], cpu["next_pc"], cpu["next_next_pc"] = clk, pc, next_pc, next_next_pc
    cpu["a"], cpu["b"], cpu["c"] = ev_a, ev_b, ev_c
    for f in ("a_record", "b_record", "c_record", "hi_record", "memory_record"):
        cpu[f]["tag"] = M.TAG_NONE
    cpu["hi"]["tag"] = 1
    # flat list of the shard's register accesses, in time order, for the MemoryLocal events
    acc = {key: [] for key in ("reg", "time", "before", "after", "shard", "ts", "pshard", "pts")}
    for s, accesses in _ACCESSES.items():
        m = sl == s
        if not m.any():
            continue
        g = G[m]
        for which, reg, pos, back, ppos in accesses:
            gp = g - back
            psh = shard_of(gp)
            pts = 5 * (gp - starts_a[psh - 1]) + ppos
            ts = clk[m] + pos
            is_write = which == "a" and s != 4
            value = {"a": ev_a, "b": ev_b, "c": ev_c}[which][m]
            r = cpu[which + "_record"]
            if is_write:
                r["tag"][m] = M.TAG_WRITE
                w = np.zeros(len(g), dtype=E.MEMORY_WRITE_RECORD)
                w["value"], w["shard"], w["timestamp"], w["prev_value"], w["prev_shard"], w["prev_timestamp"] = value, shard_no, ts, wr_prev[m], psh, pts
                r["write"][m] = w
                before = wr_prev[m]
            else:
                r["tag"][m] = M.TAG_READ
                rd = np.zeros(len(g), dtype=M.MEMORY_READ_RECORD)
                rd["value"], rd["shard"], rd["timestamp"], rd["prev_shard"], rd["prev_timestamp"] = value, shard_no, ts, psh, pts
                r["read"][m] = rd
                before = value
            acc["reg"].append(np.full(len(g), reg)); acc["time"].append(4 * g + pos); acc["before"].append(before); acc["after"].append(value)
            acc["shard"].append(np.full(len(g), shard_no)); acc["ts"].append(ts); acc["pshard"].append(psh); acc["pts"].append(pts)
    acc = {key: np.concatenate(v) for key, v in acc.items()}
    local = []
    for reg in (T0, T1, T2, T3, T4, T5):
        m = acc["reg"] == reg
        if not m.any():
            continue
        t = acc["time"][m]
        f, l = np.argmin(t), np.argmax(t)
        pick = lambda key, j: int(acc[key][m][j])      # noqa: E731
        local.append((reg, (pick("pshard", f), pick("pts", f), pick("before", f)), (shard_no, pick("ts", l), pick("after", l))))
    rec = M.Record()
    rec.cpu = cpu
    adds = sl != 1
    adds &= sl != 4
    alu = np.zeros(int(adds.sum()), dtype=E.ALU_EVENT)
    alu["pc"], alu["next_pc"], alu["opcode"], alu["a"], alu["b"], alu["c"] = pc[adds], next_pc[adds], E.ADD, ev_a[adds], ev_b[adds], ev_c[adds]
    rec.alu = {chip: (alu if chip == E.CHIP_ADD_SUB else np.zeros(0, dtype=E.ALU_EVENT)) for chip in E.CHIP_NAMES}
    d = sl == 1
    div = np.zeros(int(d.sum()), dtype=E.COMP_ALU_EVENT)
    div["shard"], div["clk"], div["pc"], div["next_pc"], div["opcode"] = shard_no, clk[d], pc[d], next_pc[d], E.MODU
    div["a"], div["b"], div["c"] = ev_a[d], ev_b[d], ev_c[d]
    rec.divrem = div
    br = sl == 4
    bev = np.zeros(int(br.sum()), dtype=E.BRANCH_EVENT)
    bev["pc"], bev["next_pc"], bev["next_next_pc"], bev["opcode"], bev["a"], bev["b"], bev["c"] = pc[br], next_pc[br], next_next_pc[br], E.BNE, ev_a[br], ev_b[br], ev_c[br]
    rec.branch = bev
    rec.mul = np.zeros(0, dtype=E.COMP_ALU_EVENT)
    rec.jump, rec.mov_cond = np.zeros(0, dtype=E.JUMP_EVENT), np.zeros(0, dtype=E.MOV_COND_EVENT)
    rec.mem_instr, rec.syscall, rec.misc = np.zeros(0, dtype=E.MEM_INSTR_EVENT), np.zeros(0, dtype=E.SYSCALL_EVENT), np.zeros(0, dtype=E.MISC_EVENT)
    rec.memory_local = np.array(local, dtype=M.MEMORY_LOCAL_EVENT)
    fa, fb = fib(n)
    pv = {"start_pc": int(pc[0]), "next_pc": int(next_pc[-1]), "execution_shard": shard_no, "shard": shard_no, "exit_code": 0,
          "committed_value_digest": [n, fa, fb, 0, 0, 0, 0, 0], "previous_init_addr": 0, "last_init_addr": 0, "previous_finalize_addr": 0,
          "last_finalize_addr": 0}
    return M.Machine(program_array(n, pc_base), [M.Shard("cpu", rec, pv)], pc_base)


def loop_event_estimate(cycles):
    """estimate_mips_event_counts (crates/core/executor/src/cost.rs:96-195) after `cycles` cycles of the loop: opcode counts (four ADDs, a MODU
    — counted under DivRem — and a BNE per six cycles), Mul and Lt raised by the DivRem count (:189-190), the six registers as touched
    addresses (MemoryLocal rows of four, two Global messages each); the other dependency events are not estimated (:192-193)."""
    it = cycles // LOOP
    return {"Cpu": cycles, "AddSub": 4 * it, "DivRem": it, "Mul": it, "Lt": it, "Branch": it, "MemoryLocal": 2, "Global": 12}


def shaped_shard(shard_size=1 << 21, pc_base=0x1000):
    """A middle shard of a long run as the reference's executor cuts it when shapes are on (the default: crates/prover/src/lib.rs:210-213,
    crates/core/machine/src/utils/prove.rs:146-148): every 16 cycles it checks that some maximal shape of the configured shard size still
    holds the estimated event counts with a margin, and closes the shard when none does (executor.rs:2429-2516) — for this loop after
    about three quarters of `shard_size` cycles, when the DivRem / Mul count reaches the tallest height any maximal shape allows.
    Returns (machine with that one shard, cycles, "shape" | "clock")."""
    from . import shape as SH
    cycles, why = SH.executor_shard_cycles(shard_size, loop_event_estimate)
    n = (3 * cycles) // LOOP + 8
    m = fib_shard(n, cycles, 2, pc_base)
    assert cycles <= len(m.shards[0].record.cpu) <= cycles + 1
    return m, cycles, why


def full_shard(log_cycles, pc_base=0x1000):
    """The second shard of a run long enough to fill it: 2^log_cycles cycles of the loop (Cpu 2^k rows, AddSub 2^k, DivRem and Branch
    2^(k-2) after padding): the shape every middle shard of a long fibonacci run has."""
    if 5 << log_cycles >= 1 << 24:
        # the shard's clock advances 5 per cycle (executor.rs:1682) and the Cpu chip range-checks it to 24 bits (cpu/air/mod.rs:39,131-138):
        # no shard of the reference holds 2^22 cycles (it closes one at 0.8 SHARD_SIZE cycles at the latest, executor.rs:325,2423), and a
        # trace that did would not satisfy the AIR — the verifier rejects such a proof (found by verifying round 3's "FIB-22" shard)
        raise ValueError(f"a shard of 2^{log_cycles} cycles overflows the 24-bit shard clock")
    S = (1 << log_cycles) - 1       # a shard that would end on a delay slot runs one cycle longer: 2^k - 1 or 2^k cycles, never 2^k + 1
    n = (3 * S) // LOOP + 8
    m = fib_shard(n, S, 2, pc_base)
    assert (1 << (log_cycles - 1)) < len(m.shards[0].record.cpu) <= (1 << log_cycles)
    return m


# ---- the shard on the device ------------------------------------------------------------------------------------------------------------

def _log2_rows(n):
    h = 16
    while h < n:
        h <<= 1
    return h.bit_length() - 1


# MipsAirId name -> (work-list key, chips.py recorder name or ALU chip id); the order is the machine's (tests/machine_lib.py::build_shard)
_ALU = {"AddSub": E.CHIP_ADD_SUB, "Bitwise": E.CHIP_BITWISE, "Lt": E.CHIP_LT, "ShiftLeft": E.CHIP_SHIFT_LEFT, "ShiftRight": E.CHIP_SHIFT_RIGHT,
        "CloClz": E.CHIP_CLO_CLZ}


def core_heights(rec, glob):
    """MipsAir::core_heights (crates/core/machine/src/mips/mod.rs:456-487) of a record with its dependency events added: rows per chip."""
    h = {"Cpu": len(rec.cpu), "Branch": len(rec.branch), "Jump": len(rec.jump), "MovCond": len(rec.mov_cond), "MiscInstrs": len(rec.misc),
         "MemoryInstrs": len(rec.mem_instr), "SyscallInstrs": len(rec.syscall), "DivRem": len(rec.divrem), "Mul": len(rec.mul),
         "MemoryLocal": -(-len(rec.memory_local) // M.MEMORY_LOCAL_ENTRIES_PER_ROW), "Global": 2 * len(rec.memory_local) + 2 * len(rec.syscall),
         "SyscallCore": len(rec.syscall)}
    for name, chip in _ALU.items():
        h[name] = len(rec.alu[chip])
    return h


class DeviceShard:
    """The chips a CPU shard of this guest includes, recorded once, and their traces generated on the device from the shard's events as
    often as asked: the step before `commit`.

    shape=None: the tight shard — a chip is in when it has events (MachineAir::included), every trace padded to the next power of two
    (at least 16 rows): Cpu, AddSub, Lt, Mul (the executor's dependency events of the branches and of the division:
    crates/core/executor/src/dependencies.rs), Branch, DivRem, MemoryLocal, Global, Byte, Program. Same chips, heights and traces as
    tests/machine_lib.py::build_shard gives for the shard (tests/test_fibfast.py holds the two against each other).

    shape="fix": the shard as the reference proves it by default — `CoreShapeConfig::fix_shape` (ziren_amd/shape.py) picks the allowed shape
    of least area that covers the record; every chip the shape names is in, events or not, padded to the shape's height
    (`fixed_log2_rows`), and Program takes its allowed height (2^19 at least). A dict {chip: log2 height} fixes the shape directly."""

    def __init__(self, machine, k=0, shape=None):
        from . import chips, shape as SH
        sh = machine.shards[k]
        assert sh.kind == "cpu"
        self.machine, self.shard = machine, sh
        self.rec = rec = M.add_dependencies(sh.record)
        self.shard_no = sh.pv["shard"]
        self.glob = M.global_lookup_events(rec.memory_local)
        assert not len(rec.jump) and not len(rec.mov_cond) and not len(rec.mem_instr) and not len(rec.syscall) and not len(rec.misc)
        self.heights = core_heights(rec, self.glob)
        self.shape_key = None
        if isinstance(shape, str):
            assert shape == "fix"
            shape, key, cluster = SH.fix_shape(self.heights)
            self.shape_key = (key, cluster)
            plen = len(machine.program)
            shape = dict(shape, Byte=16, Program=next(h for h in SH.PREPROCESSED_ALLOWED["Program"] if plen <= (1 << h)))
        self.shape = shape
        events = {"Cpu": rec.cpu, "Branch": rec.branch, "Jump": rec.jump, "MovCond": rec.mov_cond, "MiscInstrs": rec.misc, "MemoryInstrs": rec.mem_instr,
                  "SyscallInstrs": rec.syscall, "SyscallCore": rec.syscall, "DivRem": rec.divrem, "Mul": rec.mul, "MemoryLocal": rec.memory_local,
                  "Global": self.glob}
        events.update({name: rec.alu[chip] for name, chip in _ALU.items()})
        recorders = {"Cpu": chips.record_cpu_chip, "Branch": chips.record_branch_chip, "Jump": chips.record_jump_chip, "MovCond": chips.record_mov_cond_chip,
                     "MiscInstrs": chips.record_misc_instrs_chip, "MemoryInstrs": chips.record_memory_instrs_chip,
                     "SyscallInstrs": chips.record_syscall_instrs_chip, "SyscallCore": lambda lh: chips.record_syscall_table_chip(False, lh),
                     "DivRem": chips.record_divrem_chip, "Mul": chips.record_mul_chip, "MemoryLocal": chips.record_memory_local_chip,
                     "Global": chips.record_global_chip}
        recorders.update({name: (lambda lh, chip=chip: chips.record_chip(chip, lh)) for name, chip in _ALU.items()})
        order = ["Cpu"] + [E.CHIP_NAMES[c] for c in sorted(E.CHIP_NAMES)] + ["SyscallInstrs", "Jump", "MovCond", "Branch", "MemoryInstrs", "MiscInstrs", "Mul",
                                                                               "DivRem", "SyscallCore", "MemoryLocal", "Global"]
        self.work = []
        for name in order:
            rows = self.heights[name]
            if shape is None:
                if not rows:
                    continue
                lh = _log2_rows(rows)
            else:
                if name not in shape:
                    assert not rows, f"the shape leaves {name} out but the record has {rows} rows for it"
                    continue
                lh = shape[name]
                assert rows <= (1 << lh), (name, rows, lh)
            self.work.append((name, events[name], lh, recorders[name]))
        self.plh = _log2_rows(len(machine.program)) if shape is None else shape["Program"]
        self.chips = [record(lh) for _, _, lh, record in self.work]
        self.chips += [chips.record_byte_chip(prep_index=0), chips.record_program_chip(self.plh, prep_index=1)]
        self.public_values = M.public_values(sh.pv)

    def pin(self, ctx):
        """Move the shard's event vectors into page-locked host memory (zkm_host_alloc): where a host that feeds a GPU keeps them, the
        upload inside every zkm_tracegen_* call is then plain DMA at PCIe rate. Returns the bytes pinned."""
        if getattr(self, "_pinned_by", None) is ctx:
            return 0                                  # already page-locked through this context
        total = 0
        pinned = []
        for name, ev, lh, record in self.work:
            words = max(len(ev), 1) * (ev.dtype.itemsize // 4)
            buf = ctx.host_alloc((words,))
            buf[:len(ev) * (ev.dtype.itemsize // 4)] = ev.view(np.uint32).reshape(-1)
            pinned.append((name, buf[:len(ev) * (ev.dtype.itemsize // 4)].view(ev.dtype), lh, record))
            total += ev.nbytes
        self.work = pinned
        self._pinned_by = ctx
        return total

    def event_bytes(self):
        """Bytes of the executor's events this shard's traces are generated from (what crosses PCIe per shard)."""
        return int(sum(ev.nbytes for name, ev, _, _ in self.work if name != "SyscallCore"))

    def preprocessed(self, ctx):
        return [ctx.tracegen_byte_table(), ctx.tracegen_program(self.machine.program, self.machine.pc_base, self.plh)]

    def prefetch(self, ctx):
        """Queue the copy of this shard's event vectors to the device (zkm_events_upload_async, the DMA stream) and return at once:
        {chip: DeviceEvents} for `traces`. Called right before the previous shard's proof, the transfer runs under that proof. The vectors
        should be page-locked (`pin`). SyscallCore reads SyscallInstrs' vector: one copy serves both."""
        return {name: ctx.events_upload_async(ev) for name, ev, _, _ in self.work if len(ev) and name != "SyscallCore"}

    _KIND = {"Cpu": abi.TG_CPU, "Branch": abi.TG_BRANCH, "Jump": abi.TG_JUMP, "MovCond": abi.TG_MOV_COND, "Mul": abi.TG_MUL, "DivRem": abi.TG_DIVREM,
             "MemoryInstrs": abi.TG_MEMORY_INSTRS, "MiscInstrs": abi.TG_MISC_INSTRS, "SyscallInstrs": abi.TG_SYSCALL_INSTRS,
             "SyscallCore": abi.TG_SYSCALL_CORE, "MemoryLocal": abi.TG_MEMORY_LOCAL, "Global": abi.TG_GLOBAL}

    def traces(self, ctx, prefetched=None, one_call=True):
        """generate_traces on the device: one DeviceMatrix per chip of `self.chips`, in that order. `prefetched` = what `prefetch` returned
        (freed here); without it every generator uploads its own events. one_call (default): the whole shard in one zkm_tracegen_shard
        call — every generator queued behind the copy of its events, one synchronisation; False: one entry point per chip, as the
        chips' own tests call them (same traces: tests/test_fibfast.py)."""
        pre = prefetched or {}
        from . import lib as _lib
        if one_call and hasattr(_lib.load(), "zkm_tracegen_shard"):      # (an older build under A/B comparison, ZKM_HIP_LIB, has the per-chip calls only)
            items = []
            for name, ev, lh, _ in self.work:
                ev = pre.get("SyscallInstrs" if name == "SyscallCore" else name, ev)
                if name in _ALU:
                    items.append((abi.TG_ALU, ev, lh, {"chip": _ALU[name]}))
                elif name == "Cpu":
                    items.append((abi.TG_CPU, ev, lh, {"program": self.machine.program, "pc_base": self.machine.pc_base, "shard": self.shard_no}))
                else:
                    items.append((self._KIND[name], ev, lh, None))
            items.append((abi.TG_BYTE_MULTS, None, 16, None))
            items.append((abi.TG_PROGRAM_MULTS, None, self.plh, None))
            born = ctx.tracegen_shard(items)
            for d in pre.values():
                d.free()
            return born
        blu = ctx.byte_lookups()
        born, program_mults = [], None
        for name, ev, lh, _ in self.work:
            ev = pre.get("SyscallInstrs" if name == "SyscallCore" else name, ev)
            if name == "Cpu":
                cpu, program_mults = ctx.tracegen_cpu_and_program(ev, self.machine.program, self.machine.pc_base, self.shard_no, lh, self.plh, blu)
                born.append(cpu)
            elif name in _ALU:
                born.append(ctx.tracegen_alu(_ALU[name], ev, lh, blu))
            elif name == "Branch":
                born.append(ctx.tracegen_branch(ev, lh, blu))
            elif name == "Mul":
                born.append(ctx.tracegen_mul(ev, lh, blu))
            elif name == "DivRem":
                born.append(ctx.tracegen_divrem(ev, lh, blu))
            elif name == "MemoryLocal":
                born.append(ctx.tracegen_memory_local(ev, lh))
            elif name == "Global":
                born.append(ctx.tracegen_global(ev, lh, blu))
            elif name == "Jump":
                born.append(ctx.tracegen_jump(ev, lh))
            elif name == "MovCond":
                born.append(ctx.tracegen_mov_cond(ev, lh))
            elif name == "MemoryInstrs":
                born.append(ctx.tracegen_memory_instrs(ev, lh, blu))
            elif name == "MiscInstrs":
                born.append(ctx.tracegen_misc_instrs(ev, lh, blu))
            elif name == "SyscallInstrs":
                born.append(ctx.tracegen_syscall_instrs(ev, lh))
            elif name == "SyscallCore":
                born.append(ctx.tracegen_syscall(ev, False, lh, blu))
            else:
                raise KeyError(name)
        born.append(ctx.tracegen_byte_mults(blu))
        born.append(program_mults)
        blu.free()
        for d in pre.values():
            d.free()
        return born

    def committed_cells(self):
        """Cells the proof commits to: per chip rows x (preprocessed + main + permutation + quotient columns)."""
        return sum((1 << c.log_height) * (c.prep_width + c.main_width + 4 * c.perm_ext_width + (4 << c.log_quotient_degree)) for c in self.chips)
