"""AIR completeness sweep. The recorded AIRs (ziren_amd/chips.py) are hand transcriptions of the
reference's `eval` functions; "every constraint vanishes on honest rows" cannot see a constraint that was *dropped*. This sweep does:
for every recorded core chip and every main column, one cell of a real row is changed, and the change must be noticed — by a constraint
(air.debug_constraints) or by the chip's lookups (its signed multiset of sent / received values changes, so the shard's LogUp sum cannot
stay zero). A column nothing notices is either genuinely unconstrained in the reference (listed in FREE with the reason) or a hole in
the transcription. Rows: the first and last real rows (some columns only bind at a boundary) and two dozen spread over the trace (union
layouts and per-opcode columns only bind on rows of their opcode)."""
import numpy as np
import pytest

from ziren_amd import air, chips, events as E, field as F, miniexec as M

import machine_lib as ML
from test_chip_airs import lookup_tally

# (chip name, column) -> why nothing in the reference's eval reads it on any row
FREE = {
    ("SyscallInstrs", 7): "is_prev_a1_zero.inverse: IsZeroOperation::eval (operations/is_zero.rs:42-58) only binds the inverse when its input is "
                          "non-zero (1 - inverse * a = result); byte 1 of the syscall code is zero for every non-Linux syscall",
    ("MemoryGlobalInit", 0): "shard: MemoryGlobalChip::eval (memory/global.rs:294-312) sends (0, 0, addr, value) for Initialize and never reads local.shard",
    ("MemoryGlobalInit", 107): "is_prev_addr_zero.inverse: evaluated under is_first_row only (global.rs:383), and free there when the previous address is zero",
    ("MemoryGlobalFinalize", 107): "is_prev_addr_zero.inverse: as for MemoryGlobalInit",
}


def recorders():
    def memory_local():
        r = chips._Rec(M.MEMORY_LOCAL_WIDTH)
        chips._memory_local(r)
        return r
    out = {E.CHIP_NAMES[c]: (lambda c=c: chips.record_constraints(c)) for c in E.CHIP_NAMES}
    out.update({"Cpu": chips.record_cpu_constraints, "Jump": chips.record_jump_constraints, "MovCond": chips.record_mov_cond_constraints,
                "Branch": chips.record_branch_constraints, "Mul": chips.record_mul_constraints, "DivRem": chips.record_divrem_constraints,
                "MemoryInstrs": chips.record_memory_instrs_constraints, "MiscInstrs": chips.record_misc_instrs_constraints,
                "SyscallInstrs": chips.record_syscall_instrs_constraints, "MemoryLocal": memory_local, "Global": chips.record_global_constraints,
                "MemoryGlobalInit": lambda: chips.record_memory_global_constraints(False),
                "MemoryGlobalFinalize": lambda: chips.record_memory_global_constraints(True),
                "SyscallCore": lambda: chips.record_syscall_table_constraints(False), "SyscallPrecompile": lambda: chips.record_syscall_table_constraints(True),
                "Poseidon2Permute": chips.record_poseidon2_permute_constraints})
    return out


def real_rows(name, t):
    """Indices of rows that carry an event (every chip has an is_real-like column or non-zero content)."""
    nz = np.nonzero(t.any(axis=1))[0]
    if name == "Cpu":
        nz = np.nonzero(t[:, 65])[0]
    elif name == "Global":
        nz = np.nonzero(t[:, 63])[0]
    elif name == "Poseidon2Permute":
        nz = np.nonzero(t[:, 972])[0]
    return nz


def sweep(oracle, m):
    recs = recorders()
    seen, holes = set(), []
    for k, sh in enumerate(m.shards):
        cs = ML.build_shard(ML.Oracle(oracle), m, k)
        pv = F.from_monty(ML.shard_public_values(sh)).astype(np.uint64)
        for c in cs[:-2]:
            if c.name in seen or c.name not in recs:
                continue
            seen.add(c.name)
            rec = recs[c.name]()
            t = F.from_monty(c.trace)
            assert air.debug_constraints(rec.b, t, public_values=pv) == [], c.name
            base = lookup_tally([c])
            rows = real_rows(c.name, t)
            pick = sorted({int(rows[0]), int(rows[-1])} | {int(rows[i]) for i in np.linspace(0, len(rows) - 1, 24).astype(int)})
            for col in range(t.shape[1]):
                caught = False
                for row in pick:
                    bad = t.copy()
                    bad[row, col] = (int(bad[row, col]) + 1) % F.P
                    if air.debug_constraints(rec.b, bad, public_values=pv):
                        caught = True
                        break
                    c2 = chips.RecordedChip(name=c.name, log_height=c.log_height, main_width=c.main_width, sends=c.sends, receives=c.receives)
                    c2.trace, c2.prep_trace = F.to_monty(bad), c.prep_trace
                    if lookup_tally([c2]) != base:
                        caught = True
                        break
                if not caught and (c.name, col) not in FREE:
                    holes.append((c.name, col))
    return seen, holes


def windowed_sweep(rec, chip, t, rows, chunk=4096):
    """The same question for a chip whose rows come many to an event (KeccakSponge, ShaExtend, ShaCompress): every column, changed on one of
    `rows`, must be noticed by a constraint or by the chip's lookups. The constraints only look at a row and its successor, so all variants
    of a row are evaluated in one pass over a stack of three-row windows (`rows` must not contain the first or the last row of the table:
    the boundary selectors are not reproduced in the stack). Returns the columns nothing noticed."""
    t = t.astype(np.uint64)
    w = t.shape[1]
    caught = np.zeros(w, dtype=bool)
    for r in rows:
        todo = np.nonzero(~caught)[0]
        if not len(todo):
            break
        for lo in range(0, len(todo), chunk):
            cols = todo[lo:lo + chunk]
            stack = np.concatenate([t[r + 1:r + 2], np.tile(t[r - 1:r + 2], (len(cols), 1))])
            at = 2 + 3 * np.arange(len(cols))
            stack[at, cols] = (stack[at, cols] + 1) % F.P
            bad = air.violated_rows(rec.b, stack)
            caught[cols] |= bad[at - 1] | bad[at]          # the changed row as `next`, then as `local`
        # the lookups of the row: for every column nothing has noticed yet, the exact difference between the signed multiset the changed row
        # sends / receives and the honest row's (all candidates in one pass per lookup: row 0 of the stack is the honest row)
        cand = np.nonzero(~caught)[0]
        if not len(cand):
            continue
        n = len(cand) + 1
        stack = np.tile(t[r:r + 1], (n, 1))
        stack[1 + np.arange(len(cand)), cand] = (stack[1 + np.arange(len(cand)), cand] + 1) % F.P
        main = {c: stack[:, c] for c in range(w)}
        delta = [dict() for _ in range(n)]
        for sign, lks in ((1, chip.sends), (-1, chip.receives)):
            for lk in lks:
                vals = np.stack([np.broadcast_to(v.apply_np({}, main), (n,)) for v in lk.values], axis=1)
                mult = np.broadcast_to(lk.multiplicity.apply_np({}, main), (n,))
                for i in np.nonzero((vals != vals[0]).any(axis=1) | (mult != mult[0]))[0]:
                    d = delta[i]
                    for which, weight in ((0, -sign), (i, sign)):
                        if mult[which]:
                            key = (lk.kind,) + tuple(int(x) for x in vals[which])
                            d[key] = (d.get(key, 0) + weight * int(mult[which])) % F.P
        for k, col in enumerate(cand):
            caught[col] = any(delta[1 + k].values())
    return [int(c) for c in np.nonzero(~caught)[0]]


def test_every_column_of_every_core_chip_is_bound(oracle):
    m = M.run_machine(2500, seed=15, shard_cycles=1 << 20, poseidon2_calls=1)
    seen, holes = sweep(oracle, m)
    assert seen >= {"Cpu", "AddSub", "Bitwise", "Lt", "ShiftLeft", "ShiftRight", "CloClz", "Mul", "DivRem", "Branch", "Jump", "MovCond", "MemoryInstrs",
                    "MiscInstrs", "SyscallInstrs", "MemoryLocal", "Global", "SyscallCore", "SyscallPrecompile", "Poseidon2Permute", "MemoryGlobalInit",
                    "MemoryGlobalFinalize"}, seen
    assert holes == [], holes


RECURSION_FREE = {
    ("MemoryConst", 0): "MemoryCols::_nothing (crates/recursion/core/src/chips/mem/constant.rs:26-29): a placeholder column, the chip's data is all preprocessed",
}


def test_every_column_of_every_recursion_chip_is_bound(oracle):
    """The same sweep over the eleven RecursionAir chips (all-chips machine + the wrap machine's skinny hash chip). Their program
    (addresses, multiplicities, flags) lives in the preprocessed trace, so only the main columns — the executed values — are swept."""
    from ziren_amd import recursion as R
    from test_recursion_chips import all_chips_shard, wrap_machine_shard, recursion_public_values, tally_of
    rec_of = {"BaseAlu": lambda: R.record_constraints(False), "ExtAlu": lambda: R.record_constraints(True),
              "MemoryConst": lambda: R.record_mem_const(constraints_only=True), "MemoryVar": lambda: R.record_mem_var(constraints_only=True),
              "Select": lambda: R.record_select(constraints_only=True), "Poseidon2WideDeg3": lambda: R.record_poseidon2_wide(constraints_only=True),
              "ExpReverseBitsLen": lambda: R.record_exp_reverse_bits(constraints_only=True), "BatchFRI": lambda: R.record_batch_fri(constraints_only=True),
              "PublicValues": lambda: R.record_public_values(constraints_only=True), "FriFold": lambda: R.record_fri_fold(constraints_only=True),
              "Poseidon2SkinnyDeg9": lambda: R.record_poseidon2_skinny(constraints_only=True)}
    seen, holes = set(), []
    for recs, streams, digest in (all_chips_shard(oracle), wrap_machine_shard(oracle)):
        pv = F.from_monty(recursion_public_values(digest)).astype(np.uint64)
        for c in recs:
            if c.name in seen or c.name not in rec_of:
                continue
            seen.add(c.name)
            rec = rec_of[c.name]()
            main, prep = F.from_monty(c.trace), F.from_monty(c.prep_trace)
            assert air.debug_constraints(rec.b, main, prep=prep, public_values=pv) == [], c.name
            base = tally_of([c])
            rows = np.nonzero(prep.any(axis=1))[0]
            pick = sorted({int(rows[0]), int(rows[-1])} | {int(rows[i]) for i in np.linspace(0, len(rows) - 1, 12).astype(int)})
            for col in range(main.shape[1]):
                caught = False
                for row in pick:
                    bad = main.copy()
                    bad[row, col] = (int(bad[row, col]) + 1) % F.P
                    if air.debug_constraints(rec.b, bad, prep=prep, public_values=pv):
                        caught = True
                        break
                    c2 = chips.RecordedChip(name=c.name, log_height=c.log_height, main_width=c.main_width, prep_width=c.prep_width, sends=c.sends, receives=c.receives)
                    c2.trace, c2.prep_trace = F.to_monty(bad), c.prep_trace
                    if tally_of([c2]) != base:
                        caught = True
                        break
                if not caught and (c.name, col) not in RECURSION_FREE:
                    holes.append((c.name, col))
    assert seen == set(rec_of), seen
    assert holes == [], holes
