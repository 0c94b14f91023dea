"""Constraint recorder: the host-side mirror of the reference's AIR builder traits.

In Ziren a chip's constraints are Rust code (`Air::eval`) run against a builder; the prover
runs it with `ProverConstraintFolder` (crates/stark/src/folder.rs:19-149). A GPU prover cannot
call Rust per row, so the shim *records* `eval` once with a symbolic builder — the mechanism
the reference itself uses for lookups (crates/stark/src/lookup/builder.rs:14-112) and for
constraint counting (crates/stark/src/machine.rs:377) — into the straight-line bytecode of
include/zkm_hip.h. This module is that recorder in Python (the Rust one is sketched in
INTEGRATION.md), with the same method names as the builder traits
(crates/stark/src/air/builder.rs:47-570): main/preprocessed/permutation, is_first_row,
is_last_row, is_transition, when_*, assert_zero/assert_eq(_ext), public_values,
permutation_randomness, local/global_cumulative_sum.

`eval_permutation_constraints` follows crates/stark/src/permutation.rs:205-347 statement by
statement so that the constraint order (hence the alpha powers) is the reference's.
"""
from dataclasses import dataclass, field as dc_field
from typing import List, Optional, Tuple

import numpy as np

from . import field as F

# opcodes — keep in sync with include/zkm_hip.h
LD_MAIN, LD_PREP, LD_PERM, LD_CONST, LD_PV, LD_CHALLENGE = 1, 2, 3, 4, 5, 6
LD_LOCAL_SUM, LD_GLOBAL_SUM, LD_IS_FIRST, LD_IS_LAST, LD_IS_TRANS = 7, 8, 9, 10, 11
ADD_B, SUB_B, MUL_B, NEG_B = 16, 17, 18, 19
ADD_E, SUB_E, MUL_E, NEG_E = 20, 21, 22, 23
ADD_EB, SUB_EB, MUL_EB = 24, 25, 26
ASSERT_B, ASSERT_E = 32, 33

# LookupKind (crates/stark/src/lookup/lookup.rs:22-48)
KIND_MEMORY, KIND_PROGRAM, KIND_INSTRUCTION, KIND_BYTE = 1, 2, 3, 4
KIND_RANGE, KIND_SYSCALL, KIND_GLOBAL, KIND_SYSCALL_RESULT = 5, 6, 7, 8


class RegisterPressure(Exception):
    """A constraint program that keeps every value in a register until its last use needs more than the 256 registers an instruction can name."""


class Expr:
    """Node of the recorded expression DAG. ext=False: base field, True: extension field."""

    __slots__ = ("b", "op", "a", "c", "imm", "ext", "uses", "reg")

    def __init__(self, b, op, a=None, c=None, imm=0, ext=False):
        self.b, self.op, self.a, self.c, self.imm, self.ext = b, op, a, c, imm, ext
        self.uses = 0
        self.reg = None

    def _coerce(self, o):
        if isinstance(o, Expr):
            return o
        return self.b.const(int(o))

    def __add__(self, o):
        o = self._coerce(o)
        if self.ext and o.ext:
            return Expr(self.b, ADD_E, self, o, ext=True)
        if self.ext:
            return Expr(self.b, ADD_EB, self, o, ext=True)
        if o.ext:
            return Expr(self.b, ADD_EB, o, self, ext=True)
        return Expr(self.b, ADD_B, self, o)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._coerce(o)
        if self.ext and o.ext:
            return Expr(self.b, SUB_E, self, o, ext=True)
        if self.ext:
            return Expr(self.b, SUB_EB, self, o, ext=True)
        if o.ext:  # base - ext = -(ext - base)
            return Expr(self.b, NEG_E, Expr(self.b, SUB_EB, o, self, ext=True), ext=True)
        return Expr(self.b, SUB_B, self, o)

    def __rsub__(self, o):
        return self._coerce(o) - self

    def __mul__(self, o):
        o = self._coerce(o)
        if self.ext and o.ext:
            return Expr(self.b, MUL_E, self, o, ext=True)
        if self.ext:
            return Expr(self.b, MUL_EB, self, o, ext=True)
        if o.ext:
            return Expr(self.b, MUL_EB, o, self, ext=True)
        return Expr(self.b, MUL_B, self, o)

    __rmul__ = __mul__

    def __neg__(self):
        return Expr(self.b, NEG_E if self.ext else NEG_B, self, ext=self.ext)


@dataclass
class VirtualPairCol:
    """p3_air::VirtualPairCol: sum_i weight_i * (preprocessed|main)[col_i] + constant (canonical ints)."""

    terms: List[Tuple[bool, int, int]] = dc_field(default_factory=list)  # (is_main, column, weight)
    constant: int = 0

    @staticmethod
    def single_main(col):
        return VirtualPairCol([(True, col, 1)], 0)

    @staticmethod
    def single_preprocessed(col):
        return VirtualPairCol([(False, col, 1)], 0)

    @staticmethod
    def const(c):
        return VirtualPairCol([], c % F.P)

    def apply_expr(self, b, prep_row, main_row):
        acc = b.const(self.constant)
        for is_main, col, w in self.terms:
            v = main_row[col] if is_main else prep_row[col]
            acc = acc + v * b.const(w)
        return acc

    def apply_np(self, prep, main):
        """Evaluate on whole column-arrays (dict col -> uint64 array), canonical."""
        n = len(next(iter(main.values()))) if main else len(next(iter(prep.values())))
        acc = np.full(n, self.constant % F.P, dtype=np.uint64)
        for is_main, col, w in self.terms:
            v = main[col] if is_main else prep[col]
            acc = F.add(acc, F.mul(v, w))
        return acc


@dataclass
class Lookup:
    """crates/stark/src/lookup/lookup.rs:9-19 (Local scope only crosses the ABI)."""

    values: List[VirtualPairCol]
    multiplicity: VirtualPairCol
    kind: int


def encode_lookups(sends: List[Lookup], receives: List[Lookup]) -> np.ndarray:
    """Serialise to the `lookups` blob of zkm_chip_desc (Montgomery constants)."""
    w = [len(sends), len(receives)]
    for lk in list(sends) + list(receives):
        w += [lk.kind, len(lk.values)]
        for pc in list(lk.values) + [lk.multiplicity]:
            w += [len(pc.terms), F.to_monty(pc.constant)]
            for is_main, col, weight in pc.terms:
                w += [(int(is_main) << 31) | col, F.to_monty(weight)]
    return np.array(w, dtype=np.uint32)


def to_virtual_pair(e) -> VirtualPairCol:
    """symbolic_to_virtual_pair (crates/stark/src/lookup/builder.rs:123-180): an expression that is affine in the
    current row's main / preprocessed columns becomes column weights + constant; anything else is an error."""
    if not isinstance(e, Expr):
        return VirtualPairCol.const(int(e))

    def walk(x):
        if x.op == LD_CONST:
            return [], int(F.from_monty(np.uint32(x.imm)))
        if x.op == LD_MAIN or x.op == LD_PREP:
            if x.a != 0:
                raise ValueError("not an affine expression in current row elements")
            return [(x.op == LD_MAIN, x.imm, 1)], 0
        if x.op == ADD_B or x.op == SUB_B:
            vl, cl = walk(x.a)
            vr, cr = walk(x.c)
            if x.op == SUB_B:
                vr, cr = [(m, c, (F.P - w) % F.P) for m, c, w in vr], (F.P - cr) % F.P
            return vl + vr, (cl + cr) % F.P
        if x.op == NEG_B:
            v, c = walk(x.a)
            return [(m, col, (F.P - w) % F.P) for m, col, w in v], (F.P - c) % F.P
        if x.op == MUL_B:
            vl, cl = walk(x.a)
            vr, cr = walk(x.c)
            if vl and vr:
                raise ValueError("not an affine expression")
            return [(m, c, w * cr % F.P) for m, c, w in vl] + [(m, c, w * cl % F.P) for m, c, w in vr], cl * cr % F.P
        raise ValueError("not an affine expression in current row elements")

    terms, const = walk(e)
    return VirtualPairCol(terms, const)


class AirBuilder:
    """Symbolic builder that records constraints as bytecode."""

    def __init__(self, main_width, prep_width, perm_ext_width):
        self.main_width, self.prep_width, self.perm_ext_width = main_width, prep_width, perm_ext_width
        self.asserts: List[Expr] = []
        self._cache = {}

    # --- inputs ------------------------------------------------------------------------------
    def _leaf(self, op, a=0, imm=0, ext=False):
        key = (op, a, imm)
        e = self._cache.get(key)
        if e is None:
            e = Expr(self, op, a, None, imm, ext)
            self._cache[key] = e
        return e

    def main(self):
        return ([self._leaf(LD_MAIN, 0, c) for c in range(self.main_width)],
                [self._leaf(LD_MAIN, 1, c) for c in range(self.main_width)])

    def preprocessed(self):
        return ([self._leaf(LD_PREP, 0, c) for c in range(self.prep_width)],
                [self._leaf(LD_PREP, 1, c) for c in range(self.prep_width)])

    def permutation(self):
        return ([self._leaf(LD_PERM, 0, c, True) for c in range(self.perm_ext_width)],
                [self._leaf(LD_PERM, 1, c, True) for c in range(self.perm_ext_width)])

    def const(self, v):
        return self._leaf(LD_CONST, 0, F.to_monty(int(v) % F.P))

    def public_values(self, i):
        return self._leaf(LD_PV, 0, i)

    def permutation_randomness(self):
        return [self._leaf(LD_CHALLENGE, 0, 0, True), self._leaf(LD_CHALLENGE, 0, 1, True)]

    def local_cumulative_sum(self):
        return self._leaf(LD_LOCAL_SUM, 0, 0, True)

    def global_cumulative_sum(self, i):
        return self._leaf(LD_GLOBAL_SUM, 0, i)

    def is_first_row(self):
        return self._leaf(LD_IS_FIRST)

    def is_last_row(self):
        return self._leaf(LD_IS_LAST)

    def is_transition(self):
        return self._leaf(LD_IS_TRANS)

    # --- constraints ---------------------------------------------------------------------------
    def assert_zero(self, e: Expr):
        self.asserts.append(e)

    def assert_eq(self, a, b):
        self.assert_zero(a - b)

    assert_zero_ext = assert_zero
    assert_eq_ext = assert_eq

    def assert_one(self, e):
        self.assert_zero(e - 1)

    def assert_bool(self, e):
        self.assert_zero(e * (e - 1))

    def when(self, cond: Expr):
        return _Filtered(self, cond)

    def when_not(self, cond: Expr):
        return self.when(1 - cond)

    def when_first_row(self):
        return self.when(self.is_first_row())

    def when_last_row(self):
        return self.when(self.is_last_row())

    def when_transition(self):
        return self.when(self.is_transition())

    # --- emission ------------------------------------------------------------------------------
    def assemble(self) -> np.ndarray:
        """Emit the `program` blob: header {n_instr, n_ext_regs, n_constraints, n_base_regs} + 2 words/instr. A value is computed once and
        kept in its register until its last use. When that needs more than the 256 registers an instruction can name (a chip whose columns
        are each read by constraints far apart: KeccakSponge), the inputs are loaded again at every use instead of being kept."""
        try:
            return self._assemble(False)
        except RegisterPressure as e:
            import logging
            logging.getLogger(__name__).info("%s: assembling again with inputs reloaded at every use (a longer program)", e)
            return self._assemble(True)

    def _assemble(self, reload_inputs: bool) -> np.ndarray:
        # use counts over the whole DAG (each node evaluated once)
        seen = set()

        def reset(e):
            stack = [e]
            while stack:
                x = stack.pop()
                if id(x) in seen:
                    continue
                seen.add(id(x))
                x.uses, x.reg = 0, None
                if x.op >= ADD_B:
                    stack.append(x.a)
                    if x.c is not None:
                        stack.append(x.c)

        for a in self.asserts:
            reset(a)
        seen.clear()

        def count(e):
            stack = [e]
            while stack:
                x = stack.pop()
                x.uses += 1
                if id(x) in seen:
                    continue
                seen.add(id(x))
                if x.op >= ADD_B:
                    stack.append(x.a)
                    if x.c is not None:
                        stack.append(x.c)

        for a in self.asserts:
            count(a)
        instrs = []
        free = {False: [], True: []}   # separate register files for base and extension values
        nregs = {False: 0, True: 0}

        def alloc(ext):
            if free[ext]:
                return free[ext].pop()
            r = nregs[ext]
            nregs[ext] += 1
            if r > 255:
                raise RegisterPressure("constraint program needs more than 256 registers")
            return r

        def is_input(x):
            return x.op < ADD_B

        def release(x):
            x.uses -= 1
            if reload_inputs and is_input(x):
                if x.reg is not None:
                    free[x.ext].append(x.reg)
                    x.reg = None
            elif x.uses == 0:
                free[x.ext].append(x.reg)
                x.reg = None

        def load(x):
            x.reg = alloc(x.ext)
            instrs.append((x.op | x.reg << 8 | (x.a & 0xFF) << 16, x.imm))

        def emit(e):
            # iterative post-order
            stack = [(e, False)]
            while stack:
                x, ready = stack.pop()
                if x.reg is not None:
                    continue
                if is_input(x):
                    if not reload_inputs or x is e:
                        load(x)
                    continue
                if not ready:
                    stack.append((x, True))
                    if x.c is not None and x.c.reg is None and not (reload_inputs and is_input(x.c)):
                        stack.append((x.c, False))
                    if x.a.reg is None and not (reload_inputs and is_input(x.a)):
                        stack.append((x.a, False))
                    continue
                if reload_inputs:       # inputs are loaded right where they are consumed
                    for ch in (x.a, x.c):
                        if ch is not None and is_input(ch) and ch.reg is None:
                            load(ch)
                ra = x.a.reg
                rc = x.c.reg if x.c is not None else 0
                release(x.a)
                if x.c is not None:
                    release(x.c)
                x.reg = alloc(x.ext)
                instrs.append((x.op | x.reg << 8 | ra << 16 | rc << 24, 0))

        for a in self.asserts:
            emit(a)
            instrs.append(((ASSERT_E if a.ext else ASSERT_B) | a.reg << 16, 0))
            release(a)
        words = [len(instrs), max(nregs[True], 1), len(self.asserts), max(nregs[False], 1)]
        for w0, w1 in instrs:
            words += [w0 & 0xFFFFFFFF, w1 & 0xFFFFFFFF]
        return np.array(words, dtype=np.uint32)


class _Filtered:
    """FilteredAirBuilder: multiplies every asserted expression by the condition."""

    def __init__(self, b, cond):
        self.b, self.cond = b, cond

    def assert_zero(self, e):
        self.b.assert_zero(e * self.cond)

    def assert_eq(self, a, c):
        self.assert_zero(a - c)

    assert_zero_ext = assert_zero
    assert_eq_ext = assert_eq

    def assert_one(self, e):
        self.assert_zero(e - 1)

    def assert_bool(self, e):
        self.assert_zero(e * (e - 1))

    def when(self, cond):
        return _Filtered(self.b, self.cond * cond)

    def when_not(self, cond):
        return self.when(1 - cond)


def local_permutation_trace_width(nb_lookups, batch_size):
    """permutation.rs:18-23"""
    return 0 if nb_lookups == 0 else -(-nb_lookups // batch_size) + 1


def eval_permutation_constraints(b: AirBuilder, sends, receives, batch_size, commit_scope_global):
    """crates/stark/src/permutation.rs:205-347."""
    width = local_permutation_trace_width(len(sends) + len(receives), batch_size)
    assert width == b.perm_ext_width
    prep_local, _ = b.preprocessed()
    main_local, _ = b.main()
    perm_local, perm_next = b.permutation()
    alpha, beta = b.permutation_randomness()
    if sends or receives:
        lookups = [(lk, True) for lk in sends] + [(lk, False) for lk in receives]
        chunks = [lookups[i:i + batch_size] for i in range(0, len(lookups), batch_size)]
        for entry, chunk in zip(perm_local[:-1], chunks):
            rlcs, mults = [], []
            for lk, is_send in chunk:
                rlc = alpha + b.const(lk.kind)  # beta^0 * argument_index
                bp = beta
                for i, v in enumerate(lk.values):
                    rlc = rlc + bp * v.apply_expr(b, prep_local, main_local)
                    if i + 1 < len(lk.values):
                        bp = bp * beta
                rlcs.append(rlc)
                m = lk.multiplicity.apply_expr(b, prep_local, main_local)
                mults.append(m if is_send else -m)
            product = None
            numerator = None
            for i, (m, rlc) in enumerate(zip(mults, rlcs)):
                product = rlc if product is None else product * rlc
                all_but = None
                for j, other in enumerate(rlcs):
                    if j != i:
                        all_but = other if all_but is None else all_but * other
                term = m if all_but is None else all_but * m
                numerator = term if numerator is None else numerator + term
            b.assert_eq_ext(product * entry, numerator)
        sum_local = None
        sum_next = None
        for x, y in zip(perm_local[:-1], perm_next[:-1]):
            sum_local = x if sum_local is None else sum_local + x
            sum_next = y if sum_next is None else sum_next + y
        phi_local, phi_next = perm_local[-1], perm_next[-1]
        b.when_first_row().assert_eq_ext(phi_local, sum_local)
        b.when_transition().assert_eq_ext(phi_next - phi_local, sum_next)
        b.when_last_row().assert_eq_ext(phi_local, b.local_cumulative_sum())
    if commit_scope_global:
        n = len(main_local)
        for i in range(7):
            b.when_last_row().assert_eq(main_local[n - 14 + i], b.global_cumulative_sum(i))
            b.when_last_row().assert_eq(main_local[n - 7 + i], b.global_cumulative_sum(7 + i))


def count_permutation_constraints(n_lookups, batch_size, commit_scope_global):
    """permutation.rs:355-389"""
    c = 0
    if n_lookups:
        c += local_permutation_trace_width(n_lookups, batch_size) - 1 + 3
    if commit_scope_global:
        c += 14
    return c


def _constraint_values(b: "AirBuilder", main: np.ndarray, public_values=None, prep: np.ndarray = None):
    """Yield (k, values over the rows) for every recorded base-field constraint k; row i is evaluated against row (i + 1) mod n."""
    n = main.shape[0]
    cols = [main[:, c].astype(np.uint64) for c in range(main.shape[1])]
    nxt = [np.roll(c, -1) for c in cols]
    pcols = [prep[:, c].astype(np.uint64) for c in range(prep.shape[1])] if prep is not None else []
    pnxt = [np.roll(c, -1) for c in pcols]
    rows = np.arange(n)
    memo = {}

    def ev(x):
        key = id(x)
        if key in memo:
            return memo[key]
        if x.ext:
            raise ValueError("extension-field expression")
        if x.op == LD_MAIN:
            return (nxt if x.a else cols)[x.imm]
        if x.op == LD_PREP:
            return (pnxt if x.a else pcols)[x.imm]
        if x.op == LD_CONST:
            v = np.full(n, int(F.from_monty(np.uint32(x.imm))), dtype=np.uint64)
        elif x.op == LD_PV:
            v = np.full(n, int(public_values[x.imm]), dtype=np.uint64)
        elif x.op == LD_IS_FIRST:
            v = (rows == 0).astype(np.uint64)
        elif x.op == LD_IS_LAST:
            v = (rows == n - 1).astype(np.uint64)
        elif x.op == LD_IS_TRANS:
            v = (rows != n - 1).astype(np.uint64)
        elif x.op == ADD_B:
            v = F.add(ev(x.a), ev(x.c))
        elif x.op == SUB_B:
            v = F.sub(ev(x.a), ev(x.c))
        elif x.op == MUL_B:
            v = F.mul(ev(x.a), ev(x.c))
        elif x.op == NEG_B:
            v = F.sub(np.zeros(n, dtype=np.uint64), ev(x.a))
        else:
            raise ValueError(f"opcode {x.op} not supported by debug_constraints")
        memo[key] = v
        return v

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    for k, a in enumerate(b.asserts):
        if a.ext:
            continue
        yield k, ev(a)
        if len(memo) * n * 8 > (1 << 30):     # values shared between constraints are kept, up to a gigabyte of them
            memo.clear()


def constraint_values_at_point(b: "AirBuilder", n_public_values: int = 231) -> List[int]:
    """Every recorded base-field constraint, in evaluation order, at the fixed point integration/zkm-hip/examples/dump_golden.rs evaluates the
    reference's symbolic constraints at (its `airs.txt`): main local / next column c = value(7, 0 / 1, c), preprocessed = value(11, 0 / 1, c),
    public value i = value(13, 0, i), is_first_row / is_last_row / is_transition = value(17, 0, 0 / 1 / 2), with
    value(seed, r, c) = ((seed * 2654435761 + r * 40503 + c * 9973 + 12345) mod 2^32) mod p. Canonical integers."""
    val = lambda seed, r, c: ((seed * 2654435761 + r * 40503 + c * 9973 + 12345) % (1 << 32)) % F.P   # noqa: E731
    main = np.array([[val(7, r, c) for c in range(b.main_width)] for r in range(2)], dtype=np.uint64)
    prep = np.array([[val(11, r, c) for c in range(max(b.prep_width, 1))] for r in range(2)], dtype=np.uint64)
    pv = [val(13, 0, i) for i in range(n_public_values)]
    sel = {LD_IS_FIRST: val(17, 0, 0), LD_IS_LAST: val(17, 0, 1), LD_IS_TRANS: val(17, 0, 2)}
    memo = {}

    def ev(x):
        k = id(x)
        if k not in memo:
            if x.op == LD_MAIN:
                v = int(main[1 if x.a else 0, x.imm])
            elif x.op == LD_PREP:
                v = int(prep[1 if x.a else 0, x.imm])
            elif x.op == LD_CONST:
                v = int(F.from_monty(np.uint32(x.imm)))
            elif x.op == LD_PV:
                v = pv[x.imm]
            elif x.op in sel:
                v = sel[x.op]
            elif x.op == ADD_B:
                v = (ev(x.a) + ev(x.c)) % F.P
            elif x.op == SUB_B:
                v = (ev(x.a) - ev(x.c)) % F.P
            elif x.op == MUL_B:
                v = ev(x.a) * ev(x.c) % F.P
            elif x.op == NEG_B:
                v = -ev(x.a) % F.P
            else:
                raise ValueError(f"opcode {x.op} in a main constraint")
            memo[k] = v
        return memo[k]

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    return [ev(a) for a in b.asserts if not a.ext]


def debug_constraints(b: "AirBuilder", main: np.ndarray, public_values=None, prep: np.ndarray = None) -> List[Tuple[int, int]]:
    """Evaluate the recorded base-field constraints on every row of a trace (canonical values, row-major),
    row i against row (i + 1) mod n as crates/stark/src/debug.rs:30-120 does. Returns [(constraint, first failing
    row)]; empty when the trace satisfies the AIR. Extension-field (permutation) constraints are not covered."""
    bad = []
    for k, v in _constraint_values(b, main, public_values, prep):
        nz = np.nonzero(v)[0]
        if len(nz):
            bad.append((k, int(nz[0])))
    return bad


def violated_rows(b: "AirBuilder", main: np.ndarray, public_values=None, prep: np.ndarray = None) -> np.ndarray:
    """Per row: does any recorded base-field constraint fail there (the row as `local`, the next one as `next`)."""
    out = np.zeros(main.shape[0], dtype=bool)
    for _, v in _constraint_values(b, main, public_values, prep):
        out |= v != 0
    return out
