"""Per-chip quotient kernels: constraint bytecode -> straight-line HIP -> gfx950 code object.

The bytecode interpreter (`stark::quotient_kernel`) keeps its register files in LDS and pays a
dispatch, two LDS reads and an exposed global-load latency per instruction. A chip's AIR is fixed
for the life of a machine (Ziren has 50 `MipsAir` variants, crates/core/machine/src/mips/mod.rs:73-190),
so the same bytecode can be turned once into a specialised kernel whose values live in VGPRs and whose
loads the compiler hoists and batches. `specialize()` emits that kernel, compiles it with hipcc
(`--genco`, gfx950) into an in-tree cache, and `HipProver` registers the code object with the library
(`zkm_ctx_register_quotient_kernel`); `zkm_open` then picks it by the hash of the chip's program words
and falls back to the interpreter for unregistered programs. Both paths compute the same field values.

In the Rust integration this runs in the shim's build script (or `HipProver::new`) — INTEGRATION.md.
"""
import hashlib
import os
import subprocess
import tempfile
from typing import Optional

import numpy as np

from . import air

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
CACHE = os.path.join(HERE, "_jit")
KERNEL_NAME = "zkm_quotient_specialized"
UNIFORMS_KERNEL_NAME = "zkm_quotient_uniforms"
BLOCK = 256
PART_INSTRS = int(os.environ.get("ZKM_Q_PART", "1500"))              # a long program is cut into kernels of about this many statements (KeccakSponge: 6000 -> 3.58 ms of quotient per shard, 3000 -> 3.12, 1500 -> 2.75, 800 -> 3.01: a 200 KiB kernel against a 64 KiB instruction cache)
SINGLE_KERNEL_INSTRS = int(os.environ.get("ZKM_Q_SINGLE", "12000"))  # ... when it is longer than this (KeccakSponge: 114 324; every other recorded chip is below 10 000)


TEMPLATE_VERSION = b"15"  # bump when emit_source changes


# Experiment knobs (tools/ab_quotient.sh; unset in production): waves per SIMD the compiler is told to fit the kernel into, and where a
# program is cut into several kernels. They are part of the cache key.
Q_WAVES = int(os.environ.get("ZKM_Q_WAVES", "0"))
Q_AHEAD = int(os.environ.get("ZKM_Q_AHEAD", "1"))           # how many groups ahead
Q_TILE = int(os.environ.get("ZKM_Q_TILE", "1"))           # quotient_args.cuh: 1 = the 8 x 32 tile with staged vector stores, 0 = rounds 2-4's two half-tiles (A/B only)
Q_PAIR = int(os.environ.get("ZKM_Q_PAIR", "1"))           # pair_row_loads: a column's `next` load right behind its `local` load (0: where the program has them; A/B only)
P_GROUP = int(os.environ.get("ZKM_P_GROUP", "4"))       # emit_perm_source: fraction columns whose inversions share one base-field inversion (Montgomery's trick)
Q_PREFETCH = int(os.environ.get("ZKM_Q_PREFETCH", "4"))     # words of trace loads per group, issued Q_AHEAD groups ahead of their use (0: the compiler's
                                                            # order, which sinks every load to its first use; round 4: 5.35 -> 4.55 ms on the benchmarked shard)


def _template_key() -> bytes:
    """Cached code objects are keyed on the program *and* on everything the generated source pulls in, so an edit to
    the shared prologue (quotient_args.cuh) or the field arithmetic (kb31.cuh) can never leave a stale kernel behind."""
    h = hashlib.sha256(TEMPLATE_VERSION)
    h.update(f"{Q_WAVES},{SINGLE_KERNEL_INSTRS},{PART_INSTRS}".encode())
    h.update(f",pf{Q_PREFETCH},{Q_AHEAD},tile{Q_TILE},pair{Q_PAIR},forms{Q_FORMS},pg{P_GROUP},sums{Q_SUMS},ut{Q_UNITABLE},pt{Q_PARTS_TABLE}".encode() + (b",fakeuni" if Q_FAKEUNI else b""))
    for name in ("quotient_args.cuh", "kb31.cuh"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.digest()


def program_hash(program: np.ndarray) -> str:
    return hashlib.sha256(_template_key() + np.ascontiguousarray(program, dtype=np.uint32).tobytes()).hexdigest()[:24]


P = 0x7F000001
MONTY_ONE = 0x01FFFFFE
Q_FORMS = int(os.environ.get("ZKM_Q_FORMS", "1"))           # emit_form: 1 = bounded accumulators (round 5), 0 = rounds 3-4's fold_zero / fold_finish + modular additions (A/B only)


Q_UNITABLE = int(os.environ.get("ZKM_Q_UNITABLE", "1"))     # split_uniform: 1 = wave-uniform values computed once per launch into a table (round 5), 0 = by every wavefront (A/B only)
Q_PARTS_TABLE = int(os.environ.get("ZKM_Q_PARTS_TABLE", "0"))  # the table kernel for programs cut into parts as well (measured: no gain; A/B only)
Q_FAKEUNI = int(os.environ.get("ZKM_Q_FAKEUNI", "0"))       # EXPERIMENT ONLY: derived wave-uniform values loaded from a table (wrong values; timing of a uniform-table design)
Q_SUMS = int(os.environ.get("ZKM_Q_SUMS", "1"))             # _ssa_lines: 1 = base-field sums of products reduced once (round 5), 0 = one modular operation per bytecode instruction (A/B only)
MAX_SUM_TERMS = 48                                          # a deferred base-field sum is emitted when it reaches this many terms (its bound stays below 127 * 2^63)


def _base_read_counts(prog: np.ndarray):
    """How many instructions read each base-field SSA value (same renaming as _ssa_lines: one fresh name per defining instruction)."""
    n_instr = int(prog[0])
    cur_b, reads, nv = {}, {}, 0
    base_defs = (air.LD_MAIN, air.LD_PREP, air.LD_CONST, air.LD_PV, air.LD_GLOBAL_SUM, air.LD_IS_FIRST, air.LD_IS_LAST, air.LD_IS_TRANS,
                 air.ADD_B, air.SUB_B, air.MUL_B, air.NEG_B)
    ext_defs = (air.LD_PERM, air.LD_CHALLENGE, air.LD_LOCAL_SUM, air.ADD_E, air.SUB_E, air.MUL_E, air.NEG_E, air.ADD_EB, air.SUB_EB, air.MUL_EB)
    for k in range(n_instr):
        w0 = int(prog[4 + 2 * k])
        op, dst, ra, rb = w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, w0 >> 24
        if op in (air.ADD_B, air.SUB_B, air.MUL_B):
            srcs = (cur_b[ra], cur_b[rb])
        elif op in (air.NEG_B, air.ASSERT_B):
            srcs = (cur_b[ra],)
        elif op in (air.ADD_EB, air.SUB_EB, air.MUL_EB):
            srcs = (cur_b[rb],)
        else:
            srcs = ()
        for z in srcs:
            reads[z] = reads.get(z, 0) + 1
        if op in base_defs:
            nv += 1
            cur_b[dst] = f"b{nv}"
        elif op in ext_defs:
            nv += 1
    return reads


def emit_sum(name: str, prods, leaves, uniform=()) -> str:
    """Statements that define `const uint32_t name` = sum(sign * x * y for prods) + sum(sign * z for leaves) as a reduced word, x, y, z
    reduced words (row values or wave-uniform ones). Few terms: the modular operations the bytecode names. Otherwise every product and
    leaf goes into one integer sum — a leaf as z * (R mod p) (or z * (p - R mod p) when subtracted), a subtracted product as (p - x) * y
    — which is reduced once: an ordinary Montgomery reduction below 2^32 p, reduce96_bounded below 2^64, a 96-bit accumulator beyond.
    Costs (cycles per wavefront, tools/ubench_int): a modular product 22.9, a modular addition 9.3, v_mad_u64_u32 5."""
    k, m = len(prods), len(leaves)
    pos = [z for sg, z in leaves if sg > 0]
    neg = [z for sg, z in leaves if sg < 0]

    def chain(first, first_negated):
        expr = first
        rest_pos, rest_neg = list(pos), list(neg)
        if expr is None:
            if rest_pos:
                expr = rest_pos.pop(0)
            else:
                expr = f"kb::neg({rest_neg.pop(0)})" if len(rest_neg) == 1 else None
                if expr is None:
                    z = rest_neg.pop(0)
                    expr = z
                    for w in rest_neg:
                        expr = f"kb::add({expr}, {w})"
                    return f"kb::neg({expr})"
        elif first_negated:
            # -(x y) + ...: start from a positive leaf if there is one
            if rest_pos:
                expr = f"kb::sub({rest_pos.pop(0)}, {first})"
            else:
                expr = f"kb::neg({first})"
        for w in rest_pos:
            expr = f"kb::add({expr}, {w})"
        for w in rest_neg:
            expr = f"kb::sub({expr}, {w})"
        return expr

    if k == 0 and m < 6:
        return f"const uint32_t {name} = {chain(None, False)};"
    if k == 1 and m == 0:
        sg, x, y = prods[0]
        return f"const uint32_t {name} = " + (f"kb::mul({x}, {y});" if sg > 0 else f"kb::monty_reduce((uint64_t)(kb::P - {x}) * {y});")
    terms, bound, extra = [], 0, 0.0
    for sg, x, y in prods:
        if sg > 0:
            terms.append((x, y)); bound += (P - 1) ** 2
        else:
            if y in uniform and x not in uniform:
                x, y = y, x
            terms.append((f"(kb::P - {x})", y)); bound += P * (P - 1)
            extra += 0.0 if x in uniform else 2.5
    for z in pos:
        terms.append((z, "kb::ONE")); bound += (P - 1) * MONTY_ONE
    for z in neg:
        terms.append((z, "(kb::P - kb::ONE)")); bound += (P - 1) * (P - MONTY_ONE)
    n = len(terms)
    cost_chain = 22.9 * k + 9.3 * (k + m - 1)
    cost_sum = 5.0 * n + extra + (16.1 if bound < P << 32 else 30.0 if bound < 1 << 64 else 34.0 + 2.5 * n)
    if k == 1 and cost_chain <= cost_sum:
        sg, x, y = prods[0]
        return f"const uint32_t {name} = {chain(f'kb::mul({x}, {y})', sg < 0)};"
    assert bound < 127 << 63, "a deferred base-field sum beyond reduce96_bounded's bound (MAX_SUM_TERMS)"
    if bound < 1 << 64:
        total = " + ".join(f"(uint64_t){a} * {b}" for a, b in terms)
        return f"const uint32_t {name} = " + (f"kb::monty_reduce({total});" if bound < P << 32 else f"kb::reduce96_bounded(0, {total});")
    acc = f"s_{name}"
    stmts = [f"kb::Acc96 {acc} = kb::acc96_zero();"] + [f"kb::acc96_fma({acc}, {a}, {b});" for a, b in terms]
    return " ".join(stmts) + f" const uint32_t {name} = kb::reduce96_bounded({acc}.hi, {acc}.lo);"


def emit_form(name: str, consts, uterms, vterms, extras) -> str:
    """Statements that define `const kb::E4 name` = sum(consts) + sum(u * b for uterms) + sum(e * b for vterms) + sum(extras): consts
    and the u are wave-uniform extension values (challenges and their powers: SGPR operands), e and extras row values, b base-field row
    values, every word reduced. The products go unreduced into one integer accumulator per coefficient and are reduced once. The
    generator knows the form's bound — a product is below (p - 1)^2, the constant enters as c 2^32 (the accumulator starts there: no
    addition afterwards), an extra as one product by R mod p — and picks the cheapest exact shape (csrc/kb31.cuh):
      * below 2^32 p: 64-bit accumulators (one v_mad_u64_u32 per product and coefficient), an ordinary Montgomery reduction;
      * below 2^64: 64-bit accumulators, reduce96_bounded;
      * below 127 * 2^63: 96-bit accumulators (product + carry), reduce96_bounded (ten instructions instead of twenty-two);
      * beyond (no recorded chip): rounds 3-4's shape — acc96_reduce and modular additions for the constant and the extras."""
    n_prod = len(uterms) + len(vterms)
    const = None
    for c in consts:
        const = c if const is None else f"kb::eadd({const}, {c})"
    bound = n_prod * (P - 1) ** 2 + (0 if const is None else (P - 1) << 32) + len(extras) * (P - 1) * MONTY_ONE
    acc = f"l_{name}"
    if not Q_FORMS or bound >= 127 << 63:
        stmts = [f"kb::FoldAcc {acc} = kb::fold_zero();"] + [f"kb::fold_base({acc}, {u}, {b});" for u, b in uterms]
        stmts += [f"kb::fold_scaled({acc}, {e}, {b});" for e, b in vterms]
        expr = f"kb::fold_finish({acc})"
        for c in list(consts) + list(extras):
            expr = f"kb::eadd({expr}, {c})"
        return " ".join(stmts) + f" const kb::E4 {name} = {expr};"
    if n_prod == 0 and not extras:
        return f"const kb::E4 {name} = {const if const is not None else 'kb::ezero()'};"
    if bound < 1 << 64:
        stmts = [f"kb::FoldAcc64 {acc} = " + (f"kb::fold64_from({const});" if const is not None else "kb::fold64_zero();")]
        stmts += [f"kb::fold64_scaled({acc}, {u}, {b});" for u, b in list(uterms) + list(vterms)]
        stmts += [f"kb::fold64_add({acc}, {e});" for e in extras]
        return " ".join(stmts) + f" const kb::E4 {name} = kb::fold64_finish<{'true' if bound < P << 32 else 'false'}>({acc});"
    stmts = [f"kb::FoldAcc {acc} = " + (f"kb::fold_from({const});" if const is not None else "kb::fold_zero();")]
    stmts += [f"kb::fold_base({acc}, {u}, {b});" for u, b in uterms] + [f"kb::fold_scaled({acc}, {e}, {b});" for e, b in vterms]
    stmts += [f"kb::fold_add({acc}, {e});" for e in extras]
    return " ".join(stmts) + f" const kb::E4 {name} = kb::fold_finish_bounded({acc});"


def _ssa_lines(program: np.ndarray, with_uniform: bool = False, sums: Optional[bool] = None):
    """The program as straight-line HIP statements, every register write a fresh SSA value: (lines, meta) with meta[k] = (the value line
    k defines — None for an assert —, the values it reads); with_uniform: and the set of values that are the same for every row."""
    prog = np.asarray(program, dtype=np.uint32)
    n_instr = int(prog[0])
    cur_b, cur_e = {}, {}   # register -> current variable name
    lines, meta = [], []
    nv = [0]
    # Linear forms (round 3). A LogUp denominator is alpha + kind + sum_k beta^k * value_k: in the bytecode a chain of
    # ADD_E(previous, MUL_EB(beta^k, value_k)) whose extension operands beta^k are the same for every row ("uniform": they depend on
    # the challenges, constants and public values only — the compiler keeps them in SGPRs). Such a chain is not emitted step by step
    # (a Montgomery product per coefficient and term plus a modular addition: 38 instructions per term) but kept as a list of
    # (uniform extension value, base value) terms and, when something first needs the sum, emitted as 96-bit dot products
    # (kb::fold_base: 8 instructions per term, one reduction per coefficient at the end).
    uniform = set()          # SSA values that are the same for every row
    deferred = {}            # SSA extension value -> {"terms": [(uniform ext, base)], "consts": [uniform ext], "extras": [ext]} not emitted yet

    def fresh(prefix):
        nv[0] += 1
        return f"{prefix}{nv[0]}"

    def use(name):
        """The name of a value about to be read: a deferred linear form is emitted first (one line, one meta entry)."""
        form = deferred.pop(name, None)
        if form is None:
            return name
        lines.append(emit_form(name, form["consts"], form["terms"], form["vterms"], form["extras"]))
        meta.append((name, tuple(x for t in form["terms"] + form["vterms"] for x in t) + tuple(form["consts"]) + tuple(form["extras"])))
        return name

    # Sums of products in the base field (round 5). ADD_B / SUB_B / NEG_B emit nothing: their value is kept as a signed list of products
    # (of two emitted, reduced values) and leaves (emitted, reduced values), and a MUL_B as one product. When something needs the value
    # as a reduced word — a multiplication, an assert, an extension operation — emit_sum writes it the cheapest exact way: a short chain
    # of modular operations as before, or all products and leaves into one 64-bit sum (one v_mad_u64_u32 each; a leaf as a product by
    # +-R mod p, a subtracted product with p - x on one side) and ONE reduction (a b - c: three instructions less than a product, a
    # correction and a modular subtraction; a sum of k products: k - 1 reductions and k - 1 modular additions less).
    sums = bool(Q_SUMS) if sums is None else sums
    reads = _base_read_counts(prog) if sums else {}
    bdef = {}                # SSA base value -> {"prods": [(sign, x, y)], "leaves": [(sign, z)]} not emitted yet

    def buse(name):
        """The name of a base value about to be read as a reduced word: a deferred sum is emitted first."""
        f = bdef.pop(name, None)
        if f is not None:
            lines.append(emit_sum(name, f["prods"], f["leaves"], uniform))
            meta.append((name, tuple(z for _, x, y in f["prods"] for z in (x, y)) + tuple(z for _, z in f["leaves"])))
        return name

    def bview(name, sign):
        """A base value as (products, leaves) to be merged into a sum, with `sign` applied. A deferred sum with several readers is
        emitted once if recomputing it in each reader would cost more than reading the word."""
        f = bdef.get(name)
        if f is not None and reads.get(name, 1) > 1 and len(f["prods"]) + len(f["leaves"]) > 2:
            buse(name)
            f = None
        if f is None:
            return [], [(sign, name)]
        if reads.get(name, 1) <= 1:
            del bdef[name]
        return [(sign * sg, x, y) for sg, x, y in f["prods"]], [(sign * sg, z) for sg, z in f["leaves"]]

    cidx = 0
    for k in range(n_instr):
        w0, imm = int(prog[4 + 2 * k]), int(prog[5 + 2 * k])
        op, dst, ra, rb = w0 & 0xFF, (w0 >> 8) & 0xFF, (w0 >> 16) & 0xFF, w0 >> 24
        row = "q.pn" if ra else "q.p"
        if op == air.LD_MAIN:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.main_lde[(size_t){imm} * a.main_stride + {row}];")
            meta.append((v, ()))
        elif op == air.LD_PREP:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.prep_lde[(size_t){imm} * a.prep_stride + {row}];")
            meta.append((v, ()))
        elif op == air.LD_PERM:
            v = fresh("e"); cur_e[dst] = v
            base = f"a.perm_lde + (size_t){4 * imm} * a.perm_stride + {row}"
            lines.append(f"const kb::E4 {v} = kb::E4{{{{({base})[0], ({base})[a.perm_stride], ({base})[2 * a.perm_stride], "
                         f"({base})[3 * a.perm_stride]}}}};")
            meta.append((v, ()))
        elif op == air.LD_CONST:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = {imm}u;")
            meta.append((v, ()))
            uniform.add(v)
        elif op == air.LD_PV:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.public_values[{imm}];")
            meta.append((v, ()))
            uniform.add(v)
        elif op == air.LD_CHALLENGE:
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = {'a.perm_beta' if imm else 'a.perm_alpha'};")
            meta.append((v, ()))
            uniform.add(v)
        elif op == air.LD_LOCAL_SUM:
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = a.local_sum;")
            meta.append((v, ()))
            uniform.add(v)
        elif op == air.LD_GLOBAL_SUM:
            v = fresh("b"); cur_b[dst] = v
            lines.append(f"const uint32_t {v} = a.consts[{imm}];")
            meta.append((v, ()))
            uniform.add(v)
        elif op in (air.LD_IS_FIRST, air.LD_IS_LAST, air.LD_IS_TRANS):
            v = fresh("b"); cur_b[dst] = v
            sel = {air.LD_IS_FIRST: "q.is_first", air.LD_IS_LAST: "q.is_last", air.LD_IS_TRANS: "q.is_trans"}[op]
            lines.append(f"const uint32_t {v} = {sel};")
            meta.append((v, ()))
        elif op in (air.ADD_B, air.SUB_B, air.MUL_B):
            fn = {air.ADD_B: "add", air.SUB_B: "sub", air.MUL_B: "mul"}[op]
            x, y = cur_b[ra], cur_b[rb]
            v = fresh("b"); cur_b[dst] = v
            if sums and not (x in uniform and y in uniform):
                if op == air.MUL_B:
                    bdef[v] = {"prods": [(1, buse(x), buse(y))], "leaves": []}
                else:
                    px, lx = bview(x, 1)
                    py, ly = bview(y, 1 if op == air.ADD_B else -1)
                    bdef[v] = {"prods": px + py, "leaves": lx + ly}
                    if len(px) + len(py) + len(lx) + len(ly) > MAX_SUM_TERMS:
                        buse(v)
                continue
            lines.append(f"const uint32_t {v} = kb::{fn}({x}, {y});")
            meta.append((v, (x, y)))
            if x in uniform and y in uniform:
                uniform.add(v)
        elif op == air.NEG_B:
            x = cur_b[ra]
            v = fresh("b"); cur_b[dst] = v
            if sums and x not in uniform:
                px, lx = bview(x, -1)
                bdef[v] = {"prods": px, "leaves": lx}
                continue
            lines.append(f"const uint32_t {v} = kb::neg({x});")
            meta.append((v, (x,)))
            if x in uniform:
                uniform.add(v)
        elif op in (air.ADD_E, air.SUB_E, air.MUL_E):
            fn = {air.ADD_E: "eadd", air.SUB_E: "esub", air.MUL_E: "emul"}[op]
            x, y = cur_e[ra], cur_e[rb]
            if op == air.ADD_E and (x in deferred or y in deferred) and x != y:
                # a sum with a linear form stays a linear form: the other side joins as more terms, as a uniform constant or as an extra addend
                form = {"terms": [], "vterms": [], "consts": [], "extras": []}
                for z in (x, y):
                    if z in deferred:
                        # copied, not taken: z stays a deferred form of its own, so a second reader of the same value (the same
                        # beta * column product added into two different sums) still finds it and `use` can emit it
                        f2 = deferred[z]
                        for key in form:
                            form[key] += list(f2[key])
                    elif z in uniform:
                        form["consts"].append(z)
                    else:
                        form["extras"].append(z)
                v = fresh("e"); cur_e[dst] = v
                deferred[v] = form
                continue
            x, y = use(x), use(y)
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = kb::{fn}({x}, {y});")
            meta.append((v, (x, y)))
            if x in uniform and y in uniform:
                uniform.add(v)
        elif op == air.NEG_E:
            x = use(cur_e[ra])
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = kb::eneg({x});")
            meta.append((v, (x,)))
            if x in uniform:
                uniform.add(v)
        elif op in (air.ADD_EB, air.SUB_EB, air.MUL_EB):
            fn = {air.ADD_EB: "eadd_base", air.SUB_EB: "esub_base", air.MUL_EB: "escale"}[op]
            x, y = cur_e[ra], buse(cur_b[rb])
            if op == air.MUL_EB and x in uniform and x not in deferred and y not in uniform:
                v = fresh("e"); cur_e[dst] = v
                deferred[v] = {"terms": [(x, y)], "vterms": [], "consts": [], "extras": []}      # uniform extension value x row value: a term
                continue
            x = use(x)
            if Q_FORMS and op == air.MUL_EB and x not in uniform and y not in uniform:
                # row extension value x row value (a LogUp batch's multiplicity-weighted denominators, m1 d2 + m2 d1): a term as well,
                # so that the sum of two of them is two products per coefficient and ONE reduction instead of eight Montgomery products
                # and four modular additions
                v = fresh("e"); cur_e[dst] = v
                deferred[v] = {"terms": [], "vterms": [(x, y)], "consts": [], "extras": []}
                continue
            v = fresh("e"); cur_e[dst] = v
            lines.append(f"const kb::E4 {v} = kb::{fn}({x}, {y});")
            meta.append((v, (x, y)))
            if x in uniform and y in uniform:
                uniform.add(v)
        elif op == air.ASSERT_B:
            x = buse(cur_b[ra])
            lines.append(f"kb::fold_base(acc, a.alpha_pows[{cidx}], {x});")
            meta.append((None, (x,)))
            cidx += 1
        elif op == air.ASSERT_E:
            x = use(cur_e[ra])
            lines.append(f"kb::fold_ext(acc, a.alpha_pows[{cidx}], {x});")
            meta.append((None, (x,)))
            cidx += 1
        else:
            raise ValueError(f"bad opcode {op}")
    assert cidx == int(prog[2]) and len(meta) == len(lines)
    if Q_FAKEUNI:
        # EXPERIMENT ONLY (tools/ab_uniforms.sh; wrong values): every derived wave-uniform value read from a table instead of being
        # computed by each wavefront's scalar unit — what a table of precomputed uniform values would cost the kernel
        n_alpha = max(1, int(prog[2]))
        for i, (v, uses) in enumerate(meta):
            if v is not None and v in uniform and uses:
                lines[i] = (f"const kb::E4 {v} = a.alpha_pows[{i % n_alpha}];" if v.startswith("e") else f"const uint32_t {v} = a.public_values[{i % 64}];")
                meta[i] = (v, ())
    return (lines, meta, uniform) if with_uniform else (lines, meta)


def _kernel_source(lines, n_instr, n_constraints, accumulate=False, part="") -> str:
    body = "\n  ".join(lines)
    return f"""// GENERATED by ziren_amd/codegen.py from a chip's constraint bytecode ({n_instr} instructions,
// {n_constraints} constraints{part}). Same arithmetic as stark::quotient_kernel (the interpreter), values in VGPRs.
{"" if Q_TILE == 1 else f"#define ZKM_Q_TILE {Q_TILE}"}
#include "quotient_args.cuh"

extern "C" __global__ {f"__attribute__((amdgpu_waves_per_eu({Q_WAVES},{Q_WAVES}))) " if Q_WAVES else ""}__launch_bounds__({BLOCK}) void {KERNEL_NAME}(stark::QuotientArgs a) {{
  stark::QuotientPoint q;
  if (!stark::quotient_point(a, stark::quotient_row(a), q)) return;
  kb::FoldAcc acc = kb::fold_zero();
  {body}
  stark::{'quotient_accumulate' if accumulate else 'quotient_store'}(a, q, kb::fold_finish(acc));
}}
"""


def _is_load(line: str) -> int:
    """Words a statement loads from the trace LDEs (0: not a load)."""
    if "= a.main_lde[" in line or "= a.prep_lde[" in line or "= a.main[" in line or "= a.prep[" in line:
        return 1
    return 4 if "a.perm_lde + " in line else 0


def prefetch_order(lines, words_per_group: int):
    """The same statements with the trace loads moved up: the loads are cut into groups of about `words_per_group` words in the order the
    program first needs them; group 0 opens the kernel, and group k + 1 is issued where the program reaches the first load of group k, each
    group closed by a scheduling barrier so that the compiler keeps it there. While the arithmetic on one group's columns runs, the next
    group's loads are in flight (left to itself the compiler sinks every load to its first use: about one load in flight per wave)."""
    loads = [k for k, ln in enumerate(lines) if _is_load(ln)]
    groups, cur, words = [], [], 0
    for k in loads:
        cur.append(k)
        words += _is_load(lines[k])
        if words >= words_per_group:
            groups.append(cur)
            cur, words = [], 0
    if cur:
        groups.append(cur)
    if len(groups) < 2:
        return list(lines)
    first_of = {g[0]: gi for gi, g in enumerate(groups)}
    barrier = "__builtin_amdgcn_sched_barrier(0);"
    out = []
    for g in groups[:Q_AHEAD]:
        out += [lines[k] for k in g] + [barrier]
    moved = set(k for g in groups for k in g)
    for k, ln in enumerate(lines):
        gi = first_of.get(k)
        if gi is not None and gi + Q_AHEAD < len(groups):
            out += [lines[j] for j in groups[gi + Q_AHEAD]] + [barrier]
        if k not in moved:
            out.append(ln)
    return out


def _load_key(line: str):
    """(array, column, is_next) of a trace-load statement, None for anything else."""
    import re
    m = re.search(r"= a\.(main|prep)_lde\[\(size_t\)(\d+) \* a\.\w+_stride \+ (q\.pn|q\.p)\];", line)
    if m:
        return m.group(1), int(m.group(2)), m.group(3) == "q.pn"
    m = re.search(r"a\.perm_lde \+ \(size_t\)(\d+) \* a\.perm_stride \+ (q\.pn|q\.p)\)\[0\]", line)
    if m:
        return "perm", int(m.group(1)), m.group(2) == "q.pn"
    return None


def pair_row_loads(lines, meta):
    """The same statements with every column's two row loads next to each other. A column read at `next` as well as at `local` (every
    permutation column: the running-sum constraint adds up the whole next row, permutation.rs eval; a few main columns) is read twice
    through the same cache lines — once by the thread that owns the row, once by the thread that owns the row before it, which the
    quotient tile puts in the same wavefront or block (quotient_args.cuh). That only saves the second trip to HBM if the two reads are
    close in TIME: in program order the `local` reads sit with the LogUp constraints and the `next` reads thousands of instructions
    later with the cumulative-sum constraint, by when the line has left the L2 (round 5: 10.1 GB read per proof for 7.8 GB of rows,
    whatever the tile). So the second load of a pair is moved up behind the first; and so that the hoisted values do not sit in
    registers until the program reaches them (Cpu: 44 permutation words), every statement that only combines hoisted values — the
    running sum of the next row, one extension addition per column — is moved up with them: four registers stay live instead."""
    keys = [_load_key(ln) for ln in lines]
    where = {k: i for i, k in enumerate(keys) if k is not None}
    partner = {}
    for i, k in enumerate(keys):
        if k is None:
            continue
        j = where.get((k[0], k[1], not k[2]))
        if j is not None and j > i:
            partner[i] = j
    if not partner:
        return list(lines), list(meta)
    hoisted_stmt = set(partner.values())
    tainted = set()                 # values that depend on hoisted loads only
    defined = set()
    emitted = [False] * len(lines)
    order = []
    uniform_like = set()            # values with no operands that are not loads (constants, challenges, selectors): free to combine with

    def emit(i):
        emitted[i] = True
        order.append(i)
        v = meta[i][0]
        if v is not None:
            defined.add(v)

    # consumers of each value, in program order
    readers = {}
    for i, (v, uses) in enumerate(meta):
        for u in uses:
            readers.setdefault(u, []).append(i)
    for i, (v, uses) in enumerate(meta):
        if v is not None and not uses and keys[i] is None:
            uniform_like.add(v)

    def close_over(start_values):
        """Statements (not asserts) all of whose operands are hoisted-derived or operand-free values already defined: move them up now."""
        work = list(start_values)
        while work:
            x = work.pop()
            for r in readers.get(x, ()):
                v, uses = meta[r]
                if emitted[r] or v is None:
                    continue
                if all((u in tainted) or (u in uniform_like and u in defined) for u in uses) and any(u in tainted for u in uses):
                    emit(r)
                    tainted.add(v)
                    work.append(v)

    for i in range(len(lines)):
        if emitted[i]:
            continue
        emit(i)
        j = partner.get(i)
        if j is not None and not emitted[j]:
            emit(j)
            tainted.add(meta[j][0])
            close_over([meta[j][0]])
    assert len(order) == len(lines)
    return [lines[i] for i in order], [meta[i] for i in order]


UNIFORM_TABLE_WORDS = 4096     # stark::QUOTIENT_UNIFORM_WORDS (csrc/quotient_args.cuh): what the library allocates per chip


def split_uniform(lines, meta, uniform):
    """The statements of a kernel without its wave-uniform arithmetic. A value that is the same for every row — the powers of beta, a
    constant times a challenge, anything computed from challenges, constants and public values only — used to be computed by every
    wavefront on its scalar unit: for the Cpu chip 4 400 scalar instructions beside 6 500 vector ones, 14 KiB of a 78 KiB kernel (the
    instruction cache holds 64), and the wavefront issues nothing else meanwhile. Such values are now computed ONCE per launch by a
    one-wavefront kernel (`zkm_quotient_uniforms`, launched in front) into a table, and the kernel proper reads the ones it needs
    with scalar loads. Returns (main lines, main meta, prologue lines) or None when there is nothing to move or the table would not fit.
    Leaves (a literal, a challenge, a public value) stay where they are read: they cost a move or one scalar load."""
    import re
    name_re = re.compile(r"\b[eb]\d+\b")
    defined_at = {v: i for i, (v, _) in enumerate(meta) if v is not None}
    # equal uniform expressions get one representative (the bytecode builds the powers of beta again for every lookup): one slot, one
    # computation in the table kernel
    rep, by_expr, rep_line = {}, {}, {}
    for ln, (v, uses) in zip(lines, meta):
        if v is None or v not in uniform:
            continue
        head, expr = ln.split(" = ", 1)
        expr = name_re.sub(lambda m: rep.get(m.group(0), m.group(0)), expr)
        key = (v[0], expr)
        if key in by_expr:
            rep[v] = by_expr[key]
        else:
            by_expr[key] = rep[v] = v
            rep_line[v] = f"{head} = {expr}"
    derived = {v for v, uses in meta if v is not None and v in uniform and uses}
    frontier, seen = [], set()
    for v, uses in meta:
        if v is not None and v in uniform:
            continue
        for u in uses:
            if u in derived and rep[u] not in seen:
                seen.add(rep[u])
                frontier.append(rep[u])
    if not frontier:
        return None
    offsets, words = {}, 0
    for v in [u for u in frontier if u.startswith("e")] + [u for u in frontier if not u.startswith("e")]:   # extension values first: 16-byte slots stay aligned
        offsets[v] = words
        words += 4 if v.startswith("e") else 1
    if words > UNIFORM_TABLE_WORDS:
        return None
    need, stack = set(), list(frontier)
    while stack:
        u = stack.pop()
        if u in need:
            continue
        need.add(u)
        stack += [rep[w] for w in meta[defined_at[u]][1]]
    prologue = [rep_line[u] for u in sorted(need, key=lambda u: defined_at[u])]
    prologue.append("if (threadIdx.x == 0) { " + " ".join(
        (f"*(kb::E4*)(a.uniforms + {off}) = {v};" if v.startswith("e") else f"a.uniforms[{off}] = {v};") for v, off in offsets.items()) + " }")
    main_lines, main_meta = [], []
    for ln, (v, uses) in zip(lines, meta):
        if v is None or v not in uniform or not uses:
            main_lines.append(ln); main_meta.append((v, uses))           # row arithmetic, asserts, uniform leaves
        elif rep[v] in offsets:
            off = offsets[rep[v]]
            main_lines.append(f"const kb::E4 {v} = *(const kb::E4*)(a.uniforms + {off});" if v.startswith("e") else f"const uint32_t {v} = a.uniforms[{off}];")
            main_meta.append((v, ()))
    return main_lines, main_meta, prologue


def _uniforms_kernel_source(prologue) -> str:
    body = "\n  ".join(prologue)
    return f"""
// The chip's wave-uniform values, once per launch (split_uniform): one wavefront, every lane the same arithmetic, lane 0 stores.
extern "C" __global__ __launch_bounds__(64) void {UNIFORMS_KERNEL_NAME}(stark::QuotientArgs a) {{
  {body}
}}
"""


def emit_source(program: np.ndarray) -> str:
    """Straight-line HIP for one chip: one kernel, and in front of it (where the chip has wave-uniform arithmetic) the one-wavefront
    kernel that fills its table of uniform values."""
    prog = np.asarray(program, dtype=np.uint32)
    lines, meta, uniform = _ssa_lines(prog, with_uniform=True)
    prologue = None
    if Q_UNITABLE:
        cut = split_uniform(lines, meta, uniform)
        if cut is not None:
            lines, meta, prologue = cut
    if Q_PAIR:
        lines, _ = pair_row_loads(lines, meta)
    if Q_PREFETCH:
        lines = prefetch_order(lines, Q_PREFETCH)
    src = _kernel_source(lines, int(prog[0]), int(prog[2]))
    if prologue is not None:
        src += _uniforms_kernel_source(prologue)
    return src




def emit_part_sources(program: np.ndarray):
    """A long program as several kernels. The folded constraint sum is linear in the constraints, so the program is cut at assert
    boundaries: the first kernel stores its partial quotient, the others add theirs to it. A value computed before a cut and used after
    it is recomputed by the later kernel (the statements that define it, transitively, are put in front of the part)."""
    prog = np.asarray(program, dtype=np.uint32)
    # One modular operation per bytecode instruction here, and no table of uniform values: a value computed before a cut and read after it
    # is recomputed by the later part, and a deferred sum of products (emit_sum) read in several parts is recomputed whole in each —
    # KeccakSponge's quotient: 3.6 ms per shard this way, 5.9 ms with the sums (its XOR / AND terms are sums of products read again and
    # again); the table kernel costs it 0.2 ms for a few scalar instructions saved (tools/ab_keccak_quotient.sh, EXPERIMENTS.md round 5).
    lines, meta, uniform = _ssa_lines(prog, with_uniform=True, sums=False)
    prologue = None
    if Q_UNITABLE and Q_PARTS_TABLE:
        cut = split_uniform(lines, meta, uniform)
        if cut is not None:
            lines, meta, prologue = cut
    if Q_PAIR:
        lines, meta = pair_row_loads(lines, meta)
    defined_at = {v: k for k, (v, _) in enumerate(meta) if v is not None}
    cuts, start = [], 0
    for k, (v, _) in enumerate(meta):
        if v is None and k + 1 - start >= PART_INSTRS:
            cuts.append((start, k + 1))
            start = k + 1
    if start < len(lines):
        if cuts and not any(v is None for v, _ in meta[start:]):
            cuts[-1] = (cuts[-1][0], len(lines))      # trailing statements without an assert belong to the last part
        else:
            cuts.append((start, len(lines)))
    sources = []
    for n, (lo, hi) in enumerate(cuts):
        need, stack = set(), [u for _, uses in meta[lo:hi] for u in uses if defined_at[u] < lo]
        while stack:
            u = stack.pop()
            if u in need:
                continue
            need.add(u)
            stack += list(meta[defined_at[u]][1])
        prelude = [lines[k] for k in sorted(defined_at[u] for u in need)]
        sources.append(_kernel_source(prelude + lines[lo:hi], int(prog[0]), int(prog[2]), accumulate=n > 0,
                                      part=f"; part {n + 1} of {len(cuts)}: statements {lo}..{hi - 1}"))
    if prologue is not None:
        sources[0] += _uniforms_kernel_source(prologue)
    return sources


MAX_SPECIALIZED_INSTRS = 1 << 20   # beyond this the interpreter evaluates the program (no recorded chip comes near)
PARTS_MAGIC = b"ZKMQPART"         # container of a multi-kernel program: magic, u32 count, u32 pad, count x u64 lengths, the code objects


def _private_tmp(out: str) -> str:
    """A temporary name next to `out` that no other process or thread uses (os.replace publishes the finished file atomically)."""
    return f"{out}.{os.getpid()}.{__import__('threading').get_ident()}.tmp"


def _compile(src: str, out: str, verbose: bool = False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, os.path.basename(out) + ".hip")
        with open(path, "w") as f:
            f.write(src)
        tmp = _private_tmp(out)   # farm ranks (one process per GPU) may compile the same chip on a cold cache: no shared temporary
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", "-I", CSRC, path, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    os.replace(tmp, out)


def specialize(program: np.ndarray, force: bool = False, verbose: bool = False) -> Optional[bytes]:
    """Return the gfx950 code object for `program`, compiling it on first use (cached in-tree). A program of more than
    SINGLE_KERNEL_INSTRS instructions comes back as a container of several code objects (emit_part_sources; hipcc's time grows much
    faster than the length of a straight-line kernel: 114 324 instructions in one kernel do not finish in half an hour, in nineteen
    kernels they take seconds each), which zkm_ctx_register_quotient_kernel takes just the same. None for a program beyond
    MAX_SPECIALIZED_INSTRS — the library's bytecode interpreter evaluates it."""
    n_instr = int(np.asarray(program)[0])
    if n_instr > MAX_SPECIALIZED_INSTRS:
        return None
    os.makedirs(CACHE, exist_ok=True)
    h = program_hash(program)
    if n_instr <= SINGLE_KERNEL_INSTRS:
        out = os.path.join(CACHE, f"q_{h}.hsaco")
        if force or not os.path.exists(out):
            _compile(emit_source(program), out, verbose)
    else:
        out = os.path.join(CACHE, f"q_{h}.parts")
        if force or not os.path.exists(out):
            from concurrent.futures import ThreadPoolExecutor
            sources = emit_part_sources(program)
            with tempfile.TemporaryDirectory() as td:
                outs = [os.path.join(td, f"q_{h}_{n}.hsaco") for n in range(len(sources))]
                with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as pool:
                    list(pool.map(lambda so: _compile(so[0], so[1], verbose), zip(sources, outs)))
                blobs = [open(o, "rb").read() for o in outs]
            blob = PARTS_MAGIC + np.array([len(blobs), 0], dtype="<u4").tobytes() + np.array([len(b) for b in blobs], dtype="<u8").tobytes()
            blob += b"".join(blobs)
            tmp = _private_tmp(out)
            with open(tmp, "wb") as f:
                f.write(blob)
            os.replace(tmp, out)
    _note_in_manifest(program, os.path.basename(out))
    with open(out, "rb") as f:
        return f.read()


def specialize_many(programs, verbose: bool = False):
    """specialize() for a list of programs, compiling the missing ones side by side (the build step: every recorded chip)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as pool:
        return list(pool.map(lambda prog: specialize(prog, verbose=verbose), programs))


# ---- permutation-trace kernels -----------------------------------------------------------------------------------------------------------
PERM_KERNEL_NAME = "zkm_perm_rows_specialized"
MAX_PERM_LOOKUPS = 64      # beyond this (the precompiles: ShaCompress 115 lookups, KeccakSponge 357, EdAddAssign 881) the straight-line kernel is
                           # megabytes of code; those chips keep the generic kernel, which walks the blob


def parse_lookups(blob):
    """The `lookups` blob of zkm_chip_desc (air.encode_lookups) back into (n_sends, [(kind, [value forms], multiplicity form)]) with a form =
    (constant word, [(is_main, column, weight word)])."""
    b = [int(x) for x in np.asarray(blob, dtype=np.uint32)]
    n_sends, n_receives = b[0], b[1]
    pos, out = 2, []

    def form():
        nonlocal pos
        nt, const = b[pos], b[pos + 1]
        pos += 2
        terms = []
        for _ in range(nt):
            cw, weight = b[pos], b[pos + 1]
            pos += 2
            terms.append((cw >> 31, cw & 0x7FFFFFFF, weight))
        return const, terms

    for _ in range(n_sends + n_receives):
        kind, nv = b[pos], b[pos + 1]
        pos += 2
        values = [form() for _ in range(nv)]
        out.append((kind, values, form()))
    assert pos == len(b), "trailing words in the lookups blob"
    return n_sends, out


def emit_perm_source(blob, log_quotient_degree: int) -> str:
    """stark::perm_rows (csrc/stark.cuh) for ONE chip as straight-line HIP: the generic kernel walks the blob with scalar instructions
    (as many as its vector ones), reloads a column every time a lookup names it, and has about one load in flight per wave. Here the
    walk is done at generation time, every column is loaded once per row, weights of one and zero constants cost nothing, and the
    loads are issued in groups ahead of their use (prefetch_order). Same values: every operation is exact field arithmetic with
    canonical results, so the order of additions and the batching of the inversions do not show. A row's multiplicities come first; a
    wavefront whose 64 rows all have none (padding rows) writes zeros and leaves."""
    n_sends, lookups = parse_lookups(blob)
    batch = 1 << log_quotient_degree
    n = len(lookups)
    ncols = -(-n // batch)                      # fraction columns; the running-sum column follows
    ONE = 0x01FFFFFE
    pre, body = [], []                          # statements before / after the all-padding exit
    loaded = {}

    def col(is_main, c, lines):
        key = (is_main, c)
        if key not in loaded:
            name = f"{'m' if is_main else 'p'}{c}"
            lines.append(f"const uint32_t {name} = a.{'main' if is_main else 'prep'}[(size_t){c} * a.n + r];")
            loaded[key] = name
        return loaded[key]

    tmp = [0]

    def form(f, lines):
        const, terms = f
        acc = None if const == 0 else f"{const}u"
        for is_main, c, w in terms:
            v = col(is_main, c, lines)
            t = v if w == ONE else f"kb::mul({v}, {w}u)"
            acc = t if acc is None else f"kb::add({acc}, {t})"
        if acc is None:
            return "0u"
        tmp[0] += 1
        name = f"f{tmp[0]}"
        lines.append(f"const uint32_t {name} = {acc};")
        return name

    mults = []
    for k, (kind, values, mult) in enumerate(lookups):
        m = form(mult, pre)
        if k >= n_sends and m != "0u":
            tmp[0] += 1
            pre.append(f"const uint32_t f{tmp[0]} = kb::neg({m});")
            m = f"f{tmp[0]}"
        mults.append(m)
    live = [m for m in mults if m != "0u"]
    zero_stores = " ".join(f"a.perm[(size_t){j} * a.n + r] = 0;" for j in range(4 * (ncols + 1)))
    exit_line = f"if (__all(({' | '.join(live) if live else '0u'}) == 0)) {{ {zero_stores} return; }}"
    body.append("kb::E4 rowsum = kb::ezero();")
    G = P_GROUP
    for b0 in range(0, ncols, G):
        cols_here = [b for b in range(b0, min(b0 + G, ncols))]
        for g, b in enumerate(cols_here):
            first = True
            for k in range(b * batch, min((b + 1) * batch, n)):
                kind, values, _ = lookups[k]
                if Q_FORMS:
                    # alpha + kind + sum_v beta^(v+1) value_v as one bounded form: the accumulators start at the (uniform) constant
                    terms = []
                    for v, f in enumerate(values):
                        lin = form(f, body)
                        if lin != "0u":
                            terms.append((f"a.beta_pows[{v + 1}]", lin))
                    body.append(emit_form(f"d{k}", [f"kb::eadd_base(a.alpha, kb::to_monty({kind}u))"], terms, [], []))
                else:
                    body.append(f"kb::E4 d{k} = kb::eadd_base(a.alpha, kb::to_monty({kind}u));")
                    if len(values) >= 4:
                        body.append(f"kb::FoldAcc fa{k} = kb::fold_zero();")
                        for v, f in enumerate(values):
                            lin = form(f, body)
                            if lin != "0u":
                                body.append(f"kb::fold_base(fa{k}, a.beta_pows[{v + 1}], {lin});")
                        body.append(f"d{k} = kb::eadd(d{k}, kb::fold_finish(fa{k}));")
                    else:
                        for v, f in enumerate(values):
                            lin = form(f, body)
                            if lin != "0u":
                                body.append(f"d{k} = kb::eadd(d{k}, kb::escale(a.beta_pows[{v + 1}], {lin}));")
                if first:
                    body.append(f"kb::E4 num{b} = kb::efrom({mults[k]}); kb::E4 den{b} = d{k};")
                    first = False
                else:
                    body.append(f"num{b} = kb::eadd(kb::emul(num{b}, d{k}), kb::escale(den{b}, {mults[k]})); den{b} = kb::emul(den{b}, d{k});")
        names = [f"den{b}" for b in cols_here]
        body.append(f"uint32_t x{b0}[{len(names)}], y{b0}[{len(names)}], z{b0}[{len(names)}];")
        for g, b in enumerate(cols_here):
            body.append(f"z{b0}[{g}] = kb::einv_norm(den{b}, x{b0}[{g}], y{b0}[{g}]);")
        body.append(f"kb::inv_batch<{len(names)}>(z{b0});")
        for g, b in enumerate(cols_here):
            body.append(f"{{ const kb::E4 val = kb::emul(num{b}, kb::einv_finish(den{b}, x{b0}[{g}], y{b0}[{g}], z{b0}[{g}])); "
                        + " ".join(f"a.perm[(size_t){4 * b + e} * a.n + r] = val.c[{e}];" for e in range(4)) + " rowsum = kb::eadd(rowsum, val); }")
    body.append(" ".join(f"a.perm[(size_t){4 * ncols + e} * a.n + r] = rowsum.c[{e}];" for e in range(4)))
    if Q_PREFETCH:
        body = prefetch_order(body, Q_PREFETCH)
    text = "\n  ".join(pre + [exit_line] + body)
    return f"""// GENERATED by ziren_amd/codegen.py from a chip's lookups blob ({n} lookups, {n_sends} sends, batches of {batch}). Same values as
// stark::perm_rows.
#include "perm_args.cuh"

extern "C" __global__ __launch_bounds__({BLOCK}) void {PERM_KERNEL_NAME}(stark::PermArgs a) {{
  const size_t r = (size_t)blockIdx.x * {BLOCK} + threadIdx.x;
  if (r >= a.n) return;
  {text}
}}
"""


def perm_hash(blob, log_quotient_degree: int) -> str:
    h = hashlib.sha256(_template_key() + b"perm2")
    with open(os.path.join(CSRC, "perm_args.cuh"), "rb") as f:
        h.update(f.read())
    h.update(np.ascontiguousarray(blob, dtype=np.uint32).tobytes() + bytes([log_quotient_degree]))
    return h.hexdigest()[:24]


def specialize_perm(blob, log_quotient_degree: int, force: bool = False, verbose: bool = False) -> Optional[bytes]:
    """The gfx950 code object of a chip's permutation-trace kernel (compiled on first use, cached in-tree like the quotient kernels);
    None for a chip without lookups (nothing to specialise)."""
    blob = np.ascontiguousarray(blob, dtype=np.uint32)
    if len(blob) < 2 or int(blob[0]) + int(blob[1]) == 0 or int(blob[0]) + int(blob[1]) > MAX_PERM_LOOKUPS:
        return None
    os.makedirs(CACHE, exist_ok=True)
    out = os.path.join(CACHE, f"p_{perm_hash(blob, log_quotient_degree)}.hsaco")
    if force or not os.path.exists(out):
        _compile(emit_perm_source(blob, log_quotient_degree), out, verbose)
    key = hashlib.sha256(np.ascontiguousarray(blob, dtype="<u4").tobytes() + bytes([log_quotient_degree])).hexdigest()
    import fcntl
    path = os.path.join(CACHE, "manifest.json")
    with _manifest_lock, open(path + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        _update_manifest(path, "perm:" + key, os.path.basename(out))
    with open(out, "rb") as f:
        return f.read()


def specialize_perm_many(chips, verbose: bool = False):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as pool:
        return list(pool.map(lambda c: specialize_perm(c.lookups_blob, c.log_quotient_degree, verbose=verbose), chips))


def prune_cache():
    """Remove code objects the manifest no longer names (left behind when the template key changes: every kernel gets a new file name).
    Called at the end of `__graft_entry__.build()`; the cache travels to the GPU box with the snapshot."""
    import json
    try:
        with open(os.path.join(CACHE, "manifest.json")) as f:
            keep = set(json.load(f).values()) | {"manifest.json", "manifest.json.lock"}
    except (OSError, ValueError):
        return 0
    stale = [f for f in os.listdir(CACHE) if f not in keep and not f.endswith(".tmp")]
    for f in stale:
        os.remove(os.path.join(CACHE, f))
    return len(stale)


_manifest_lock = __import__("threading").Lock()


def _note_in_manifest(program: np.ndarray, filename: str):
    """Ahead-of-time story for a host with no compiler at run time (the Rust shim): `_jit/manifest.json` maps the sha256
    of a chip's program words (little-endian bytes — a key any host language can compute) to its code object. `__graft_entry__.build()`
    specialises every recorded chip, so the manifest and the code objects ship with the build; an unknown program falls back to the
    bytecode interpreter inside the library."""
    import json
    key = hashlib.sha256(np.ascontiguousarray(program, dtype="<u4").tobytes()).hexdigest()
    path = os.path.join(CACHE, "manifest.json")
    import fcntl
    with _manifest_lock, open(path + ".lock", "w") as lk:   # threads of this process, then the other processes (farm ranks)
        fcntl.flock(lk, fcntl.LOCK_EX)
        _update_manifest(path, key, filename)


def _update_manifest(path, key, filename):
    import json
    try:
        with open(path) as f:
            m = json.load(f)
    except (OSError, ValueError):
        m = {}
    if m.get(key) != filename:
        m[key] = filename
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(m, f, indent=0, sort_keys=True)
        os.replace(tmp, path)
