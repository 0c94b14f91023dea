"""Synthetic shards shaped like Ziren's MIPS core shards ("SYN-k").

The reference's executor and its 50 chip AIRs are Rust (crates/core/*) and cannot run here,
so workloads are shape-driven, like the reference's own dummy-record test
(crates/core/machine/src/shape/mod.rs:647-725): each synthetic chip has the height, main
width, permutation width and quotient degree of a real chip (per-row cost pinned in
crates/core/executor/src/artifacts/mips_costs.json), LogUp sends/receives expressed as
`VirtualPairCol` linear forms, degree-3 constraints touching local and next rows, boundary
constraints against public values, and a *valid* witness, so the restated verifier accepts
the proof. The constraint contents are stand-ins for the real AIRs and are flagged as such in
DESIGN.md.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import air
from . import field as F

NUM_PV_ELTS = 176        # ZKM_PROOF_NUM_PV_ELTS, crates/stark/src/air/public_values.rs:11
PROOF_MAX_NUM_PVS = 231  # crates/stark/src/types.rs:73

# name, log2 height relative to k, main width, permutation base width (= 4 * ext cols), lqd
SYN_CHIPS = [
    ("Cpu", 0, 67, 44),
    ("AddSub", -1, 31, 8),
    ("MemoryInstrs", -2, 79, 28),
    ("Branch", -2, 54, 28),
    ("Lt", -3, 32, 12),
    ("DivRem", -3, 118, 36),
    ("MemoryLocal", -5, 64, 28),
    ("Global", -4, 75, 32),
]


@dataclass
class SynChip:
    name: str
    log_height: int
    main_width: int
    prep_width: int
    prep_index: int
    log_quotient_degree: int
    local_only: bool
    commit_scope_global: bool
    sends: List[air.Lookup]
    receives: List[air.Lookup]
    program: np.ndarray
    lookups_blob: np.ndarray
    num_constraints: int
    trace: Optional[np.ndarray] = None        # (n, main_width) uint32 Montgomery, row-major
    prep_trace: Optional[np.ndarray] = None   # (n, prep_width) uint32 Montgomery

    @property
    def perm_ext_width(self):
        return air.local_permutation_trace_width(len(self.sends) + len(self.receives),
                                                 1 << self.log_quotient_degree)


def _build_chip(name, log_height, main_width, n_lookups, chip_index, seed, prep_width=0, prep_index=-1,
                global_scope=False, with_trace=True, local_only=False, lqd=1):
    """Lay out columns, record lookups + constraints, and (optionally) generate a valid trace.
    lqd = log2 of the quotient degree: 1 for the core machine (degree-3 constraints); the recursion
    machines go higher (chip.rs:81-89), which also widens the LogUp batches to 2^lqd lookups."""
    n = 1 << log_height
    batch = 1 << lqd
    rng = F.SplitMix64(seed)
    cols = {}       # main column -> uint64 canonical array
    pcols = {}      # preprocessed column -> array
    next_col = [0]
    next_pcol = [0]

    def new_col(values=None):
        c = next_col[0]
        next_col[0] += 1
        if with_trace:
            cols[c] = rng.uniform_field(n) if values is None else values
        return c

    def new_pcol(values=None):
        c = next_pcol[0]
        next_pcol[0] += 1
        if with_trace:
            pcols[c] = rng.uniform_field(n) if values is None else values
        return c

    budget = main_width - (14 if global_scope else 0)
    n_pairs = n_lookups // 2
    odd = n_lookups % 2
    sends, receives = [], []
    kinds = [air.KIND_MEMORY, air.KIND_BYTE, air.KIND_RANGE, air.KIND_INSTRUCTION, air.KIND_PROGRAM,
             air.KIND_SYSCALL]

    def roll(a):
        return np.roll(a, -1)

    # --- lookup column pairs: a send over source columns, its receive over the columns rolled by one row
    for k in range(n_pairs):
        nvals = 1 + (k % 2)
        col_mult = (k % 4 == 3)
        need = 2 * nvals + (2 if col_mult else 0)
        # keep at least 8 columns for arithmetic constraints
        if budget - next_col[0] - need < 8 + 2 * (n_pairs - k - 1):
            nvals, col_mult, need = 1, False, 2
        svals, rvals = [], []
        for j in range(nvals):
            use_prep = prep_width and (next_pcol[0] + 2 <= prep_width) and j == 0
            if use_prep:
                s = new_pcol()
                r = new_pcol(roll(pcols[s]) if with_trace else None)
                w = 1 + ((k + j) % 3)
                svals.append(air.VirtualPairCol([(False, s, w)], (k * 7 + j) % 5))
                rvals.append(air.VirtualPairCol([(False, r, w)], (k * 7 + j) % 5))
            else:
                s = new_col()
                r = new_col(roll(cols[s]) if with_trace else None)
                w = 1 + ((k + j) % 3)
                svals.append(air.VirtualPairCol([(True, s, w)], (k * 7 + j) % 5))
                rvals.append(air.VirtualPairCol([(True, r, w)], (k * 7 + j) % 5))
        if col_mult:
            small = (rng.uniform_field(n) % np.uint64(4)) if with_trace else None
            ms = new_col(small)
            mr = new_col(roll(small) if with_trace else None)
            smult, rmult = air.VirtualPairCol.single_main(ms), air.VirtualPairCol.single_main(mr)
        else:
            smult = rmult = air.VirtualPairCol.const(1 + (k % 2))
        kind = kinds[k % len(kinds)]
        sends.append(air.Lookup(svals, smult, kind))
        receives.append(air.Lookup(rvals, rmult, kind))
    if odd:  # an extra send with multiplicity 0 keeps the chip's cumulative sum at zero
        c = new_col()
        sends.append(air.Lookup([air.VirtualPairCol.single_main(c)], air.VirtualPairCol.const(0), air.KIND_RANGE))
    # any preprocessed columns not consumed by lookups stay as free columns
    while next_pcol[0] < prep_width:
        new_pcol()

    perm_w = air.local_permutation_trace_width(len(sends) + len(receives), batch)
    b = air.AirBuilder(main_width, prep_width, perm_w)
    local, nxt = b.main()
    plocal, _ = b.preprocessed()

    # --- arithmetic constraints over the remaining columns
    pv0, pv1 = 2 * chip_index, 2 * chip_index + 1
    pv_vals = {}
    first_acc = True
    while budget - next_col[0] >= 3:
        remaining = budget - next_col[0]
        if first_acc and remaining >= 3 and not local_only:
            # accumulator: acc' = acc + a*b on transitions, pinned to public values at both ends
            a, c2 = new_col(), new_col()
            start = int(rng.uniform_field(1)[0]) if with_trace else 0
            if with_trace:
                prod = F.mul(cols[a], cols[c2])
                csum = np.cumsum(prod, dtype=np.uint64)
                accv = np.empty(n, dtype=np.uint64)
                accv[0] = start
                accv[1:] = (csum[:-1] + np.uint64(start)) % np.uint64(F.P)
                acc = new_col(accv)
                pv_vals[pv0] = start
                pv_vals[pv1] = int(accv[-1])
            else:
                acc = new_col()
            b.when_first_row().assert_eq(local[acc], b.public_values(pv0))
            b.when_transition().assert_eq(nxt[acc], local[acc] + local[a] * local[c2])
            b.when_last_row().assert_eq(local[acc], b.public_values(pv1))
            first_acc = False
        elif remaining >= 4 and (next_col[0] % 3 == 0):
            # degree-3 product d = a*b*c (degree 2^lqd + 1 for lqd > 1: d = a^(2^lqd - 1) * b * c)
            a, c2, c3 = new_col(), new_col(), new_col()
            ea = (1 << lqd) - 1
            if with_trace:
                pa = cols[a]
                for _ in range(ea - 1):
                    pa = F.mul(pa, cols[a])
                d = new_col(F.mul(F.mul(pa, cols[c2]), cols[c3]))
            else:
                d = new_col()
            expr = local[a]
            for _ in range(ea - 1):
                expr = expr * local[a]
            b.assert_eq(expr * local[c2] * local[c3], local[d])
        elif remaining >= 4 and (next_col[0] % 3 == 1):
            # boolean selector and a select: e = s*a + (1-s)*c
            sv = (rng.uniform_field(n) & np.uint64(1)) if with_trace else None
            s = new_col(sv)
            a, c2 = new_col(), new_col()
            e = new_col(np.where(sv == 1, cols[a], cols[c2]) if with_trace else None)
            b.assert_zero(local[s] * (local[s] - 1))
            b.assert_eq(local[s] * (local[a] - local[c2]) + local[c2], local[e])
        else:
            # c = a*b + 3 (constants exercise LD_CONST) and a transition-tied copy
            a, c2 = new_col(), new_col()
            c3 = new_col(F.add(F.mul(cols[a], cols[c2]), 3) if with_trace else None)
            b.assert_eq(local[a] * local[c2] + 3, local[c3])
    # leftover columns (< 3) and the global digest columns are unconstrained
    while next_col[0] < main_width:
        new_col()
    if prep_width:
        # tie one main column to a preprocessed one so LD_PREP is exercised beyond lookups
        pass

    air.eval_permutation_constraints(b, sends, receives, batch, global_scope)
    program = b.assemble()
    chip = SynChip(name=name, log_height=log_height, main_width=main_width, prep_width=prep_width,
                   prep_index=prep_index, log_quotient_degree=lqd, local_only=local_only,
                   commit_scope_global=global_scope, sends=sends, receives=receives, program=program,
                   lookups_blob=air.encode_lookups(sends, receives), num_constraints=int(program[2]))
    if with_trace:
        mat = np.empty((n, main_width), dtype=np.uint32)
        for c in range(main_width):
            mat[:, c] = F.to_monty(cols[c])
        chip.trace = mat
        if prep_width:
            pm = np.empty((n, prep_width), dtype=np.uint32)
            for c in range(prep_width):
                pm[:, c] = F.to_monty(pcols[c])
            chip.prep_trace = pm
    return chip, pv_vals


@dataclass
class SynShard:
    chips: List[SynChip]
    public_values: np.ndarray   # (PROOF_MAX_NUM_PVS,) uint32 Montgomery
    pc_start: int               # Montgomery
    initial_global_cumulative_sum: np.ndarray  # (14,) Montgomery


def syn_shard(k: int, with_prep: bool = False, with_trace: bool = True, seed: int = 0x5A4B4D00,
              chips=None) -> SynShard:
    """SYN-k: Cpu-like chip at 2^k rows plus seven smaller chips.

    with_prep adds a Byte-like chip (2^min(16,k) rows, 16 preprocessed columns) so the opening
    has the reference's 4 rounds (preprocessed, main, permutation, quotient).
    """
    spec = chips if chips is not None else SYN_CHIPS
    out, pvs = [], {}
    for i, (name, dlog, m, p) in enumerate(spec):
        lh = max(k + dlog, 1)
        n_lookups = p // 2 - 2
        chip, pv = _build_chip(name, lh, m, n_lookups, i, seed + i, global_scope=(name == "Global"),
                               with_trace=with_trace)
        out.append(chip)
        pvs.update(pv)
    if with_prep:
        i = len(out)
        chip, pv = _build_chip("Byte", min(16, max(k - 1, 1)), 30, 6, i, seed + i, prep_width=16, prep_index=0,
                               with_trace=with_trace)
        out.append(chip)
        pvs.update(pv)
    pv_arr = np.zeros(PROOF_MAX_NUM_PVS, dtype=np.uint64)
    rng = F.SplitMix64(seed ^ 0xABCDEF)
    pv_arr[:NUM_PV_ELTS] = rng.uniform_field(NUM_PV_ELTS)
    for idx, v in pvs.items():
        pv_arr[idx] = v
    igcs = F.to_monty(rng.uniform_field(14))
    return SynShard(chips=out, public_values=F.to_monty(pv_arr), pc_start=F.to_monty(0x400000),
                    initial_global_cumulative_sum=igcs)


def edge_shard(k: int, seed: int = 0xED6E, lqd: int = 1) -> SynShard:
    """Shapes the reference's machine has but SYN-k lacks: a `local_only` chip (opened at zeta only,
    prover.rs:526-544), a chip without lookups (empty permutation trace, zero local sum), a global-scope
    chip, a preprocessed chip, and two chips of equal height (ordering by name, prover.rs:264)."""
    out, pvs = [], {}
    specs = [
        dict(name="Plain", lh=k, m=20, n_lookups=4),
        dict(name="LocalOnly", lh=k, m=18, n_lookups=2, local_only=True),
        dict(name="NoLookups", lh=max(k - 1, 1), m=9, n_lookups=0),
        dict(name="Global", lh=max(k - 2, 1), m=30, n_lookups=3, global_scope=True),
        dict(name="Byte", lh=max(k - 1, 1), m=12, n_lookups=4, prep_width=6, prep_index=0, local_only=True),
    ]
    if lqd > 1:  # recursion-style machine: one quotient degree for all chips, wider LogUp batches
        specs = [dict(sp, n_lookups=sp["n_lookups"] + (3 if sp["n_lookups"] else 0)) for sp in specs]
    for i, sp in enumerate(specs):
        chip, pv = _build_chip(sp["name"], sp["lh"], sp["m"], sp["n_lookups"], i, seed + i,
                               prep_width=sp.get("prep_width", 0), prep_index=sp.get("prep_index", -1),
                               global_scope=sp.get("global_scope", False), local_only=sp.get("local_only", False), lqd=lqd)
        out.append(chip)
        pvs.update(pv)
    pv_arr = np.zeros(PROOF_MAX_NUM_PVS, dtype=np.uint64)
    rng = F.SplitMix64(seed ^ 0x1234)
    pv_arr[:NUM_PV_ELTS] = rng.uniform_field(NUM_PV_ELTS)
    for idx, v in pvs.items():
        pv_arr[idx] = v
    return SynShard(chips=out, public_values=F.to_monty(pv_arr), pc_start=F.to_monty(0x400000),
                    initial_global_cumulative_sum=F.to_monty(rng.uniform_field(14)))


def shard_algorithmic_bytes(shard: SynShard) -> int:
    """Compulsory HBM bytes of commit+open, SURVEY.md §8d: sum over chips of
    n * (36 m + 36 p + 24 r + 12 q + 672)."""
    total = 0
    for c in shard.chips:
        n = 1 << c.log_height
        p = 4 * c.perm_ext_width
        q = 4 << c.log_quotient_degree
        total += n * (36 * c.main_width + 36 * p + 24 * c.prep_width + 12 * q + 672)
    return total


def shard_poseidon2_permutations(shard: SynShard, log_blowup: int = 1) -> int:
    """Poseidon2 permutations of commit+open (SURVEY.md §8d op counts): per commitment, every LDE row of a height is
    one sponge over the concatenated widths of that height's matrices (⌈w/8⌉ permutations), every shorter height costs
    one more compression per row where it is injected, and the tree is H_max − 1 compressions; the FRI commit phase
    hashes len/2 two-element leaves and builds a tree per layer. Proof-of-work (2^16 expected) and queries are left out."""
    def commit(mats):   # mats: (lde height, width)
        by_h = {}
        for h, w in mats:
            by_h[h] = by_h.get(h, 0) + w
        hmax = max(by_h)
        return sum(h * -(-w // 8) + (h if h != hmax else 0) for h, w in by_h.items()) + hmax - 1

    B = 1 << log_blowup
    main = [((B << c.log_height), c.main_width) for c in shard.chips]
    perm = [((B << c.log_height), 4 * c.perm_ext_width) for c in shard.chips]
    quot = [((B << c.log_height), 4 << c.log_quotient_degree) for c in shard.chips]
    total = commit(main) + commit(perm) + commit(quot)
    length = max(h for h, _ in main)
    while length > B:
        total += length // 2 + length // 2 - 1
        length //= 2
    return total
