"""The recursion-tree reduce over the shard farm (BASELINE.json north_star: "RCCL/xGMI only for the final
recursion-tree reduce"): `ZKMProver::compress` (crates/prover/src/lib.rs:617-957) as this prover sees it.

The reference turns the core shard proofs into one proof in a tree: a first layer with one compress-machine shard per core proof
(`first_layer_batch_size = 1`, lib.rs:626), then layers that each verify REDUCE_BATCH_SIZE = 2 proofs of the layer below (lib.rs:114; an odd
last proof goes up alone, :903-912) until one is left, and the shrink prover re-proves that root under the (2, 42) FRI configuration
(kb31_poseidon2.rs:217-227). Every one of those shards is `compress_prover.setup` + `commit` + `open` (lib.rs:813-836) — the same hot
path as a core shard, over the nine chips of the compress machine (crates/recursion/core/src/machine.rs:112-132), padded to one of the three
shapes of `RecursionShapeConfig::default()` (crates/recursion/core/src/shape.rs:134-171; data/recursion_shapes.json, extracted by
tests/golden/gen_recursion_shapes.py).

What cannot be had here is the *program* of such a shard: it comes out of the reference's recursion compiler (Rust, out of scope, SURVEY
section 2). `standin_program` builds a stand-in with the properties the prover's cost depends on: the reference's chips, heights and
widths; every chip's events filling `fill` (default 3/4) of its padded height; memory lookups that balance exactly (every address written
once with the multiplicity it is read with), so the shard verifies; and the children's commitments and public-values digests witnessed
through MemoryVar, absorbed by a Poseidon2 chain and committed as this shard's own public-values digest — a parent's proof depends on its
children's proofs, so the levels of the tree are a true dependency chain. It is NOT a verifier of its children.

The tree runs level by level through the farm (`Farm.run_queue`: whichever lane of whichever rank is free claims the next node of the
level); between levels every rank learns the 32 witnessed words of every node (`Farm.gather_words`, one small all-reduce) and rank 0
receives the proof streams (`Farm.gather_proofs`) — the only collectives on the path.
"""
import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import abi, field as F, recursion as R

ROOT = os.path.dirname(os.path.abspath(__file__))
CHIP_ORDER = ("BaseAlu", "ExtAlu", "MemoryConst", "MemoryVar", "Select", "Poseidon2WideDeg3", "ExpReverseBitsLen", "BatchFRI", "PublicValues")
CHILD_WORDS = 32          # what a node witnesses of each child: its three commitments (24 words) and eight public-value words
COMPRESS_FRI = (1, 84, 16)     # InnerSC::default(), crates/prover/src/lib.rs:192, kb31_poseidon2.rs:203-213
SHRINK_FRI = (2, 42, 16)       # InnerSC::compressed(), lib.rs:196, kb31_poseidon2.rs:217-227


def load_shapes() -> List[Dict[str, int]]:
    """`RecursionShapeConfig::default().allowed_shapes` (shape.rs:134-171), fastest first: chip name -> log2 height."""
    with open(os.path.join(ROOT, "data", "recursion_shapes.json")) as f:
        return json.load(f)["shapes"]


def tree_levels(n_leaves: int) -> List[List[tuple]]:
    """The reduce layers above `n_leaves` first-layer proofs: per layer, per node the indices of its children in the layer below —
    pairs, an odd last one alone (lib.rs:622-641: `while num_layer_inputs > batch_size { div_ceil(2) }`, :903-912)."""
    levels, n = [], n_leaves
    while n > 1:
        nodes = [(2 * i, 2 * i + 1) if 2 * i + 1 < n else (2 * i,) for i in range((n + 1) // 2)]
        levels.append(nodes)
        n = len(nodes)
    return levels


def child_words(proof: np.ndarray, recursion: bool) -> np.ndarray:
    """The 32 canonical words a parent witnesses of a child's ShardProof stream: the main / permutation /
    quotient commitments, and eight public values — a recursion shard's committed digest (RecursionPublicValues::digest, the last eight of
    231), a core shard's first eight."""
    w = np.asarray(proof, dtype=np.uint32)
    pv = w[-8:] if recursion else w[-231:-223]
    return F.from_monty(np.concatenate([w[:24], pv])).astype(np.uint64)


# ---- vectorised field helpers (host-side input generation only) -----------------------------------------------------------------------------

def _inv(a):
    """a^(p-2), element-wise (canonical uint64 in, canonical out; 0 -> 0)."""
    a = np.asarray(a, dtype=np.uint64)
    out = np.ones_like(a)
    e, base = F.P - 2, a.copy()
    while e:
        if e & 1:
            out = F.mul(out, base)
        base = F.mul(base, base)
        e >>= 1
    return out


def _ext_mul(x, y):
    """(n, 4) x (n, 4) over F[X]/(X^4 - 3)."""
    out = np.zeros_like(x)
    for i in range(4):
        for j in range(4):
            t = F.mul(x[:, i], y[:, j])
            if i + j >= 4:
                out[:, i + j - 4] = F.add(out[:, i + j - 4], F.mul(t, R.W))
            else:
                out[:, i + j] = F.add(out[:, i + j], t)
    return out


def _ext_inv(a):
    """Inverse in F[X]/(X^4 - 3): a(X) a(-X) = c0 + c2 X^2 lies in F[X^2]/((X^2)^2 - 3), whose norm to F is c0^2 - 3 c2^2."""
    b = a.copy()
    b[:, 1] = F.sub(0, a[:, 1])
    b[:, 3] = F.sub(0, a[:, 3])
    c = _ext_mul(a, b)                      # components 1 and 3 vanish
    n = F.sub(F.mul(c[:, 0], c[:, 0]), F.mul(R.W, F.mul(c[:, 2], c[:, 2])))
    ni = _inv(n)
    d = np.zeros_like(a)                    # (c0 - c2 X^2) / n
    d[:, 0] = F.mul(c[:, 0], ni)
    d[:, 2] = F.mul(F.sub(0, c[:, 2]), ni)
    return _ext_mul(b, d)


# ---- the stand-in program -----------------------------------------------------------------------------------------------------------------

class StandinProgram:
    """One compress-machine program at a shape (module docstring). `streams`: the flat Montgomery record streams the device generators
    take — per chip (preprocessed words, main events) in CHIP_ORDER, in the formats of ziren_amd/recursion.py. `patch(inputs)` rewrites the
    input-dependent words (the witnessed values, the chain's Poseidon2 events, the committed digest) in place: the program — every
    preprocessed trace, hence the proving key — does not depend on the inputs."""

    def __init__(self, shape: Dict[str, int], n_inputs: int, seed: int, permute_batch, fill: float = 0.75):
        assert n_inputs % 8 == 0 and n_inputs > 0
        self.shape, self.n_inputs, self.permute_batch = dict(shape), n_inputs, permute_batch
        self._build(seed, fill)

    # numpy throughout: the largest shape has 2^21 BatchFRI rows and two million extension-field instructions
    def _build(self, seed, fill):
        rng = np.random.default_rng(seed)
        P = F.P
        sh = self.shape
        cap = lambda name, per_row=1: int(fill * (per_row << sh[name]))   # noqa: E731
        n_in, n_chain = self.n_inputs, self.n_inputs // 8
        top = [2]                             # addresses 0 and 1 stay unused

        def new(n):
            a = np.arange(top[0], top[0] + n, dtype=np.int64)
            top[0] += n
            return a

        reads = []                            # every read of an address, tallied at the end
        vals = {}                             # address block -> (n, 4) canonical values; flattened into one table at the end

        def define(addrs, v):
            v = np.asarray(v, dtype=np.uint64)
            if v.ndim == 1:
                v = np.stack([v, np.zeros_like(v), np.zeros_like(v), np.zeros_like(v)], axis=1)
            vals[int(addrs[0])] = (addrs, v)

        table = [None]

        def value_of(addrs):                  # values of already-defined addresses (the table is rebuilt lazily)
            if table[0] is None or table[0].shape[0] < top[0]:
                t = np.zeros((top[0], 4), dtype=np.uint64)
                for a, v in vals.values():
                    t[a] = v
                table[0] = t
            return table[0][addrs]

        def invalidate():
            table[0] = None

        pick = lambda pool, shape_: pool[rng.integers(0, len(pool), shape_)]   # noqa: E731

        # --- witnessed values (MemoryVar): the inputs first, then bits and base elements
        n_var = max(cap("MemoryVar", R.VAR_MEM_ENTRIES_PER_ROW), n_in + 64)
        var_addr = new(n_var)
        var_val = rng.integers(0, P, n_var, dtype=np.uint64)
        is_bit = (np.arange(n_var) % 4 == 0) & (np.arange(n_var) >= n_in)
        var_val[is_bit] = rng.integers(0, 2, int(is_bit.sum()), dtype=np.uint64)
        var_val[:n_in] = 0                    # patched
        define(var_addr, var_val)
        bits = var_addr[is_bit]
        base_pool = [var_addr[n_in:][~is_bit[n_in:]]]
        # --- constants (MemoryConst writes): a zero for the sponge's initial capacity, base and extension values
        mem_budget = cap("MemoryConst", R.CONST_MEM_ENTRIES_PER_ROW)
        rows_fri = cap("BatchFRI")
        avg_k = max(4.5, rows_fri / (0.35 * mem_budget))
        kmax = int(np.ceil(2 * avg_k - 1))
        ks = rng.integers(1, kmax + 1, int(rows_fri / ((kmax + 1) / 2)) + 8)
        ks = ks[:int(np.searchsorted(np.cumsum(ks), rows_fri, side="right"))]
        if 0 < rows_fri - int(ks.sum()) <= kmax:
            ks = np.append(ks, rows_fri - int(ks.sum()))          # the last instruction takes the rows that are left
        n_fri = len(ks)
        n_readback = 8
        n_const = max(16, mem_budget - n_fri - n_readback - 1)
        az = new(1)
        define(az, np.zeros(1, dtype=np.uint64))
        const_addr = new(n_const)
        const_val = rng.integers(0, P, (n_const, 4), dtype=np.uint64)
        const_is_ext = np.arange(n_const) % 2 == 1
        const_val[~const_is_ext, 1:] = 0
        define(const_addr, const_val)
        base_pool.append(const_addr[~const_is_ext])
        ext_pool = [const_addr[const_is_ext]]
        invalidate()
        # --- the input chain: an overwrite-mode sponge over the witnessed inputs (patched per node; addresses and multiplicities fixed)
        chain_out = new(16 * n_chain).reshape(n_chain, 16)
        chain_in = np.zeros((n_chain, 16), dtype=np.int64)
        for k in range(n_chain):
            chain_in[k, :8] = var_addr[8 * k:8 * k + 8]
            chain_in[k, 8:] = az[0] if k == 0 else chain_out[k - 1, 8:]
        reads.append(chain_in.reshape(-1))
        define(chain_out.reshape(-1), np.zeros(16 * n_chain, dtype=np.uint64))     # patched
        digest_addr = chain_out[-1, :8]
        # --- Select: out1 = bit ? in2 : in1, out2 = bit ? in1 : in2
        n_sel = cap("Select")
        pool = np.concatenate(base_pool)
        s_bit, s_in1, s_in2 = pick(bits, n_sel), pick(pool, n_sel), pick(pool, n_sel)
        bv, xv, yv = value_of(s_bit)[:, 0], value_of(s_in1)[:, 0], value_of(s_in2)[:, 0]
        o1, o2 = np.where(bv == 1, yv, xv), np.where(bv == 1, xv, yv)
        s_out1, s_out2 = new(n_sel), new(n_sel)
        define(s_out1, o1)
        define(s_out2, o2)
        reads += [s_bit, s_in1, s_in2]
        base_pool += [s_out1, s_out2]
        invalidate()
        sel_events = np.stack([bv, o1, o2, xv, yv], axis=1)
        # --- Poseidon2Wide: the chain's rows first, then hashes of values that exist already
        n_pos = max(cap("Poseidon2WideDeg3"), n_chain + 1) - n_chain
        pool = np.concatenate(base_pool)
        p_in = pick(pool, (n_pos, 16))
        p_in_val = value_of(p_in.reshape(-1))[:, 0].reshape(n_pos, 16)
        p_out_val = np.asarray(self.permute_batch(p_in_val), dtype=np.uint64)
        p_out = new(16 * n_pos).reshape(n_pos, 16)
        define(p_out.reshape(-1), p_out_val.reshape(-1))
        reads.append(p_in.reshape(-1))
        base_pool.append(p_out.reshape(-1))
        invalidate()
        # --- ExpReverseBitsLen: result = x^(bit-reversed exponent), one row per exponent bit
        rows_exp = cap("ExpReverseBitsLen")
        nb = rng.integers(1, 32, rows_exp // 16 + 8)
        nb = nb[:int(np.searchsorted(np.cumsum(nb), rows_exp, side="right"))]
        if 0 < rows_exp - int(nb.sum()) <= 31:
            nb = np.append(nb, rows_exp - int(nb.sum()))
        n_exp = len(nb)
        exp_off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        pool = np.concatenate(base_pool)
        e_x = pick(pool, n_exp)
        e_bits = pick(bits, int(exp_off[-1]))
        e_xv, e_bv = value_of(e_x)[:, 0], value_of(e_bits)[:, 0]
        inst_of_row = np.repeat(np.arange(n_exp), nb)
        idx_in_inst = np.arange(int(exp_off[-1])) - exp_off[inst_of_row]
        accum = np.ones(n_exp, dtype=np.uint64)
        for i in range(int(nb.max())):                    # all instructions advance one bit per pass
            live = nb > i
            b_i = e_bv[exp_off[:-1][live] + i]
            m = np.where(b_i == 1, e_xv[live], np.uint64(1))
            accum[live] = F.mul(F.mul(accum[live], accum[live]), m)
        e_res = new(n_exp)
        define(e_res, accum)
        reads += [e_x, e_bits]
        base_pool.append(e_res)
        invalidate()
        # --- base and extension ALU, four waves each: a wave reads what the waves before it wrote
        alu = {}
        for which, name, pool_list in (("base", "BaseAlu", base_pool), ("ext", "ExtAlu", ext_pool)):
            n_alu = cap(name, R.ENTRIES_PER_ROW)
            ops, a_out, a_in1, a_in2, v_out, v_in1, v_in2 = [], [], [], [], [], [], []
            for wave in np.array_split(np.arange(n_alu), 4):
                n = len(wave)
                pool = np.concatenate(pool_list)
                op = rng.integers(0, 4, n)
                i1, i2 = pick(pool, n), pick(pool, n)
                x, y = value_of(i1), value_of(i2)
                if which == "base":
                    x0, y0 = x[:, 0], y[:, 0]
                    op[(op == R.DIV) & (y0 == 0)] = R.ADD
                    o = np.select([op == R.ADD, op == R.SUB, op == R.MUL], [F.add(x0, y0), F.sub(x0, y0), F.mul(x0, y0)], F.mul(x0, _inv(y0)))
                    o = np.stack([o, np.zeros_like(o), np.zeros_like(o), np.zeros_like(o)], axis=1)
                else:
                    op[(op == R.DIV) & ~y.any(axis=1)] = R.ADD
                    quo = _ext_mul(x, _ext_inv(y))
                    o = np.where((op == R.ADD)[:, None], F.add(x, y), np.where((op == R.SUB)[:, None], F.sub(x, y),
                                 np.where((op == R.MUL)[:, None], _ext_mul(x, y), quo)))
                ao = new(n)
                define(ao, o)
                invalidate()
                reads += [i1, i2]
                pool_list.append(ao)
                ops.append(op); a_out.append(ao); a_in1.append(i1); a_in2.append(i2); v_out.append(o); v_in1.append(x); v_in2.append(y)
            alu[which] = [np.concatenate(z) for z in (ops, a_out, a_in1, a_in2, v_out, v_in1, v_in2)]
        # --- BatchFRI: acc = sum over an instruction's rows of alpha_pow * (p_at_z - p_at_x); the accumulator is written on the last row
        n_rows_fri = int(ks.sum())
        epool, bpool = np.concatenate(ext_pool), np.concatenate(base_pool)
        f_al, f_pz, f_px = pick(epool, n_rows_fri), pick(epool, n_rows_fri), pick(bpool, n_rows_fri)
        al, pz, px = value_of(f_al), value_of(f_pz), value_of(f_px)[:, 0]
        d = pz.copy()
        d[:, 0] = F.sub(pz[:, 0], px)
        term = _ext_mul(al, d)
        f_off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
        cs = np.zeros((n_rows_fri + 1, 4), dtype=np.uint64)
        # running sums modulo p: a cumulative sum of values below 2^31 stays below 2^64 for 2^33 rows
        cs[1:] = np.cumsum(term, axis=0, dtype=np.uint64)
        inst = np.repeat(np.arange(n_fri), ks)
        acc = (cs[1:] + (np.uint64(P) - cs[f_off[:-1]][inst] % np.uint64(P))) % np.uint64(P)
        f_acc = new(n_fri)
        define(f_acc, acc[f_off[1:] - 1])
        is_end = np.zeros(n_rows_fri, dtype=np.uint64)
        is_end[f_off[1:] - 1] = 1
        reads += [f_al, f_pz, f_px, f_acc]                     # the accumulator is read back once by the constant table (below)
        invalidate()
        # --- read-backs by the constant table (MemAccessKind::Read: negated multiplicity), the committed digest
        results = np.concatenate([alu["base"][1], alu["ext"][1]])
        readback = results[:: max(1, len(results) // n_readback)][:n_readback]
        reads += [readback, digest_addr]
        count = np.bincount(np.concatenate([np.asarray(r, dtype=np.int64).reshape(-1) for r in reads]), minlength=top[0]).astype(np.uint64)
        self.n_addresses = int(top[0])

        def mont(rows):
            return F.to_monty(np.asarray(rows, dtype=np.uint64)).reshape(-1)

        # --- record streams (formats: ziren_amd/recursion.py)
        s = {}
        for which, (ik, ek) in (("base", ("base_instrs", "base_events")), ("ext", ("ext_instrs", "ext_events"))):
            op, ao, i1, i2, o, x, y = alu[which]
            ins = np.zeros((len(op), R.ACCESS_COLS), dtype=np.uint64)
            ins[:, 0], ins[:, 1], ins[:, 2] = ao, i1, i2
            ins[np.arange(len(op)), 3 + op] = 1
            ins[:, 7] = count[ao]
            ev = np.concatenate([o, x, y], axis=1) if which == "ext" else np.stack([o[:, 0], x[:, 0], y[:, 0]], axis=1)
            s[ik], s[ek] = mont(ins), mont(ev)
        mem_val = np.concatenate([np.zeros((1, 4), dtype=np.uint64), const_val, value_of(readback), value_of(f_acc)])
        mem_addr = np.concatenate([az, const_addr, readback, f_acc])
        mem_mult = np.concatenate([count[az], count[const_addr], np.full(len(readback) + n_fri, P - 1, dtype=np.uint64)])
        s["mem_entries"] = mont(np.concatenate([mem_val, mem_addr[:, None].astype(np.uint64), mem_mult[:, None]], axis=1))
        assert len(mem_addr) <= (R.CONST_MEM_ENTRIES_PER_ROW << sh["MemoryConst"])
        s["var_prep"] = mont(np.stack([var_addr.astype(np.uint64), count[var_addr]], axis=1))
        self._var_values = np.zeros((n_var, 4), dtype=np.uint64)
        self._var_values[:, 0] = var_val
        sel_prep = np.stack([np.ones(n_sel, dtype=np.uint64)] + [a.astype(np.uint64) for a in (s_bit, s_out1, s_out2, s_in1, s_in2)] + [count[s_out1], count[s_out2]], axis=1)
        s["select_prep"], s["select_events"] = mont(sel_prep), mont(sel_events)
        all_in = np.concatenate([chain_in, p_in])
        all_out = np.concatenate([chain_out, p_out])
        pos_prep = np.zeros((n_chain + n_pos, R.POSEIDON2_WIDE_PREP_WIDTH), dtype=np.uint64)
        pos_prep[:, :16] = all_in
        pos_prep[:, 16:48:2] = all_out
        pos_prep[:, 17:48:2] = count[all_out]
        pos_prep[:, 48] = P - 1
        s["poseidon2_prep"] = mont(pos_prep)
        self._pos_events = np.zeros((n_chain + n_pos, 32), dtype=np.uint64)
        self._pos_events[n_chain:, :16], self._pos_events[n_chain:, 16:] = p_in_val, p_out_val
        n_rows_exp = int(exp_off[-1])
        exp_prep = np.zeros((n_rows_exp, R.EXP_REVERSE_BITS_PREP_COLS), dtype=np.uint64)
        first, last = idx_in_inst == 0, idx_in_inst == nb[inst_of_row] - 1
        exp_prep[:, 0] = e_x[inst_of_row]
        exp_prep[first, 1] = P - 1
        exp_prep[:, 2], exp_prep[:, 3] = e_bits, P - 1
        exp_prep[:, 4] = e_res[inst_of_row]
        exp_prep[last, 5] = count[e_res]
        exp_prep[:, 6], exp_prep[:, 7], exp_prep[:, 8], exp_prep[:, 9] = idx_in_inst, first, last, 1
        s["exp_prep"] = mont(exp_prep)
        s["exp_bases"], s["exp_bits"], s["exp_offsets"] = mont(e_xv), mont(e_bv), exp_off.astype(np.uint32)
        fri_prep = np.stack([np.ones(n_rows_fri, dtype=np.uint64), is_end] + [a.astype(np.uint64) for a in (f_acc[inst], f_al, f_pz, f_px)], axis=1)
        s["batch_fri_prep"] = mont(fri_prep)
        s["batch_fri_main"] = mont(np.concatenate([acc, al, pz, px[:, None]], axis=1))
        pv_prep = np.zeros((R.DIGEST_SIZE, R.PUBLIC_VALUES_PREP_COLS), dtype=np.uint64)
        pv_prep[np.arange(8), np.arange(8)] = 1
        pv_prep[:, 8], pv_prep[:, 9] = digest_addr, P - 1
        s["pv_prep"] = mont(pv_prep)
        self.streams = s
        self.counts = {"BaseAlu": len(alu["base"][0]), "ExtAlu": len(alu["ext"][0]), "MemoryConst": len(mem_addr), "MemoryVar": n_var, "Select": n_sel,
                       "Poseidon2WideDeg3": n_chain + n_pos, "ExpReverseBitsLen": n_rows_exp, "BatchFRI": n_rows_fri, "PublicValues": 8}
        self.n_chain = n_chain
        self.patch(np.zeros(self.n_inputs, dtype=np.uint64))

    def fill(self) -> Dict[str, float]:
        """Events per chip over the rows its padded height holds."""
        per_row = {"BaseAlu": 4, "ExtAlu": 4, "MemoryConst": 2, "MemoryVar": 2}
        return {c: round(self.counts[c] / (per_row.get(c, 1) << self.shape[c]), 3) for c in CHIP_ORDER}

    def witness(self, inputs: Sequence[int]) -> dict:
        """The words `inputs` (canonical, n_inputs of them) decide, as Montgomery words: the leading rows of MemoryVar's values and of
        the Poseidon2 events (the chain), PublicValues' eight rows, and the digest (canonical). The streams themselves are not touched:
        every lane writes a node's witness into its own copy."""
        inputs = np.asarray(inputs, dtype=np.uint64) % np.uint64(F.P)
        assert len(inputs) == self.n_inputs
        var = np.zeros((self.n_inputs, 4), dtype=np.uint64)
        var[:, 0] = inputs
        # the chain on the host, through the library's own host permutation (zkm_host_poseidon2_permute: no context, no device round
        # trip, the GIL released meanwhile) — a handful of permutations per node
        from . import lib
        L = lib.load()
        mont_in = F.to_monty(inputs)
        pos = np.zeros((self.n_chain, 32), dtype=np.uint32)
        state = np.zeros(16, dtype=np.uint32)
        for k in range(self.n_chain):
            state[:8] = mont_in[8 * k:8 * k + 8]
            pos[k, :16] = state
            L.zkm_host_poseidon2_permute(abi.as_u32p(state))
            pos[k, 16:] = state
        return {"var_values": F.to_monty(var).reshape(-1), "poseidon2_events": pos.reshape(-1),
                "pv_main": state[:8].copy(), "digest": F.from_monty(state[:8]).astype(np.uint64)}

    def patch(self, inputs: Sequence[int]):
        """The streams with `inputs` witnessed, in place (host-side users: the oracle's traces in tests)."""
        w = self.witness(inputs)
        for k in ("var_values", "poseidon2_events", "pv_main"):
            if k not in self.streams:
                self.streams[k] = {"var_values": F.to_monty(self._var_values).reshape(-1), "poseidon2_events": F.to_monty(self._pos_events).reshape(-1),
                                   "pv_main": np.zeros(8, dtype=np.uint32)}[k]
            self.streams[k][:len(w[k])] = w[k]
        self.digest = w["digest"]

    @staticmethod
    def public_values(digest) -> np.ndarray:
        pv = np.zeros(231, dtype=np.uint64)            # PROOF_MAX_NUM_PVS
        pv[R.PV_DIGEST_POS:R.PV_DIGEST_POS + 8] = digest
        return F.to_monty(pv)


def chip_specs():
    """Per chip of CHIP_ORDER: (preprocessed stream key, main stream key, preprocessed width, main width, records per row, recorder)."""
    return [("base_instrs", "base_events", R.ENTRIES_PER_ROW * R.ACCESS_COLS, R.ENTRIES_PER_ROW * R.BASE_VALUE_COLS, R.ENTRIES_PER_ROW, lambda lh, i: R.record_chip(False, lh, i)),
            ("ext_instrs", "ext_events", R.ENTRIES_PER_ROW * R.ACCESS_COLS, R.ENTRIES_PER_ROW * R.EXT_VALUE_COLS, R.ENTRIES_PER_ROW, lambda lh, i: R.record_chip(True, lh, i)),
            ("mem_entries", None, R.CONST_MEM_ENTRIES_PER_ROW * R.CONST_MEM_ENTRY_COLS, 1, R.CONST_MEM_ENTRIES_PER_ROW, R.record_mem_const),
            ("var_prep", "var_values", 2 * R.VAR_MEM_ENTRIES_PER_ROW, 4 * R.VAR_MEM_ENTRIES_PER_ROW, R.VAR_MEM_ENTRIES_PER_ROW, R.record_mem_var),
            ("select_prep", "select_events", R.SELECT_PREP_COLS, R.SELECT_COLS, 1, R.record_select),
            ("poseidon2_prep", "poseidon2_events", R.POSEIDON2_WIDE_PREP_WIDTH, R.POSEIDON2_WIDE_WIDTH, 1, R.record_poseidon2_wide),
            ("exp_prep", "exp_main", R.EXP_REVERSE_BITS_PREP_COLS, R.EXP_REVERSE_BITS_COLS, 1, R.record_exp_reverse_bits),
            ("batch_fri_prep", "batch_fri_main", R.BATCH_FRI_PREP_COLS, R.BATCH_FRI_COLS, 1, R.record_batch_fri),
            ("pv_prep", "pv_main", R.PUBLIC_VALUES_PREP_COLS, 1, 1, lambda lh, i: R.record_public_values(i))]


def record_machine(shape: Dict[str, int]):
    """The compress machine's nine chips (machine.rs:112-132) recorded at a shape's heights, in CHIP_ORDER, prep_index = position."""
    return [spec[5](shape[name], i) for i, (name, spec) in enumerate(zip(CHIP_ORDER, chip_specs()))]


# ---- one GPU lane ---------------------------------------------------------------------------------------------------------------------------

class ReduceLane:
    """One context of one GPU proving recursion shards: per (shape, FRI configuration) a `HipProver` over the recorded machine; per program a
    proving key (`setup`: the preprocessed traces born on the device, committed) kept for every node that runs that program — the
    reference's `compress_prover.setup(&program)` (lib.rs:816) per proof is `setup_ms` in the measurements, not redone per node."""

    def __init__(self, ctx, specialize: bool = True, pin: bool = True):
        from . import prover
        self.ctx, self.specialize, self.pin = ctx, specialize, pin
        self._prover_mod = prover
        self.provers, self.keys = {}, {}
        self.out = np.zeros(1 << 23, dtype=np.uint32)
        self.setup_ms, self._pins = {}, {}
        self.host_s = {"prefetch": 0.0, "traces": 0.0, "prove_shard": 0.0, "free": 0.0, "nodes": 0}     # wall seconds of this lane's host thread per phase

    def prover_for(self, shape_idx: int, shape, fri):
        key = (shape_idx, tuple(fri))
        if key not in self.provers:
            from . import synth
            recs = record_machine(shape)
            hp = self._prover_mod.HipProver(recs, abi.FriConfig(*fri), synth.NUM_PV_ELTS, ctx=self.ctx, specialize=self.specialize)
            self.provers[key] = (hp, recs)
        return self.provers[key]

    PATCHED = ("var_values", "poseidon2_events", "pv_main")

    def _pinned(self, prog: StandinProgram):
        """This lane's own page-locked copies of the program's main-event streams (once per program and lane): H2D at the link's rate
        instead of a pageable copy's. The streams a node's witness is written into have TWO copies, used alternately: node i + 1's words
        are written (and its copy queued) while node i's copy may still be in flight."""
        pins = self._pins.setdefault(id(prog), {})
        if not pins:
            alloc = (lambda n: self.ctx.host_alloc((n,))) if self.pin else (lambda n: np.empty(n, dtype=np.uint32))
            for name, spec in zip(CHIP_ORDER, chip_specs()):
                ek = spec[1]
                if ek is None:
                    continue
                if name == "ExpReverseBitsLen":       # one buffer [bases | offsets | bits] (ZKM_TG_EXP_REVERSE_BITS)
                    src = np.concatenate([prog.streams["exp_bases"], prog.streams["exp_offsets"], prog.streams["exp_bits"]]).astype(np.uint32)
                else:
                    src = prog.streams[ek]
                copies = []
                for _ in range(2 if ek in self.PATCHED else 1):
                    buf = alloc(len(src))
                    buf[:] = src
                    copies.append(buf)
                pins[ek] = copies
            pins["_turn"] = 0
        return pins

    def key_for(self, prog_id, prog: StandinProgram, shape_idx: int, fri):
        """(prover, chips, proving key, challenger after the key) of a program under a FRI configuration."""
        import time
        k = (prog_id, tuple(fri))
        if k not in self.keys:
            hp, recs = self.prover_for(shape_idx, prog.shape, fri)
            t0 = time.perf_counter()
            preps = self.ctx.tracegen_shard([(abi.TG_FLAT, prog.streams[spec[0]], r.log_height, {"width": spec[2]}) for spec, r in zip(chip_specs(), recs)])
            pk = hp.setup(preps, [int(r.local_only) for r in recs], F.to_monty(0), F.to_monty(np.zeros(14, dtype=np.uint64)))
            self.ctx.synchronize()
            self.setup_ms[k] = 1e3 * (time.perf_counter() - t0)
            ch0 = self._prover_mod.new_challenger()
            pk.observe_into(ch0)
            self.keys[k] = (hp, recs, pk, ch0)
        return self.keys[k]

    def prefetch(self, prog: StandinProgram, inputs):
        """A node's events on their way to HBM (zkm_events_upload_async on the context's DMA stream: under whatever this lane is proving):
        the witness is written into this turn's page-locked copies, every stream's copy queued. Returns the handle `prove` takes."""
        import time
        t0 = time.perf_counter()
        w = prog.witness(inputs)
        pins = self._pinned(prog)
        turn = pins["_turn"]
        pins["_turn"] = turn ^ 1
        dev = {}
        for ek, copies in pins.items():
            if ek == "_turn":
                continue
            buf = copies[turn % len(copies)]
            if ek in self.PATCHED:
                buf[:len(w[ek])] = w[ek]
            dev[ek] = self.ctx.events_upload_async(buf)
        self.host_s["prefetch"] += time.perf_counter() - t0
        return (w, dev)

    def traces(self, prog: StandinProgram, recs, witness=None, handle=None):
        """The nine main traces born on the device from the program's event streams — one zkm_tracegen_shard call: every generator queued
        behind the copy of its events (already in HBM with `handle`, else copied now from the page-locked streams), one synchronisation."""
        if handle is None:
            pins = self._pinned(prog)
            ev = {ek: copies[0] for ek, copies in pins.items() if ek != "_turn"}
            if witness is not None:
                for k in self.PATCHED:
                    ev[k][:len(witness[k])] = witness[k]
        else:
            ev = handle[1]
        n_exp = len(prog.streams["exp_bases"])
        items = []
        for name, spec, r in zip(CHIP_ORDER, chip_specs(), recs):
            ek, mw = spec[1], spec[3]
            if name == "Poseidon2WideDeg3":
                items.append((abi.TG_POSEIDON2_WIDE, ev[ek], r.log_height, None))
            elif name == "ExpReverseBitsLen":
                items.append((abi.TG_EXP_REVERSE_BITS, ev[ek], r.log_height, {"n": n_exp, "rows": int(prog.streams["exp_offsets"][-1]) if n_exp else 0}))
            elif ek is None:
                items.append((abi.TG_FLAT, None, r.log_height, {"width": mw}))
            else:
                items.append((abi.TG_FLAT, ev[ek], r.log_height, {"width": mw}))
        return self.ctx.tracegen_shard(items)

    def prove(self, prog_id, prog: StandinProgram, shape_idx: int, fri, inputs, salt: int = 0, handle=None) -> np.ndarray:
        """One recursion shard: witness `inputs`, events -> device traces -> commit + open. `handle` = what `prefetch` returned for these
        inputs (the events are in HBM or on their way), else they are copied now. `salt` (the node's index in the tree) is observed into
        the transcript after the key, as bench.py's queue does for core shards."""
        hp, recs, pk, ch0 = self.key_for(prog_id, prog, shape_idx, fri)
        import time
        t0 = time.perf_counter()
        w = handle[0] if handle is not None else prog.witness(inputs)
        born = self.traces(prog, recs, w, handle)
        t1 = time.perf_counter()
        ch = ch0.copy()
        if salt:
            from . import lib
            idx = np.array([salt], dtype=np.uint32)
            lib.load().zkm_challenger_observe(C.byref(ch), abi.as_u32p(idx), C.c_size_t(1))
        proof = hp.prove_shard(pk, prog.public_values(w["digest"]), born, ch, out=self.out)
        t2 = time.perf_counter()
        for t in born:
            t.free()
        if handle is not None:
            for d in handle[1].values():
                d.free()
        t3 = time.perf_counter()
        self.host_s["traces"] += t1 - t0
        self.host_s["prove_shard"] += t2 - t1
        self.host_s["free"] += t3 - t2
        self.host_s["nodes"] += 1
        return proof

    def close(self):
        for hp, recs, pk, ch0 in self.keys.values():
            pk.free()
        self.keys.clear()
        if self.pin:
            for pins in self._pins.values():
                for ek, copies in pins.items():
                    if ek != "_turn":
                        for arr in copies:
                            self.ctx.host_free(arr)
        self._pins.clear()


# ---- the tree -----------------------------------------------------------------------------------------------------------------------------------

class TreePlan:
    """Which shape and FRI configuration each layer of the tree runs at. The reference picks, per program, the smallest allowed shape that
    holds its events; which that is for a first-layer program (it verifies a whole core shard proof) and for a reduce program (two
    recursion proofs) comes out of the recursion compiler — here it is a stated assumption: first layer at shape `leaf_shape`, reduce
    layers at `reduce_shape`, the shrink at `shrink_shape`."""

    def __init__(self, leaf_shape: int = 1, reduce_shape: int = 0, shrink_shape: int = 0):
        self.leaf_shape, self.reduce_shape, self.shrink_shape = leaf_shape, reduce_shape, shrink_shape


class _Board:
    """Where the nodes of a pipelined tree publish the 32 words their parents witness: the process group's key-value store across ranks
    (c10d `Store.set / check / get`), a dictionary under a condition variable inside one process. A store client is one connection:
    a blocking `get` would hold it against this process's other lanes (whose `set` may be the very thing it waits for), so waiting is
    `check` (non-blocking) between short sleeps, and every store call of the process goes through the farm's one lock."""

    def __init__(self, farm, prefix):
        import threading
        self.prefix, self.local, self.cv, self.failed = prefix, {}, threading.Condition(), False
        self.store, self.lock = farm.store()

    def publish(self, g, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        with self.cv:
            self.local[g] = w
            self.cv.notify_all()
        if self.store is not None:
            with self.lock:
                self.store.set(f"{self.prefix}_{g}", w.tobytes())

    def ready(self, g):
        with self.cv:
            if g in self.local:
                return True
        if self.store is None:
            return False
        with self.lock:
            return bool(self.store.check([f"{self.prefix}_{g}"]))

    def wait(self, g, timeout_s: float = 600.0):
        import time
        t0 = time.perf_counter()
        while True:
            with self.cv:
                if g in self.local:
                    return self.local[g]
                if self.failed:
                    raise RuntimeError("another lane of the tree failed")
                if self.store is None:
                    self.cv.wait(timeout=0.05)
                    continue
            with self.lock:
                have = bool(self.store.check([f"{self.prefix}_{g}"]))
                w = np.frombuffer(self.store.get(f"{self.prefix}_{g}"), dtype=np.uint64).copy() if have else None
            if w is not None:
                with self.cv:
                    self.local[g] = w
                return w
            if time.perf_counter() - t0 > timeout_s:
                raise RuntimeError(f"node {g} of the tree was not published within {timeout_s:.0f} s (a rank failed?)")
            time.sleep(0.0002)

    def fail(self):
        with self.cv:
            self.failed = True
            self.cv.notify_all()

    def close(self):
        self.local.clear()


class ReduceTree:
    """The programs of a tree (one per (shape, children) pair: every node of a layer runs the same program on its own inputs) and the
    level-by-level driver over a `Farm`."""

    def __init__(self, plan: TreePlan, permute_batch, fill: float = 0.75, seed: int = 0x7ee, shapes=None):
        self.plan, self.shapes = plan, shapes or load_shapes()
        self.permute_batch, self.fill, self.seed = permute_batch, fill, seed
        self.programs = {}

    def program(self, shape_idx: int, n_children: int) -> StandinProgram:
        k = (shape_idx, n_children)
        if k not in self.programs:
            self.programs[k] = StandinProgram(self.shapes[shape_idx], CHILD_WORDS * n_children, self.seed + 16 * shape_idx + n_children, self.permute_batch, self.fill)
        return self.programs[k]

    def layers(self, n_core: int):
        """[(layer name, shape index, FRI, [children per node])]: first layer, reduce layers, shrink."""
        out = [("first", self.plan.leaf_shape, COMPRESS_FRI, [(i,) for i in range(n_core)])]
        for lv, nodes in enumerate(tree_levels(n_core)):
            out.append((f"reduce{lv + 1}", self.plan.reduce_shape, COMPRESS_FRI, nodes))
        out.append(("shrink", self.plan.shrink_shape, SHRINK_FRI, [(0,)]))
        return out

    def run(self, farm, lanes: Sequence[ReduceLane], core_words: np.ndarray, pipelined: bool = True):
        """Prove the whole tree over `core_words` ((n_core, 32) canonical: `child_words` of the gathered core proofs, known on every rank).
        Returns (per layer the gathered proof streams — rank 0 only, None elsewhere —, per layer the (n, 32) witnessed words of its
        nodes). Every rank calls this with the same arguments.

        `pipelined` (default), as the reference runs it (lib.rs:655-915: inputs, traces and proofs flow through channels, a layer-r node is
        generated as soon as its two children are proven): all nodes of the tree are ONE queue in layer order; a free lane of any rank
        claims the next node, waits — only if it has to — for its children's 32 words (published through the process group's key-value store,
        128 bytes per node; a dictionary inside one process), proves it and publishes its own. A node only waits for nodes in front of it in
        the queue, which somebody has already claimed: no deadlock, and no barrier between layers — the narrow top of the tree overlaps with
        the wide layers' tails. The proof streams go to rank 0 in ONE gather at the end (`Farm.gather_proofs`): the tree's only collective.
        `pipelined=False`: layer by layer (`run_layers`), a queue, an all-reduce of the words and a gather per layer."""
        if not pipelined:
            return self.run_layers(farm, lanes, core_words)
        import threading
        import time
        core = np.asarray(core_words, dtype=np.uint64)
        layers = self.layers(len(core))
        offs = np.concatenate([[0], np.cumsum([len(nodes) for *_, nodes in layers])]).astype(int)
        n_total = int(offs[-1])
        layer_of = np.repeat(np.arange(len(layers)), [len(nodes) for *_, nodes in layers])
        progs = [{nc: self.program(si, nc) for nc in {len(ch) for ch in nodes}} for _, si, _, nodes in layers]
        epoch = farm.open_epoch("reduce_tree")
        board = _Board(farm, f"zkm_tree_{epoch}")

        def inputs_of(g):             # blocks until the children's words are published
            L = int(layer_of[g])
            ch = layers[L][3][g - offs[L]]
            if L == 0:
                return np.concatenate([core[c] for c in ch])
            return np.concatenate([board.wait(int(offs[L - 1]) + c) for c in ch])

        def children_ready(g):
            L = int(layer_of[g])
            return L == 0 or all(board.ready(int(offs[L - 1]) + c) for c in layers[L][3][g - offs[L]])

        results = [([], [], []) for _ in lanes]
        errors = []
        t_start = time.perf_counter()

        def work(j):
            lane = lanes[j]
            ids, proofs, done_at = results[j]
            try:
                cur, handle = farm.claim("reduce_tree"), None
                while cur < n_total:
                    nxt = farm.claim("reduce_tree")
                    L = int(layer_of[cur])
                    _, si, fri, nodes = layers[L]
                    ch = nodes[cur - offs[L]]
                    inputs = inputs_of(cur)
                    if handle is None and hasattr(lane, "prefetch"):
                        handle = lane.prefetch(progs[L][len(ch)], inputs)
                    nxt_handle = None
                    if nxt < n_total and hasattr(lane, "prefetch") and children_ready(nxt):      # its events cross PCIe under this proof
                        Ln = int(layer_of[nxt])
                        nxt_handle = lane.prefetch(progs[Ln][len(layers[Ln][3][nxt - offs[Ln]])], inputs_of(nxt))
                    kw = {"handle": handle} if handle is not None else {}
                    proof = lane.prove((si, len(ch)), progs[L][len(ch)], si, fri, inputs, salt=cur + 1, **kw).copy()
                    board.publish(cur, child_words(proof, True))
                    ids.append(cur)
                    proofs.append(proof)
                    done_at.append(time.perf_counter() - t_start)
                    cur, handle = nxt, nxt_handle
            except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread)
                errors.append(e)
                board.fail()

        try:
            if len(lanes) == 1:
                work(0)
            else:
                ts = [threading.Thread(target=work, args=(j,)) for j in range(len(lanes))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
        finally:
            farm.close_epoch("reduce_tree")
        if errors:
            raise errors[0]
        ids = [i for r in results for i in r[0]]
        proofs = [p for r in results for p in r[1]]
        gathered = farm.gather_proofs(ids, proofs, n_total)
        all_words = np.stack([board.wait(g) for g in range(n_total)])
        # when each layer's last node finished on this rank (seconds from the start): the layers overlap, so these are not durations
        self.layer_seconds = [(layers[L][0], len(layers[L][3]), max([t for r in results for g, t in zip(r[0], r[2]) if layer_of[g] == L], default=0.0)) for L in range(len(layers))]
        self.layer_seconds_are = "finish times (pipelined: layers overlap)"
        streams = [None if gathered is None else gathered[offs[L]:offs[L + 1]] for L in range(len(layers))]
        words = [all_words[offs[L]:offs[L + 1]] for L in range(len(layers))]
        board.close()
        return streams, words

    def run_layers(self, farm, lanes: Sequence[ReduceLane], core_words: np.ndarray, on_layer=None):
        """The tree layer by layer: every layer a claim queue (`Farm.run_queue`), then `gather_words` (every rank learns what the next layer
        witnesses) and `gather_proofs` (rank 0 receives the streams). Node salts as in `run`: the node's position in the tree + 1."""
        import time
        below = np.asarray(core_words, dtype=np.uint64)
        streams, words, salt0 = [], [], 1
        self.layer_seconds = []
        for name, shape_idx, fri, nodes in self.layers(len(below)):
            progs = {nc: self.program(shape_idx, nc) for nc in {len(ch) for ch in nodes}}

            def lane_pair(lane, _nodes=nodes, _progs=progs, _below=below, _salt0=salt0, _si=shape_idx, _fri=fri):
                inputs_of = lambda i: np.concatenate([_below[c] for c in _nodes[i]])   # noqa: E731

                def prefetch(i):          # node i's events start crossing PCIe while this lane proves the node before it
                    return lane.prefetch(_progs[len(_nodes[i])], inputs_of(i)) if hasattr(lane, "prefetch") else None

                def prove(i, handle):
                    ch = _nodes[i]
                    kw = {"handle": handle} if handle is not None else {}
                    return lane.prove((_si, len(ch)), _progs[len(ch)], _si, _fri, inputs_of(i), salt=_salt0 + i, **kw).copy()
                return prove, prefetch

            t0 = time.perf_counter()
            ids, proofs = farm.run_queue(len(nodes), queue="reduce", lanes=[lane_pair(l) for l in lanes])
            mine = np.stack([child_words(p, True) for p in proofs]) if proofs else np.zeros((0, CHILD_WORDS), dtype=np.uint64)
            below = farm.gather_words(ids, mine, len(nodes), CHILD_WORDS).astype(np.uint64)
            gathered = farm.gather_proofs(ids, proofs, len(nodes))
            self.layer_seconds.append((name, len(nodes), time.perf_counter() - t0))
            self.layer_seconds_are = "durations (layer by layer)"
            streams.append(gathered)
            words.append(below)
            salt0 += len(nodes)
            if on_layer is not None:
                on_layer(name, shape_idx, fri, nodes, gathered)
        return streams, words
