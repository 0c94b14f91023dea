"""The recursion-tree reduce (ziren_amd/reduce.py): the reference's shapes as data, the tree's layer structure, the stand-in program's
soundness (the restated verifier accepts a shard of it, rejects it with a wrong multiplicity), the witness, and the level-by-level driver
over a gloo process group with stub lanes. GPU: the smallest reference shape bit-exact against the oracle, the two larger ones through
the restated verifier, an eight-leaf tree through the farm on one device."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from ziren_amd import abi, field as F, recursion as R, reduce as RD, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = {"MemoryVar": 8, "Select": 8, "MemoryConst": 7, "BatchFRI": 8, "BaseAlu": 6, "ExtAlu": 6, "ExpReverseBitsLen": 8, "Poseidon2WideDeg3": 6, "PublicValues": 4}


def oracle_permute(oracle):
    return lambda v: F.from_monty(oracle.poseidon2_permute_batch(F.to_monty(np.asarray(v, dtype=np.uint64))))


def host_shard(prog, oracle, inputs):
    """The program as the oracle proves it: recorded chips with host traces (the reference's generate_trace / generate_preprocessed_trace:
    records end to end, zero padded; Poseidon2Wide and ExpReverseBitsLen rows from the oracle's row builders)."""
    prog.patch(inputs)
    recs = RD.record_machine(prog.shape)
    for name, spec, r in zip(RD.CHIP_ORDER, RD.chip_specs(), recs):
        pk_key, ev_key, pw, mw, per_row, _ = spec
        r.prep_trace = R.flat_trace(prog.streams[pk_key], pw, r.log_height, per_row)
        if name == "Poseidon2WideDeg3":
            r.trace = oracle.tracegen_poseidon2_wide(prog.streams[ev_key], r.log_height)
        elif name == "ExpReverseBitsLen":
            r.trace = oracle.tracegen_exp_reverse_bits(prog.streams["exp_bases"], prog.streams["exp_bits"], prog.streams["exp_offsets"], r.log_height)
        elif ev_key is None:
            r.trace = np.zeros((1 << r.log_height, mw), dtype=np.uint32)
        else:
            r.trace = R.flat_trace(prog.streams[ev_key], mw, r.log_height, per_row)
    return recs


def oracle_key(oracle, recs, log_blowup):
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    opk = oracle.Pk([r.prep_trace for r in recs], [int(r.local_only) for r in recs], F.to_monty(0), igcs, log_blowup)
    ch = oracle.new_challenger()
    opk.observe_into(ch)
    return opk, ch


def test_shapes_are_the_references():
    shapes = RD.load_shapes()
    assert len(shapes) == 3 and all(set(s) == set(RD.CHIP_ORDER) for s in shapes)
    assert shapes[0] == {"MemoryVar": 18, "Select": 18, "MemoryConst": 16, "BatchFRI": 17, "BaseAlu": 15, "ExtAlu": 15, "ExpReverseBitsLen": 17,
                         "Poseidon2WideDeg3": 16, "PublicValues": 4}
    cells = [sum((1 << s[r.name]) * (r.prep_width + r.main_width) for r in RD.record_machine(s)) for s in shapes]
    assert cells[0] < cells[1] < cells[2]          # "fastest shape" first (shape.rs:135)
    src = "/root/reference/crates/recursion/core/src/shape.rs"
    if not os.path.exists(src):
        pytest.skip("the reference is not on this box: the committed data was compared with it in the build container")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "gen_recursion_shapes.py")], capture_output=True, text=True, check=True)
    assert subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "--", "ziren_amd/data/recursion_shapes.json"]).returncode == 0, out.stdout


def test_tree_levels_follow_the_references_layer_rule():
    """lib.rs:622-641: height 0 for one first-layer proof, else 1 + the number of halvings until at most REDUCE_BATCH_SIZE inputs are left;
    every layer pairs its inputs and an odd last one goes up alone (:903-912)."""
    for n in range(1, 70):
        expected = 0 if n == 1 else 1
        k = n
        while k > 2:
            k = -(-k // 2)
            expected += 1
        levels = RD.tree_levels(n)
        assert len(levels) == expected
        below = n
        for nodes in levels:
            assert [c for ch in nodes for c in ch] == list(range(below))
            assert all(len(ch) == 2 for ch in nodes[:-1]) and len(nodes[-1]) in (1, 2)
            below = len(nodes)
        assert below == 1
    tree = RD.ReduceTree(RD.TreePlan(2, 1, 0), None)
    names = [(nm, si, fri) for nm, si, fri, _ in tree.layers(5)]
    assert names == [("first", 2, RD.COMPRESS_FRI), ("reduce1", 1, RD.COMPRESS_FRI), ("reduce2", 1, RD.COMPRESS_FRI), ("reduce3", 1, RD.COMPRESS_FRI),
                     ("shrink", 0, RD.SHRINK_FRI)]


def test_standin_program_fills_its_shape_and_balances(oracle):
    """Every chip's events fill three quarters of its padded height; the memory lookups cancel exactly (address by address: the
    multiplicity written = the reads), which the restated verifier sees as a zero cumulative sum; one wrong multiplicity and it rejects."""
    prog = RD.StandinProgram(TINY, 64, 3, oracle_permute(oracle))
    fill = prog.fill()
    assert all(0.7 <= fill[c] <= 0.76 for c in RD.CHIP_ORDER if c != "PublicValues"), fill
    inputs = np.arange(1, 65, dtype=np.uint64) * 1000003 % F.P
    fri = abi.FriConfig(1, 20, 8)
    for tamper, want in ((False, True), (True, False)):
        recs = host_shard(prog, oracle, inputs)
        if tamper:
            recs[2].prep_trace[1, 5] = F.to_monty((int(F.from_monty(recs[2].prep_trace[1, 5])) + 1) % F.P)
        opk, ch = oracle_key(oracle, recs, 1)
        start = ch.copy()
        pv = prog.public_values(prog.digest)
        proof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], pv, fri, synth.NUM_PV_ELTS, ch)
        assert (oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0) == want
        if not tamper:
            assert np.array_equal(RD.child_words(proof, True)[24:], prog.digest)


def test_witness_binds_the_digest_to_the_inputs(oracle):
    prog = RD.StandinProgram(TINY, 32, 4, oracle_permute(oracle))
    a = np.arange(32, dtype=np.uint64) + 7
    w = prog.witness(a)
    st = np.zeros(16, dtype=np.uint64)
    for k in range(0, 32, 8):
        st[:8] = a[k:k + 8]
        st = F.from_monty(oracle.poseidon2_permute_batch(F.to_monty(st.reshape(1, 16)))[0]).astype(np.uint64)
    assert np.array_equal(w["digest"], st[:8])
    b = a.copy()
    b[17] ^= 1
    assert not np.array_equal(prog.witness(b)["digest"], w["digest"])
    before = {k: v.copy() for k, v in prog.streams.items()}
    prog.witness(b)                                   # a witness never touches the shared streams
    assert all(np.array_equal(before[k], prog.streams[k]) for k in before)
    assert len(w["var_values"]) == 4 * 32 and len(w["poseidon2_events"]) == 32 * 4 and len(w["pv_main"]) == 8


class StubLane:
    """Stands in for a GPU lane: a "proof" whose commitments are a hash of (program, inputs, salt) and whose last eight words are the
    program's real digest of the inputs."""

    def __init__(self, rank, delay=0.0):
        self.rank, self.delay, self.proved = rank, delay, []

    def prove(self, prog_id, prog, shape_idx, fri, inputs, salt=0):
        time.sleep(self.delay)
        w = prog.witness(inputs)
        h = np.random.default_rng([int(x) for x in inputs[:8]] + [salt, shape_idx, fri[0]]).integers(0, F.P, 24, dtype=np.uint64)
        self.proved.append(salt)
        return np.concatenate([F.to_monty(h), np.array([0x5AFE, self.rank, salt, len(inputs)], dtype=np.uint32), F.to_monty(w["digest"])]).astype(np.uint32)


def _tree_worker(rank, world, port, n_core, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from ziren_amd import farm
    f = farm.Farm(backend="gloo")
    tree = RD.ReduceTree(RD.TreePlan(1, 0, 0), oracle_permute(O), shapes=[TINY, dict(TINY, Select=9)])
    core = (np.arange(n_core * 32, dtype=np.uint64).reshape(n_core, 32) * 7919 + 11) % F.P
    lanes = [StubLane(rank, delay=0.02 * rank), StubLane(rank, delay=0.01)]
    streams, words = tree.run(f, lanes, core)
    q.put((rank, [None if s is None else [p.tolist() for p in s] for s in streams], [w.tolist() for w in words], [l.proved for l in lanes]))
    f.close()


@pytest.mark.parametrize("n_core", [5, 8])
def test_tree_runs_level_by_level_over_two_ranks(n_core):
    """World 2 (gloo), two lanes per rank: every node of every layer is proven exactly once by some lane, rank 0 holds every layer's
    streams in node order, every rank holds the same witnessed words, and a parent's digest is the sponge of its children's words."""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_tree_worker, args=(r, 2, port, n_core, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, streams0, words0, proved0), (_, streams1, words1, proved1) = res
    assert words0 == words1 and all(s is None for s in streams1)
    sizes = [n_core] + [len(nodes) for nodes in RD.tree_levels(n_core)] + [1]
    assert [len(s) for s in streams0] == sizes
    salts = sorted(x for lanes in (proved0, proved1) for l in lanes for x in l)
    assert salts == list(range(1, sum(sizes) + 1))              # every node once, nobody twice
    import oracle_lib as O
    permute = oracle_permute(O)

    def sponge(words):
        st = np.zeros(16, dtype=np.uint64)
        for k in range(0, len(words), 8):
            st[:8] = words[k:k + 8]
            st = permute(st.reshape(1, 16)).reshape(16).astype(np.uint64)
        return [int(x) for x in st[:8]]

    core = ((np.arange(n_core * 32, dtype=np.uint64).reshape(n_core, 32) * 7919 + 11) % F.P).tolist()
    below = core
    structure = [[(i,) for i in range(n_core)]] + RD.tree_levels(n_core) + [[(0,)]]
    for nodes, layer_words, layer_streams in zip(structure, words0, streams0):
        for i, ch in enumerate(nodes):
            assert layer_words[i][24:] == sponge([w for c in ch for w in below[c]])
            assert [int(x) for x in RD.child_words(np.array(layer_streams[i], dtype=np.uint32), True)] == layer_words[i]
        below = layer_words


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------------

def device_permute(ctx):
    from ziren_amd import prover
    return lambda v: F.from_monty(prover.poseidon2_permute_batch(ctx, F.to_monty(np.asarray(v, dtype=np.uint64))))


@pytest.mark.gpu
def test_gpu_smallest_reference_shape_bit_exact(hip_ctx, oracle):
    """A compress-machine shard at the reference's fastest shape (shape.rs:136-146: MemoryVar / Select 2^18, BatchFRI / ExpReverseBitsLen
    2^17, Poseidon2Wide / MemoryConst 2^16, the ALUs 2^15) under the compress prover's FRI configuration: preprocessed commitment, every
    device-born trace and the whole proof stream equal to the oracle's, accepted by the restated verifier."""
    shape = RD.load_shapes()[0]
    prog = RD.StandinProgram(shape, 64, 11, device_permute(hip_ctx))
    inputs = (np.arange(64, dtype=np.uint64) * 2654435761 + 99) % F.P
    lane = RD.ReduceLane(hip_ctx)
    proof = lane.prove("t", prog, 0, RD.COMPRESS_FRI, inputs, salt=0).copy()
    hp, recs_dev, pk, ch0 = lane.key_for("t", prog, 0, RD.COMPRESS_FRI)
    recs = host_shard(prog, oracle, inputs)
    born = lane.traces(prog, recs_dev, prog.witness(inputs))
    for m, r in zip(born, recs):
        assert np.array_equal(m.to_host(), r.trace), r.name
        m.free()
    opk, och = oracle_key(oracle, recs, 1)
    assert np.array_equal(pk.commit, opk.commitment())
    start = och.copy()
    fri = abi.FriConfig(*RD.COMPRESS_FRI)
    oproof, _ = oracle.prove_shard(opk, recs, [c.trace for c in recs], prog.public_values(prog.digest), fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(opk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    lane.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape_idx,fri", [(1, RD.COMPRESS_FRI), (2, RD.COMPRESS_FRI), (0, RD.SHRINK_FRI)], ids=["shape1-compress", "shape2-compress", "shape0-shrink"])
def test_gpu_larger_reference_shapes_verify(hip_ctx, oracle, shape_idx, fri):
    """The second and third shape (MemoryVar / Select up to 2^20, BatchFRI 2^21, ExtAlu 2^19, Poseidon2Wide 2^18) and the shrink prover's
    (2, 42) configuration at the first: GPU proofs accepted by the restated verifier; a flipped word is rejected."""
    shape = RD.load_shapes()[shape_idx]
    prog = RD.StandinProgram(shape, 32, 20 + shape_idx, device_permute(hip_ctx))
    inputs = (np.arange(32, dtype=np.uint64) * 40503 + shape_idx) % F.P
    lane = RD.ReduceLane(hip_ctx)
    proof = lane.prove("t", prog, shape_idx, fri, inputs, salt=0).copy()
    recs = host_shard(prog, oracle, inputs)
    opk, och = oracle_key(oracle, recs, fri[0])
    cfg = abi.FriConfig(*fri)
    assert oracle.verify_shard(opk, recs, cfg, synth.NUM_PV_ELTS, och.copy(), proof) == 0
    bad = proof.copy()
    bad[len(bad) // 2] ^= 1
    assert oracle.verify_shard(opk, recs, cfg, synth.NUM_PV_ELTS, och.copy(), bad) != 0
    lane.close()


@pytest.mark.gpu
def test_gpu_eight_leaf_tree_through_the_farm(hip_ctx, oracle):
    """Eight gathered core proofs -> first layer -> three reduce layers -> shrink, on one device, two lanes, through the farm's queue and
    gathers: 8 + 4 + 2 + 1 + 1 proofs; every layer's digests are the sponge of the layer below's words; the root of every layer and the
    shrink proof are accepted by the restated verifier with the transcript the tree gave them."""
    from ziren_amd import farm, prover
    tree = RD.ReduceTree(RD.TreePlan(0, 0, 0), device_permute(hip_ctx))
    rng = np.random.default_rng(5)
    core = rng.integers(0, F.P, (8, 32), dtype=np.uint64)          # stand-ins for child_words of eight core proofs
    ctx2 = prover.Context(0)
    lanes = [RD.ReduceLane(hip_ctx), RD.ReduceLane(ctx2)]
    f = farm.Farm()
    streams, words = tree.run(f, lanes, core)
    assert [len(s) for s in streams] == [8, 4, 2, 1, 1]
    assert len({p.tobytes() for s in streams for p in s}) == 16
    layered, lwords = tree.run(f, lanes, core, pipelined=False)       # the layer-by-layer schedule: the same proofs, word for word
    assert all(np.array_equal(a, b) for x, y in zip(streams, layered) for a, b in zip(x, y)) and all(np.array_equal(a, b) for a, b in zip(words, lwords))
    below = core
    structure = [[(i,) for i in range(8)]] + RD.tree_levels(8) + [[(0,)]]
    salt = 1
    for (name, si, fri, nodes), layer_streams, layer_words in zip(tree.layers(8), streams, words):
        assert nodes == structure[[n for n, *_ in tree.layers(8)].index(name)]
        i = len(nodes) - 1                                            # the layer's last node through the verifier
        prog = tree.program(si, len(nodes[i]))
        inputs = np.concatenate([below[c] for c in nodes[i]])
        recs = host_shard(prog, oracle, inputs)
        opk, och = oracle_key(oracle, recs, fri[0])
        idx = np.array([salt + i], dtype=np.uint32)
        oracle.challenger_observe(och, idx)
        assert oracle.verify_shard(opk, recs, abi.FriConfig(*fri), synth.NUM_PV_ELTS, och, layer_streams[i]) == 0, name
        assert np.array_equal(layer_words[i][24:], prog.digest)
        below, salt = layer_words.astype(np.uint64), salt + len(nodes)
    for l in lanes:
        l.close()
    ctx2.close()


TREE_WORKER = """
import sys, json, os
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from ziren_amd import farm, field as F, prover, reduce as RD
f = farm.Farm(backend="gloo")                      # two processes on ONE device: RCCL refuses that, the collectives are the same calls
ctx = prover.Context(0)
permute = lambda v: F.from_monty(prover.poseidon2_permute_batch(ctx, F.to_monty(np.asarray(v, dtype=np.uint64))))
tree = RD.ReduceTree(RD.TreePlan(0, 0, 0), permute)
core = (np.arange(6 * 32, dtype=np.uint64).reshape(6, 32) * 104729 + 3) %% F.P
lane = RD.ReduceLane(ctx)
streams, words = tree.run(f, [lane], core)
mine = sorted(k[0] for k in lane.keys)             # the programs this rank proved nodes of
out = {"rank": f.rank, "layers": [len(w) for w in words], "digests": [w[:, 24:].tolist() for w in words],
       "streams": None if streams[0] is None else [[int(p[0]) for p in s] for s in streams], "nodes_proved_here": len(lane.setup_ms)}
if f.rank == 0:
    import oracle_lib as O
    from test_reduce import host_shard, oracle_key
    from ziren_amd import abi, synth
    prog = tree.program(0, 1)
    recs = host_shard(prog, O, words[-2][0].astype(np.uint64))             # the shrink node witnesses the root of the reduce layers
    opk, och = oracle_key(O, recs, RD.SHRINK_FRI[0])
    O.challenger_observe(och, np.array([sum(len(w) for w in words)], dtype=np.uint32))      # its salt: the last node of the tree
    out["shrink_verified"] = O.verify_shard(opk, recs, abi.FriConfig(*RD.SHRINK_FRI), synth.NUM_PV_ELTS, och, streams[-1][0]) == 0
print("RESULT " + json.dumps(out), flush=True)
lane.close()
f.close()
"""


@pytest.mark.gpu
def test_gpu_tree_over_two_real_ranks_sharing_the_device(tmp_path):
    """The N > 1 path of the reduce tree with real lanes: two processes (gloo), both proving on this box's one GPU, claim the nodes of every
    layer from the shared queue; both end with the same witnessed words, rank 0 with every layer's streams, and the shrink proof — made by
    whichever rank claimed it — is accepted by the restated verifier."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "tree_worker.py"
    script.write_text(TREE_WORKER % (ROOT, os.path.join(ROOT, "tests")))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    res = sorted((json.loads(next(l for l in o[0].splitlines() if l.startswith("RESULT "))[7:]) for o in outs), key=lambda d: d["rank"])
    assert res[0]["layers"] == res[1]["layers"] == [6, 3, 2, 1, 1]
    assert res[0]["digests"] == res[1]["digests"]
    assert res[1]["streams"] is None and [len(s) for s in res[0]["streams"]] == [6, 3, 2, 1, 1]
    assert res[0]["shrink_verified"] is True


@pytest.mark.gpu
def test_gpu_bench_reduce_tree_line():
    """`bench.py --workload reduce-tree`: one JSON line — per-shape legs (each proof through the restated verifier), a 4-leaf tree under both
    schedules with the same proofs, recursion shards per second as `value`."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "reduce-tree", "--leaves", "4", "--steps", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("recursion-shard-proofs/sec") and d["value"] > 0 and d["verified"] is True and d["n_gpus"] == 1
    red = d["reduce"]
    assert set(red["per_shape"]) == {"shape0_compress_1_84", "shape1_compress_1_84", "shape2_compress_1_84", "shape0_shrink_2_42"}
    for leg in red["per_shape"].values():
        assert leg["verified"] is True and 0 < leg["roofline"]["frac"] < 1 and leg["ms_per_shard_traces_resident"] <= leg["ms_per_shard_from_events_unpipelined"] * 1.05
    t = red["trees"][0]
    assert t["leaves"] == 4 and t["recursion_shards"] == 4 + 2 + 1 + 1 and t["wall_ms"] > 0 and t["layer_by_layer"]["wall_ms"] > 0
    assert "STAND-IN" in red["program"]


@pytest.mark.gpu
def test_gpu_bench_reduce_tree_over_two_ranks_on_one_device():
    """`bench.py --workload reduce-tree --gpus 2` self-launching two ranks that share this box's GPU (ZKM_BENCH_ONE_DEVICE: gloo): one claim
    queue over all nodes, words through the store, one gather per tree, the last tree's root and shrink proofs verified, one line with
    n_gpus = 2 and the whole job's recursion shards per second."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZKM_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "reduce-tree", "--gpus", "2", "--leaves", "3", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_in_process_group"] == 2 and d["verified"] is True and d["value"] > 0
    assert d["recursion_shards_per_tree"] == 6 + 3 + 2 + 1 + 1 and d["steps"] == 2 and abs(d["value"] - 13 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 0.5
    assert d["fewest_nodes_on_a_rank_warmup_included"] >= 1
