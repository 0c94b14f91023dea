"""CPU, world_size 2, gloo: the N > 1 path of bench.py / the shard farm (assignment, barrier-bracketed
timing with max over ranks, commitment gather). The prover itself needs a GPU; here the step is a stub."""
import os
import socket
import subprocess
import sys
import textwrap

from ziren_amd import farm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_robin_assignment_covers_every_shard_once():
    for n, w in [(8, 1), (8, 2), (7, 4), (3, 8), (32, 8)]:
        seen = sorted(i for r in range(w) for i in farm.shards_for_rank(n, r, w))
        assert seen == list(range(n))


WORKER = textwrap.dedent("""
    import sys, time, json
    import numpy as np
    sys.path.insert(0, %r)
    from ziren_amd import farm
    f = farm.Farm(backend="gloo")
    assert f.world == 2
    n_shards = 5
    mine = farm.shards_for_rank(n_shards, f.rank, f.world)
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.02 * (1 + f.rank))          # rank 1 is slower: max-over-ranks must see it
    elapsed = f.timed(step, steps=3, warmup=1)
    assert len(calls) == 4
    assert elapsed >= 3 * 0.04 * 0.9, elapsed       # the slow rank's time, on both ranks
    commits = [np.full(24, 1000 * i + 7, dtype=np.uint32) for i in mine]
    table = f.gather_commitments(mine, commits, n_shards)
    assert table.shape == (n_shards, 24)
    assert [int(table[i, 0]) for i in range(n_shards)] == [1000 * i + 7 for i in range(n_shards)]
    total = f.sum_over_ranks(float(len(mine)))
    assert total == n_shards
    # the work queue: shards go to whoever is free; rank 1 is three times slower, so rank 0 proves more of them; every shard exactly once;
    # whole proof streams (different lengths) arrive at rank 0 in shard order
    def prove(i):
        time.sleep(0.01 * (1 + 2 * f.rank))
        return np.arange(100 + 7 * i, dtype=np.uint32) + 1000 * i
    f.barrier()
    ids, proofs = f.run_queue(12, prove)
    counts = f.sum_over_ranks(float(len(ids)))
    assert counts == 12
    got = f.gather_proofs(ids, proofs, 12)
    if f.rank == 0:
        assert len(ids) > 6, ids
        assert all(np.array_equal(got[i], np.arange(100 + 7 * i, dtype=np.uint32) + 1000 * i) for i in range(12))
    else:
        assert got is None and 0 < len(ids) < 6
    # a second batch on the same farm and queue name: a fresh counter, every shard dealt again (one run_queue per batch of records)
    ids2, proofs2 = f.run_queue(7, lambda i: np.full(3 + i, 50 + i, dtype=np.uint32))
    assert f.sum_over_ranks(float(len(ids2))) == 7
    got2 = f.gather_proofs(ids2, proofs2, 7)
    if f.rank == 0:
        assert [len(p) for p in got2] == [3 + i for i in range(7)] and all(int(p[0]) == 50 + i for i, p in enumerate(got2))
    assert len(f.host_ms) == len(ids2) and all(ms >= 0 for ms in f.host_ms)
    # a shard nobody proved is an error on every rank (not a hang of the ranks that wait in the gather)
    try:
        f.gather_proofs(ids2, proofs2, 8)
        raise SystemExit("a missing shard went unnoticed")
    except RuntimeError as e:
        assert "no rank" in str(e)
    # words above 2^31 survive the 32-bit transport
    hi = f.gather_proofs([f.rank], [np.array([0xFFFFFFFE - f.rank, 0x80000000], dtype=np.uint32)], 2)
    if f.rank == 0:
        assert [int(x) for x in hi[1]] == [0xFFFFFFFD, 0x80000000]
    if f.rank == 0:
        print(json.dumps({"elapsed": elapsed, "ok": True}))
    f.close()
""")

STRAGGLER = textwrap.dedent("""
    import sys, time, json
    import numpy as np
    sys.path.insert(0, %r)
    from ziren_amd import farm
    f = farm.Farm(backend="gloo")
    assert f.world == 4
    # sixteen shards (4 per rank if dealt statically), rank 3 ten times slower than the others: the queue gives it fewer
    def prove(i):
        time.sleep(0.1 if f.rank == 3 else 0.01)
        return np.full(10 + i, i, dtype=np.uint32)
    f.barrier()
    t0 = time.perf_counter()
    ids, proofs = f.run_queue(16, prove)
    f.barrier()
    elapsed = f.max_over_ranks(time.perf_counter() - t0)
    got = f.gather_proofs(ids, proofs, 16)
    n3 = f.sum_over_ranks(float(len(ids)) if f.rank == 3 else 0.0)
    assert f.sum_over_ranks(float(len(ids))) == 16
    assert n3 <= 2, n3
    assert elapsed < 0.35, elapsed                 # a static deal would take 4 x 0.1 s on the slow rank
    if f.rank == 0:
        assert all(len(got[i]) == 10 + i and (got[i] == i).all() for i in range(16))
        print(json.dumps({"ok": True, "elapsed": elapsed, "slow_rank_shards": n3}))
    f.close()
""")


def _run_world(tmp_path, source, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(source % ROOT)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert '"ok": true' in outs[0][0]


def test_two_rank_farm_over_gloo(tmp_path):
    _run_world(tmp_path, WORKER, 2)


def test_four_rank_farm_with_a_straggler_over_gloo(tmp_path):
    """The claim queue against a static deal: one of four ranks is ten times slower (prove.rs:484: records go to the next free prover)."""
    _run_world(tmp_path, STRAGGLER, 4)


def test_a_second_batch_without_a_process_group_gets_a_fresh_counter():
    f = farm.Farm()
    assert f.dist is None
    for n in (5, 3):
        ids, proofs = f.run_queue(n, lambda i: [i, i + 1])
        assert ids == list(range(n)) and [int(p[0]) for p in f.gather_proofs(ids, proofs, n)] == list(range(n))


def test_lanes_share_one_queue_claim_ahead_and_surface_a_lane_s_failure():
    """Several lanes (host threads) of one rank: every shard proven exactly once, each lane's `prefetch(j)` called for shard j before its
    `prove(j, handle)` and handed over unchanged; a lane that raises ends the batch with that exception (not a hang, not a silent loss) and
    the next batch still gets a fresh counter."""
    import threading
    import time
    import numpy as np
    f = farm.Farm()
    lock = threading.Lock()
    prefetched, proved = [], []

    def make_lane(tag, delay):
        def prefetch(i):
            with lock:
                prefetched.append(i)
            return (tag, i)

        def prove(i, handle):
            assert handle == (tag, i)
            with lock:
                assert i in prefetched
                proved.append(i)
            time.sleep(delay)
            return np.array([i, i + 1], dtype=np.uint32)
        return prove, prefetch

    ids, proofs = f.run_queue(17, lanes=[make_lane("a", 0.002), make_lane("b", 0.005), make_lane("c", 0.0)])
    assert sorted(ids) == list(range(17)) and sorted(proved) == list(range(17)) and len(f.host_ms) == 17
    assert all(int(p[0]) == i for i, p in zip(ids, proofs))

    def bad_prove(i, handle):
        if i == 3:
            raise RuntimeError("shard 3 is broken")
        return np.array([i], dtype=np.uint32)

    try:
        f.run_queue(8, lanes=[(bad_prove, lambda i: i), (bad_prove, lambda i: i)])
        raise AssertionError("the lane's failure was swallowed")
    except RuntimeError as e:
        assert "shard 3" in str(e)
    ids, _ = f.run_queue(4, lanes=[(lambda i, h: np.array([i], dtype=np.uint32), lambda i: i)])
    assert sorted(ids) == [0, 1, 2, 3]
