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
    if f.rank == 0:
        print(json.dumps({"elapsed": elapsed, "ok": True}))
    f.close()
""")


def test_two_rank_farm_over_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert '"ok": true' in outs[0][0]
