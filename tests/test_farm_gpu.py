"""BASELINE config 4 (multi-shard fan-out) on one GPU: N = 8 distinct SYN-21 shards (SURVEY.md 8d's mapping of the keccak multi-shard
workload) go through ziren_amd.farm.Farm — process group on RCCL (ZKM_FORCE_DIST=1, world 1), shards claimed from the shared work queue,
whole proof streams gathered to rank 0 — and every gathered proof is accepted by the restated shard verifier; the gathered table is in
shard order. The N > 1 ranks path of the same code runs over gloo in tests/test_farm_gloo.py."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, json
    import numpy as np
    sys.path.insert(0, %r)
    sys.path.insert(0, %r)
    from ziren_amd import abi, farm, prover, synth
    import oracle_lib as O
    N, K = 8, 21
    hold = {}
    f = farm.Farm(device_sync=lambda: hold["hp"].ctx.synchronize() if "hp" in hold else None)
    assert f.dist is not None and f.device.type == "cuda", "the farm must be on RCCL here"
    fri = abi.FriConfig(1, 84, 16)
    shape = synth.syn_shard(K, with_trace=False)
    hp = prover.HipProver(shape.chips, fri, synth.NUM_PV_ELTS, device=f.local_rank)
    hold["hp"] = hp
    pk = hp.setup([], [], shape.pc_start, shape.initial_global_cumulative_sum)
    start = prover.new_challenger()
    pk.observe_into(start)
    pvs = {}
    def prove(i):
        sh = synth.syn_shard(K, seed=0x5A4B4D00 + 7919 * i)          # a distinct witness per shard
        pvs[i] = sh.public_values
        tr = hp.upload_traces([c.trace for c in sh.chips])
        proof = hp.prove_shard(pk, sh.public_values, tr, start.copy()).copy()
        for t in tr:
            t.free()
        return proof
    f.barrier()
    ids, proofs = f.run_queue(N, prove)
    assert ids == list(range(N))                                      # one rank: it claims every shard, in order
    table = f.gather_commitments(ids, [p[:24] for p in proofs], N)
    got = f.gather_proofs(ids, proofs, N)
    assert len(got) == N and all(np.array_equal(got[i], proofs[i]) for i in range(N))
    assert all(np.array_equal(table[i], proofs[i][:24]) for i in range(N))
    assert len({bytes(p[:8]) for p in got}) == N                      # eight different main commitments
    opk = O.Pk([], [], shape.pc_start, shape.initial_global_cumulative_sum, fri.log_blowup)
    ostart = O.new_challenger()
    opk.observe_into(ostart)
    for i, p in enumerate(got):
        assert O.verify_shard(opk, shape.chips, fri, synth.NUM_PV_ELTS, ostart.copy(), p) == 0, i
        assert np.array_equal(p[-231:], np.asarray(pvs[i], dtype=np.uint32)[:231])
    bad = got[3].copy()
    bad[40] ^= 1
    assert O.verify_shard(opk, shape.chips, fri, synth.NUM_PV_ELTS, ostart.copy(), bad) != 0
    print(json.dumps({"ok": True, "shards": N, "proof_words": int(len(got[0]))}))
    f.close()
""")


@pytest.mark.gpu
def test_gpu_eight_shard_farm_over_rccl(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "farm_worker.py"
    script.write_text(WORKER % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKM_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    assert '"ok": true' in r.stdout
