"""BASELINE config 4 (multi-shard fan-out) on one GPU, twice: the synthetic mapping and the real workload.

(1) N = 8 distinct SYN-21 shards (SURVEY.md 8d's mapping of the keccak multi-shard
workload) go through ziren_amd.farm.Farm — process group on RCCL (ZKM_FORCE_DIST=1, world 1), shards claimed from the shared work queue,
whole proof streams gathered to rank 0 — and every gathered proof is accepted by the restated shard verifier; the gathered table is in
shard order. The N > 1 ranks path of the same code runs over gloo in tests/test_farm_gloo.py.

(2) The workload config 4 names — a guest that hashes with the KECCAK_SPONGE precompile, several shards (examples/keccak-precompile): every
shard of such a run (CPU shards, the Poseidon2 and Keccak precompile shards, the memory shard) is claimed from the farm's queue, generated on
the device from its events and proven; the gathered proofs pass the restated machine verifier (public values chain, global digests sum to
zero)."""
import json
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


MACHINE_WORKER = textwrap.dedent("""
    import sys, json
    import numpy as np
    sys.path.insert(0, %r)
    sys.path.insert(0, %r)
    from ziren_amd import abi, farm, field as F, miniexec as M, prover, synth
    import oracle_lib as O
    import machine_lib as ML
    from test_machine import ZERO_DIGEST, check_machine_airs
    m = %s
    N = len(m.shards)
    f = farm.Farm()
    assert f.dist is not None and f.device.type == "cuda", "the farm must be on RCCL here"
    fri = abi.FriConfig(1, 84, 16)
    ctx = prover.Context(f.local_rank)
    pc_start = F.to_monty(m.pc_base)
    state = {}
    def prove(k):
        dev = ML.Device(ctx)
        dcs = ML.build_shard(dev, m, k)                       # traces born on the device from the shard's events
        hp = prover.HipProver(dcs, fri, synth.NUM_PV_ELTS, ctx=ctx)
        hp.specialize_quotient_kernels(dcs)
        if "pk" not in state:
            state["pk"] = hp.setup([ctx.tracegen_byte_table(), ctx.tracegen_program(m.program, m.pc_base, dcs[-1].log_height)], [0, 0], pc_start, ZERO_DIGEST)
        ch = prover.new_challenger()
        state["pk"].observe_into(ch)
        proof = hp.prove_shard(state["pk"], ML.shard_public_values(m.shards[k]), [c.trace for c in dcs], ch).copy()
        for c in dcs:
            c.trace.free()
        dev.blu.free()
        return proof
    f.barrier()
    ids, proofs = f.run_queue(N, prove)
    got = f.gather_proofs(ids, proofs, N)
    assert len(got) == N
    oshards = check_machine_airs(O, m)
    opk = O.Pk([oshards[0][-2].prep_trace, oshards[0][-1].prep_trace], [0, 0], pc_start, ZERO_DIGEST, fri.log_blowup)
    assert ML.verify_machine(O, opk, oshards, got, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
    assert ML.verify_machine(O, opk, oshards[:-2] + oshards[-1:], got[:-2] + got[-1:], fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is not None
    kinds = [s.kind for s in m.shards]
    print(json.dumps({"ok": True, "shards": N, "kinds": kinds, "chips": sorted({c.name for cs in oshards for c in cs})}))
    f.close()
""")


@pytest.mark.gpu
def test_gpu_keccak_machine_through_the_farm(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "farm_machine_worker.py"
    script.write_text(MACHINE_WORKER % (ROOT, os.path.join(ROOT, "tests"), "M.run_machine(3000, seed=5, shard_cycles=1024, poseidon2_calls=1, keccak_calls=2)"))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKM_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    assert '"ok": true' in r.stdout and '"precompile"' in r.stdout and '"KeccakSponge"' in r.stdout


@pytest.mark.gpu
def test_gpu_run_with_every_chip_through_the_farm(tmp_path):
    """The run of tests/test_all_chips.py — thirty-odd shards of very different shapes, all fifty chips — dealt out by the farm's work queue over
    RCCL (world 1), gathered as proof streams and verified as a machine."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "farm_all_chips_worker.py"
    script.write_text(MACHINE_WORKER % (ROOT, os.path.join(ROOT, "tests"), "__import__('test_all_chips').everything_machine()"))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKM_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and len(out["chips"]) == 50 and out["kinds"].count("precompile") == 27


@pytest.mark.gpu
def test_gpu_bench_queue_mode_over_rccl():
    """`bench.py --gpus N --queue`: the benchmark's multi-GPU line through the code path the farm tests cover (claim queue with one shard
    claimed ahead per lane, two lanes, events prefetched, RCCL gather to rank 0), here with world size 1 on RCCL: six distinct shaped
    fibonacci shards (SHARD_SIZE 2^16), one JSON line, every gathered proof verified."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZKM_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--queue", "6", "--shard-size-log", "16", "--warmup", "1", "--no-extra"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["verified"] is True and d["config"]["shards_in_flight_per_gpu"] == 2 and d["config"]["ranks_in_process_group"] == 1
    assert "nccl" in d["config"]["backend"] and d["event_bytes_per_shard"] > 0
    assert d["steps"] == 6 and d["shards_proved"] == 6 and d["value"] > 0 and d["host_ms_per_shard"]["rank0_mean"] > 0
    # the resident one-lane leg after the timed region: the per-kernel objects of the same line
    assert d["resident_one_lane"]["value"] > 0 and d["roofline"]["kernel"] and 0 < d["roofline"]["frac"] < 1
    src = d["kernels_ms_source"]
    assert src["kernels_ms_sum"] <= src["ms_per_step_of_that_pass"] and "overlap off" in src["pass"]
    assert d["lib_digest"] and d["verified_proofs"]["checked_by_the_verifier"] == 6 and d["cpu_baseline"] is None      # N = 1: every gathered proof


@pytest.mark.gpu
def test_gpu_bench_two_ranks_on_one_device_share_the_queue():
    """`bench.py --gpus 2` self-launching two ranks that both prove on this box's one GPU (ZKM_BENCH_ONE_DEVICE: process group over gloo, since
    RCCL wants a device per rank): two processes, four lanes, one claim counter in the store, events prefetched, proofs gathered to rank 0,
    every lane's last proof verified — the N > 1 line's code path with real proofs."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ZKM_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard-size-log", "16", "--steps", "4", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_in_process_group"] == 2 and d["shards"] == 8 and d["shards_proved"] == 8 and d["steps"] == 4
    assert d["verified"] is True and d["fewest_shards_on_a_rank"] >= 1 and d["config"]["shards_in_flight_per_gpu"] == 2
    assert 2 <= d["verified_proofs"]["checked_by_the_verifier"] <= 4 and d["resident_one_lane"]["value"] > 0 and d["roofline"]["kernel"] and d["cpu_baseline"] is None    # every lane that proved something
