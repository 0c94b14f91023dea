"""bench.py's command line where it does not need a GPU: `--gpus N` must mean N.

The prover is stubbed (ZKM_BENCH_STUB_PROVER=1 -> bench.StubLane, gloo): what runs is the real launch, claim queue with one shard claimed
ahead, gather to rank 0 and the JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2"], {"ZKM_BENCH_STUB_PROVER": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_in_process_group"] == 2 and line["stub"] is True
    assert line["steps"] == 10 and line["shards"] == 20 and line["shards_proved"] == 20 and line["scaling"] == "weak"          # --steps (10) shards per GPU
    assert line["fewest_shards_on_a_rank"] >= 1 and line["verified"] is True
    assert "claim queue" in line["config"]["parallelism"]


def test_n1_and_n2_lines_are_one_experiment_with_the_same_keys():
    """`--gpus 1` runs the same claim-queue path as `--gpus N > 1` — same keys in the line, same config fields, two
    shards in flight per GPU, --steps shards per GPU — so value(N) / (N value(1)) compares like with like."""
    lines = {}
    for n in (1, 2):
        r = _run(["--gpus", str(n), "--steps", "6", "--warmup", "1"], {"ZKM_BENCH_STUB_PROVER": "1"})
        assert r.returncode == 0, r.stderr[-2000:]
        out = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(out) == 1, r.stdout
        lines[n] = json.loads(out[0])
    a, b = lines[1], lines[2]
    assert list(a.keys()) == list(b.keys())
    assert set(a["config"].keys()) == set(b["config"].keys())
    for line, n in ((a, 1), (b, 2)):
        assert line["n_gpus"] == n and line["steps"] == 6 and line["shards"] == 6 * n and line["shards_proved"] == 6 * n and line["warmup"] == 1
        assert line["config"]["shards_in_flight_per_gpu"] == 2 and line["config"]["ranks_in_process_group"] == n
        assert "claim queue" in line["config"]["parallelism"] and line["scaling"] == "weak"
        assert abs(line["ms_per_step"] * line["steps"] / 1e3 - line["wall_s"]["timed_region"]) < 2e-3           # K steps fit the timed region exactly
        for key in ("roofline", "cpu_baseline", "resident_one_lane", "kernels_ms", "lde", "valu", "verified_proofs"):
            assert key in line
    assert a["config"]["backend"] == "none (one process)" and b["config"]["backend"] == "gloo"
    assert a["verified_proofs"]["of"] == 6 and a["verified_proofs"]["checked_by_the_verifier"] == 6        # N = 1: every gathered proof
    assert b["verified_proofs"]["checked_by_the_verifier"] == 4                                            # N > 1: every lane's last proof on every rank


def test_steps_means_exactly_that_many_shards_per_gpu():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "1"], {"ZKM_BENCH_STUB_PROVER": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["steps"] == 1 and line["shards"] == 1 and line["shards_proved"] == 1


def test_gpus_must_agree_with_the_process_group():
    """A launcher that sets WORLD_SIZE=1 for a `--gpus 2` command gets an error, never a line that says n_gpus 1."""
    r = _run(["--gpus", "2"], {"ZKM_BENCH_STUB_PROVER": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not r.stdout.strip()
    assert "refusing" in r.stderr


def test_under_torchrun_style_env_the_queue_runs_with_three_ranks():
    """The driver's way: every rank started by a launcher with RANK / WORLD_SIZE / MASTER_* set (here by hand, world 3, strong scaling
    over 7 shards: the ranks cannot all get the same number)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(3):
        env = {k: v for k, v in os.environ.items()}
        env.update({"ZKM_BENCH_STUB_PROVER": "1", "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": "3", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--queue", "7"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert not outs[1][0].strip() and not outs[2][0].strip()           # only rank 0 prints
    assert line["n_gpus"] == 3 and line["steps"] == 7 and line["shards_proved"] == 7 and line["scaling"] == "strong"
