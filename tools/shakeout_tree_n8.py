#!/usr/bin/env python3
"""The reduce tree's N > 1 path with real lanes and EIGHT processes, all proving on this box's one GPU (gloo: RCCL wants a device per rank):
the claim counter and the children's words go through the process group's store, every rank proves what it claims, rank 0 gathers all
proofs once at the end. What a one-GPU box can show of an eight-GPU node's host side: that nothing deadlocks or starves, how the nodes
spread over the ranks, what the store traffic costs.   python tools/shakeout_tree_n8.py [--ranks 8] [--leaves 32]"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import sys, json, os, time
import numpy as np
sys.path.insert(0, %r)
from ziren_amd import farm, field as F, prover, reduce as RD
import ctypes as C
from ziren_amd import lib
f = farm.Farm(backend="gloo")
ctx = prover.Context(0)
lib.load().zkm_ctx_set_host_wait(ctx.h, C.c_int(1))
permute = lambda v: F.from_monty(prover.poseidon2_permute_batch(ctx, F.to_monty(np.asarray(v, dtype=np.uint64))))
tree = RD.ReduceTree(RD.TreePlan(1, 0, 0), permute)
leaves = %d
core = (np.arange(leaves * 32, dtype=np.uint64).reshape(leaves, 32) * 104729 + 3) %% F.P
lane = RD.ReduceLane(ctx)
times = []
for rep in range(3):
    f.barrier()
    t0 = time.perf_counter()
    c0 = time.process_time()
    streams, words = tree.run(f, [lane], core)
    f.barrier()
    times.append((time.perf_counter() - t0, time.process_time() - c0))
print("RESULT " + json.dumps({"rank": f.rank, "wall_s": [round(t, 4) for t, _ in times], "cpu_s": [round(c, 3) for _, c in times], "nodes_proved": lane.host_s["nodes"],
                              "root": words[-1][0][24:].tolist(), "gathered": None if streams[0] is None else [len(s) for s in streams]}), flush=True)
lane.close(); f.close()
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--leaves", type=int, default=32)
    args = ap.parse_args()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = "/tmp/zkm_tree_worker.py"
    open(script, "w").write(WORKER % (ROOT, args.leaves))
    procs = []
    t0 = time.time()
    for r in range(args.ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    if any(p.returncode for p in procs):
        sys.exit("a rank failed:\n" + "\n".join(o[1][-800:] for o in outs))
    res = sorted((json.loads(next(l for l in o[0].splitlines() if l.startswith("RESULT "))[7:]) for o in outs), key=lambda d: d["rank"])
    assert all(r["root"] == res[0]["root"] for r in res)
    print(json.dumps({"what": f"{args.ranks} processes sharing ONE MI355X (gloo), one lane each, a {args.leaves}-leaf reduce tree three times (the third is quoted)",
                      "tree_wall_ms": round(1e3 * max(r["wall_s"][-1] for r in res), 1), "first_tree_wall_ms_with_keys_and_tables": round(1e3 * max(r["wall_s"][0] for r in res), 1),
                      "nodes_proved_per_rank_over_the_three_trees": [r["nodes_proved"] for r in res], "cpu_seconds_per_rank_third_tree": [r["cpu_s"][-1] for r in res],
                      "proofs_gathered_on_rank_0_per_layer": res[0]["gathered"], "same_root_digest_on_every_rank": True, "seconds_in_all": round(time.time() - t0, 1)}, indent=1))


if __name__ == "__main__":
    main()
