#!/usr/bin/env python3
"""N proofs of one compress-machine shard (stand-in program at an allowed shape, traces resident) — the command rocprofv3 wraps for the
recursion shards' kernel trace:  rocprofv3 --kernel-trace --stats -d DIR -o t -- python tools/prof_recursion_shard.py --shape 0 --steps 6"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, default=0)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--shrink", action="store_true")
    args = ap.parse_args()
    import bench_reduce_tree as B
    from ziren_amd import prover, reduce as RD
    ctx = prover.Context(0)
    lane = RD.ReduceLane(ctx)
    prog = RD.StandinProgram(RD.load_shapes()[args.shape], 64, 5, B.device_permute(ctx))
    fri = RD.SHRINK_FRI if args.shrink else RD.COMPRESS_FRI
    inputs = np.arange(64, dtype=np.uint64)
    hp, recs, pk, ch0 = lane.key_for("p", prog, args.shape, fri)
    w = prog.witness(inputs)
    born = lane.traces(prog, recs, w)
    pv = prog.public_values(w["digest"])
    for _ in range(args.steps):
        hp.prove_shard(pk, pv, born, ch0.copy(), out=lane.out)
    ctx.synchronize()
    print("phases", dict(ctx.last_timings()))


if __name__ == "__main__":
    main()
