import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np
import bench_reduce_tree as B
from ziren_amd import prover, reduce as RD
ctx = prover.Context(0)
lane = RD.ReduceLane(ctx)
for si in (0, 2):
    prog = RD.StandinProgram(RD.load_shapes()[si], 64, 5, B.device_permute(ctx))
    inputs = np.arange(64, dtype=np.uint64)
    lane.prove(("p", si), prog, si, RD.COMPRESS_FRI, inputs)
    for rep in range(3):
        t0 = time.perf_counter(); w = prog.witness(inputs); t1 = time.perf_counter()
        pins = lane._pinned(prog)
        ts = {}
        dev = {}
        for ek, copies in pins.items():
            if ek == "_turn": continue
            a = time.perf_counter()
            dev[ek] = ctx.events_upload_async(copies[0])
            ts[ek] = (round((time.perf_counter() - a) * 1e6), copies[0].nbytes >> 10)
        ctx.synchronize()
        for d in dev.values(): d.free()
        print(si, "witness us", round((t1 - t0) * 1e6), ts)
