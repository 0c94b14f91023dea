#!/usr/bin/env python3
"""How the CPU restatement (oracle/, test infrastructure: the cpu_baseline leg of bench.py) scales with threads on this host:
one SYN-k shard proof per thread count. Usage: cpu_scaling.py [log_rows] [threads,threads,...]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from ziren_amd import abi, synth  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
avail = os.cpu_count() or 1
counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else sorted({avail, avail // 2, 64, 32, 16, 8} & set(range(1, avail + 1)), reverse=True)
fri = abi.FriConfig(1, 84, 16)
L = O.lib()
L.orc_lde_seconds.restype = C.c_double
sh = synth.syn_shard(k)
out = {"log_rows": k, "cores_available": avail, "runs": []}
for t in counts:
    L.orc_set_num_threads(t)
    pk = O.Pk([], [], sh.pc_start, sh.initial_global_cumulative_sum, fri.log_blowup)
    ch = O.new_challenger()
    pk.observe_into(ch)
    L.orc_lde_seconds(C.c_int(1))
    t0 = time.time()
    O.prove_shard(pk, sh.chips, [c.trace for c in sh.chips], sh.public_values, fri, synth.NUM_PV_ELTS, ch)
    dt = time.time() - t0
    out["runs"].append({"threads": t, "seconds": round(dt, 3), "lde_seconds": round(float(L.orc_lde_seconds(C.c_int(0))), 3)})
    print(out["runs"][-1], flush=True)
print(json.dumps(out))
