import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
import bench
from ziren_amd import abi, lib
fri = abi.FriConfig(1, 84, 16)
lane = bench.FibQueueLane(0, 0, fri, 18, True)
def threads():
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        f = open(f"/proc/self/task/{tid}/stat").read()
        name = f[f.index("(") + 1:f.rindex(")")]
        rest = f[f.rindex(")") + 2:].split()
        out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tick)
    return out
lane.prove(0, lane.prefetch(0))
a = threads(); t0 = time.time()
for i in range(40):
    lane.prove(i, lane.prefetch(i))
b = threads(); dt = time.time() - t0
print("wall", dt, "main tid", os.getpid())
for tid in b:
    d = b[tid][1] - a.get(tid, ("", 0))[1]
    if d > 0.01:
        print(tid, b[tid][0], round(d / dt, 2), "cores")
        try:
            print("   wchan:", open(f"/proc/self/task/{tid}/wchan").read(), " stack:", open(f"/proc/self/task/{tid}/stack").read()[:300])
        except Exception as e:
            print("   ", e)
os.system(f"cat /proc/{os.getpid()}/task/*/comm | sort | uniq -c")
