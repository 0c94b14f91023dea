"""Times Farm.gather_proofs and its pieces under RCCL at world size 1 (forty proof streams of the benchmarked length):
   gpurun -- 'python tools/diag_gather.py'   (EXPERIMENTS.md, round 5: the pageable send buffer that took 25 ms now and then)."""
import os, sys, time
os.environ.update(ZKM_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from ziren_amd import farm as F
f = F.Farm()
torch, dist = f.torch, f.dist
n = 40
proofs = [np.random.randint(0, 2**31, 30716, dtype=np.uint32) for _ in range(n)]
ids = list(range(n))
for rep in range(4):
    t0 = time.perf_counter(); out = f.gather_proofs(ids, proofs, n); t1 = time.perf_counter()
    print("gather_proofs", round((t1 - t0) * 1e3, 2), "ms")
# pieces
lens = np.zeros(n + 1, dtype=np.int64)
for rep in range(3):
    t0 = time.perf_counter(); t = torch.from_numpy(lens).to(f.device); torch.cuda.synchronize(); t1 = time.perf_counter()
    dist.all_reduce(t); torch.cuda.synchronize(); t2 = time.perf_counter()
    l = t.cpu().numpy(); t3 = time.perf_counter()
    mine = np.zeros(n * 30716 + n, dtype=np.uint32)
    off = n
    for p in proofs:
        mine[off:off + len(p)] = p; off += len(p)
    t4 = time.perf_counter()
    buf = torch.from_numpy(mine.view(np.int32)).to(f.device); torch.cuda.synchronize(); t5 = time.perf_counter()
    bufs = [torch.empty_like(buf)]; torch.cuda.synchronize(); t6 = time.perf_counter()
    dist.gather(buf, bufs, dst=0); torch.cuda.synchronize(); t7 = time.perf_counter()
    b = bufs[0].cpu().numpy(); t8 = time.perf_counter()
    print("h2d lens %.2f allreduce %.2f d2h %.2f pack %.2f h2d buf %.2f empty %.2f gather %.2f d2h %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7)))
