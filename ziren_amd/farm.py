"""Shard farm: independent shards dealt to one prover process per GPU.

Each shard's commit+open depends only on (pk, its traces, a clone of the post-pk challenger)
(crates/core/machine/src/utils/prove.rs:208-209,492-497), so there is no data-path collective:
rank r proves shards r, r + world, r + 2*world, ... The process group is used only for the
barrier around the timed region, the max-over-ranks time, and (optionally) gathering the small
per-shard commitments so rank 0 can hand the proofs to the recursion tree
(crates/prover/src/lib.rs:617-957) in shard order.
"""
import contextlib
import os
import sys
import time
from typing import Callable, List, Optional, Sequence

import numpy as np


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on stdout when the first communicator is created; bench.py's stdout
    must carry exactly one JSON line, so native-level stdout is pointed at stderr meanwhile."""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)   # the banner sits in the C library's stdout buffer: push it out while fd 1 still points at stderr
        os.dup2(saved, 1)
        os.close(saved)


def shards_for_rank(n_shards: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: shard i goes to rank i % world (the reference's phase-2 channel deals
    records to the next free prover thread, prove.rs:484; with equal-cost shards this is the same)."""
    return list(range(rank, n_shards, world))


class Farm:
    """Barrier / max-reduce / gather over torch.distributed (RCCL on GPUs, gloo on CPU tests)."""

    def __init__(self, backend: Optional[str] = None, device_sync: Optional[Callable[[], None]] = None):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.dist = None
        self.device = None
        self.device_sync = device_sync or (lambda: None)
        if self.world > 1 or os.environ.get("ZKM_FORCE_DIST") == "1":
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                with _stdout_to_stderr():
                    dist.init_process_group(backend="nccl", device_id=self.device)
                    dist.barrier()  # creates the communicator now (and prints RCCL's banner to stderr)
                    torch.cuda.synchronize()
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend=backend)
            self.dist = dist
            self.torch = torch

    def barrier(self):
        self.device_sync()
        if self.dist is not None:
            self.dist.barrier()
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_commitments(self, shard_ids: Sequence[int], commits: Sequence[np.ndarray], n_shards: int):
        """All ranks contribute (shard id, 24-word commitment triple); returns on every rank the table in
        shard order. A few hundred bytes per shard — point-to-point over xGMI would be equally fine."""
        table = np.zeros((n_shards, 24), dtype=np.int64)
        for i, c in zip(shard_ids, commits):
            table[i] = np.asarray(c, dtype=np.int64)[:24]
        if self.dist is None:
            return table.astype(np.uint32)
        t = self.torch.from_numpy(table).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)  # disjoint rows: sum == union
        return t.cpu().numpy().astype(np.uint32)

    def timed(self, step: Callable[[], None], steps: int, warmup: int) -> float:
        """W untimed warm-up steps, then exactly K steps between barriers; max over ranks (seconds)."""
        for _ in range(warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.dist is not None:
            with _stdout_to_stderr():   # anything RCCL prints while tearing down stays off stdout as well
                self.dist.destroy_process_group()
            self.dist = None
