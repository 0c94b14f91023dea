"""Shard farm: independent shards dealt to one prover process per GPU.

Each shard's commit+open depends only on (pk, its traces, a clone of the post-pk challenger)
(crates/core/machine/src/utils/prove.rs:208-209,492-497), so there is no data-path collective. Shards are dealt the way the
reference's phase-2 channel deals records to the next free prover thread (prove.rs:484): every rank claims the next
unclaimed shard from one shared counter when it becomes free (`Farm.claim` — an atomic add on the process group's store; a
static round-robin, `shards_for_rank`, is kept for equal-cost benchmark shards). The process group is otherwise used for the
barrier around the timed region, the max-over-ranks time, and gathering the finished ShardProof streams to rank 0, which
hands them to the recursion tree (crates/prover/src/lib.rs:617-957) in shard order.
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
        import threading
        self._claim_lock = threading.Lock()
        self._epochs = {}
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
                with _stdout_to_stderr():       # gloo announces its peers on stdout
                    dist.init_process_group(backend=backend)
                    dist.barrier()
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

    def gather_words(self, shard_ids: Sequence[int], rows: Sequence[np.ndarray], n_shards: int, words: int) -> np.ndarray:
        """All ranks contribute (shard id, `words` 32-bit words); returns on every rank the (n_shards, words) table in shard order. A few
        hundred bytes per shard (one small all-reduce): what the next layer of the recursion tree witnesses of this one
        (ziren_amd/reduce.py) — point-to-point over xGMI would be equally fine."""
        table = np.zeros((n_shards, words), dtype=np.int64)
        for i, c in zip(shard_ids, rows):
            table[i] = np.asarray(c, dtype=np.int64)[:words]
        if self.dist is None:
            return table.astype(np.uint32)
        t = self.torch.from_numpy(table).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)  # disjoint rows: sum == union
        return t.cpu().numpy().astype(np.uint32)

    def gather_commitments(self, shard_ids: Sequence[int], commits: Sequence[np.ndarray], n_shards: int):
        """The (main, permutation, quotient) commitment triple of every shard, on every rank."""
        return self.gather_words(shard_ids, commits, n_shards, 24)

    # ---- dealing shards to the next free rank (prove.rs:484) ---------------------------------------------------------------
    def _queue_key(self, queue: str) -> str:
        """One counter per batch: `run_queue` opens a new epoch of `queue` every time it is called (every rank calls it the same
        number of times, so the epochs agree without a message), and a key nobody has added to yet counts from zero."""
        return f"zkm_farm_{queue}_{self._epochs.get(queue, 0)}"

    def claim(self, queue: str = "shards") -> int:
        """The index of the next unclaimed shard of the queue's current batch: an atomic fetch-and-add on a counter every rank shares
        (the process group's key-value store; a plain counter without one). Whoever is free first gets the next shard."""
        key = self._queue_key(queue)
        with self._claim_lock:                 # several lanes (host threads) of this rank may claim
            if self.dist is None:
                self._local_counters = getattr(self, "_local_counters", {})
                self._local_counters[key] = self._local_counters.get(key, 0) + 1
                return self._local_counters[key] - 1
            store = self.dist.distributed_c10d._get_default_store()
            return int(store.add(key, 1)) - 1

    def store(self):
        """(the process group's key-value store or None, the lock every store call of this process goes through). A c10d store client is one
        connection: calls from several lanes are serialised, and nothing may block inside one (ziren_amd/reduce.py's board polls `check`)."""
        return (self.dist.distributed_c10d._get_default_store() if self.dist is not None else None), self._claim_lock

    def open_epoch(self, queue: str) -> int:
        """The number of the batch `queue` is in (every rank calls run_queue / a tree the same number of times, so the numbers agree)."""
        return self._epochs.get(queue, 0)

    def close_epoch(self, queue: str):
        self._epochs[queue] = self._epochs.get(queue, 0) + 1

    def prime_queue(self, n_shards: int, lanes, queue: str = "shards"):
        """Before a timed `run_queue`: every lane claims its first shard of the batch and queues its prefetch now, so that the batch starts
        the way it continues — each lane's next input already crossing PCIe (or landed) when its proof starts. The caller synchronises
        (barrier) between this and `run_queue`, which then begins with these claims instead of making its own."""
        primed = []
        for pv, pf in lanes:
            cur = self.claim(queue)
            primed.append((cur, pf(cur) if (pf is not None and cur < n_shards) else None))
        self._primed = getattr(self, "_primed", {})
        self._primed[self._queue_key(queue)] = primed

    def run_queue(self, n_shards: int, prove: Optional[Callable] = None, queue: str = "shards", prefetch: Optional[Callable] = None, lanes=None):
        """Prove shards of one batch until its queue is empty; returns ([shard ids this rank proved], [their proof streams]). Calling
        it again deals a new batch (a fresh counter). `self.host_ms` collects, per shard proven here, the wall-clock milliseconds of
        its `prove` call (what a rank spends per shard including host work: the number that limits an 8-GPU node once events, not
        resident traces, are the input).

        With `prefetch`, the rank claims one shard ahead: `prefetch(j)` is called for the next shard before `prove(i, handle_i)` of the
        current one, so the next shard's input (the executor's events) crosses PCIe while the current one is proven — the records and
        traces channel of the reference holds one record ahead of the prover the same way (crates/stark/src/opts.rs:11,
        DEFAULT_RECORDS_AND_TRACES_CHANNEL_CAPACITY = 1).

        `lanes` = [(prove, prefetch), ...]: several shards in flight on this rank's GPU, one host thread per lane (each lane its own
        context), all claiming from the same queue — the reference's `shard_batch_size` prover threads (prove.rs:487-497) on one device:
        one lane's transcript round trips and launch gaps are filled by the other's kernels."""
        import threading
        lanes = lanes or [(prove, prefetch)]
        results = [None] * len(lanes)
        lane_cpu = [0.0] * len(lanes)
        errors = []
        primed = getattr(self, "_primed", {}).pop(self._queue_key(queue), None)
        if primed is not None and len(primed) != len(lanes):
            raise ValueError("prime_queue was called with another number of lanes")

        def work(j):
            pv, pf = lanes[j]
            ids, proofs, ms = [], [], []
            results[j] = (ids, proofs, ms)
            cpu_start = time.thread_time()
            try:
                if primed is not None:
                    cur, handle = primed[j]
                else:
                    cur = self.claim(queue)
                    handle = pf(cur) if (pf is not None and cur < n_shards) else None
                while cur < n_shards:
                    t0 = time.perf_counter()
                    nxt, nxt_handle = n_shards, None
                    if pf is not None:
                        nxt = self.claim(queue)
                        if nxt < n_shards:
                            nxt_handle = pf(nxt)
                    ids.append(cur)
                    proofs.append(np.asarray(pv(cur, handle) if pf is not None else pv(cur), dtype=np.uint32).copy())
                    ms.append(1e3 * (time.perf_counter() - t0))
                    if pf is None:
                        nxt = self.claim(queue)
                    cur, handle = nxt, nxt_handle
            except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread)
                errors.append(e)
            finally:
                lane_cpu[j] = time.thread_time() - cpu_start      # CPU seconds of this lane's host thread (Python, ctypes, waiting inside the library)

        try:
            if len(lanes) == 1:
                work(0)
            else:
                ts = [threading.Thread(target=work, args=(j,)) for j in range(len(lanes))]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
        finally:
            self._epochs[queue] = self._epochs.get(queue, 0) + 1
        if errors:
            raise errors[0]
        self.host_ms = [m for r in results for m in r[2]]
        self.lane_cpu_s = lane_cpu
        return [i for r in results for i in r[0]], [p for r in results for p in r[1]]

    def gather_proofs(self, shard_ids: Sequence[int], proofs: Sequence[np.ndarray], n_shards: int, copy: bool = True) -> Optional[List[np.ndarray]]:
        """Whole ShardProof streams to rank 0, in shard order (the input of the recursion tree, lib.rs:617-641). Streams differ in
        length: the lengths are exchanged first (one small all-reduce), then every rank sends one buffer — the ids it holds and their
        words end to end, as 32-bit words — to rank 0 only (`gather`; a few MB per proof, far below one xGMI link's bandwidth, once
        per batch). Returns the list on rank 0, None elsewhere; every rank raises if some shard was proven by nobody. `copy=False`: the
        streams are views of the landing buffer, valid until the next call (rank 0 of an eight-GPU batch otherwise copies 160 streams, 0.1 ms
        each, before it may stop the clock)."""
        if self.dist is None:
            out = [None] * n_shards
            for i, p in zip(shard_ids, proofs):
                out[i] = np.asarray(p, dtype=np.uint32)
            if any(p is None for p in out):
                raise RuntimeError("a shard was proven by no rank")
            return out
        torch, dist = self.torch, self.dist
        trace = os.environ.get("ZKM_FARM_TRACE") == "1"
        marks = [("start", time.perf_counter())]
        mark = (lambda name: marks.append((name, time.perf_counter()))) if trace else (lambda name: None)
        lens = np.zeros(n_shards + self.world, dtype=np.int64)     # [length of every shard's stream] ++ [words held per rank]
        for i, p in zip(shard_ids, proofs):
            lens[i] = len(p)
        lens[n_shards + self.rank] = int(sum(len(p) for p in proofs))
        t = torch.from_numpy(lens).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)           # disjoint entries: every rank learns every length
        lens = t.cpu().numpy()
        mark("lengths all-reduced")
        if (lens[:n_shards] == 0).any():                   # known on every rank: nobody is left waiting in the gather
            raise RuntimeError("a shard was proven by no rank")
        cap = int(lens[n_shards:].max())
        # [ids this rank holds (0xffffffff padded)] ++ [their words, concatenated]; on a GPU the send buffer is page-locked and kept: a fresh
        # pageable array of a few MB now and then takes 25 ms to cross PCIe (first-touch faults under the staging copy), inside the timed region
        pinned = None
        if self.device.type == "cuda":
            pinned = getattr(self, "_send_pin", None)
            if pinned is None or pinned.numel() < cap + n_shards:
                pinned = self._send_pin = torch.empty(max(cap + n_shards, 1 << 20), dtype=torch.int32, pin_memory=True)
            mine = pinned.numpy()[:cap + n_shards].view(np.uint32)      # what lies behind this rank's own words is never read
        else:
            mine = np.empty(cap + n_shards, dtype=np.uint32)
        mine[:n_shards] = 0xFFFFFFFF
        mine[:len(shard_ids)] = shard_ids
        off = n_shards
        for p in proofs:
            mine[off:off + len(p)] = np.asarray(p, dtype=np.uint32)
            off += len(p)
        buf = (pinned[:cap + n_shards].to(self.device, non_blocking=True) if pinned is not None else torch.from_numpy(mine.view(np.int32)).to(self.device))
        mark("packed, copy queued")
        bufs = [torch.empty_like(buf) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(buf, bufs, dst=0)
        mark("gather queued")
        if self.rank != 0:
            return None
        out = [None] * n_shards
        if self.device.type == "cuda":          # one page-locked landing buffer for all ranks' words, one synchronisation
            need = sum(int(b.numel()) for b in bufs)
            recv = getattr(self, "_recv_pin", None)
            if recv is None or recv.numel() < need:
                recv = self._recv_pin = torch.empty(max(need, 1 << 20), dtype=torch.int32, pin_memory=True)
            views, o = [], 0
            for b in bufs:
                recv[o:o + b.numel()].copy_(b, non_blocking=True)
                views.append(recv[o:o + b.numel()])
                o += b.numel()
            torch.cuda.current_stream().synchronize()
            bufs = views
        for b in bufs:
            b = b.numpy().view(np.uint32)
            off = n_shards
            for i in b[:n_shards]:
                if i == 0xFFFFFFFF:
                    break
                out[int(i)] = b[off:off + int(lens[i])].copy() if copy else b[off:off + int(lens[i])]
                off += int(lens[i])
        mark("unpacked")
        if trace:
            print("gather_proofs: " + ", ".join(f"{b[0]} {1e3 * (b[1] - a[1]):.2f} ms" for a, b in zip(marks, marks[1:])), file=sys.stderr, flush=True)
        return out

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
