"""SysLinux (crates/core/machine/src/syscall/precompiles/sys_linux/): the Linux syscalls of a MIPS guest's runtime — brk, mmap / mmap2, clone, exit_group,
fcntl, read, write, and no-ops for the rest — one per row, with the result returned through the SyscallResult lookups. Pinned by the
reference's cost (187: width 103, 36 lookups) and by the executor's handlers (syscalls/precompiles/sys_linux/*.rs) restated in
events.linux_syscall."""
import json
import os

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine

CASES = [(E.SYS_BRK, 0x30000, 0), (E.SYS_BRK, 0x100, 0), (E.SYS_BRK, 0x20000, 0), (E.SYS_MMAP, 0, 0x2345), (E.SYS_MMAP2, 0, 0x4000), (E.SYS_MMAP, 0, 0xfff001),
         (E.SYS_MMAP, 0, 0x12fff123), (E.SYS_MMAP, 0x5000, 0x100), (E.SYS_CLONE, 5, 6), (E.SYS_EXT_GROUP, 0, 0), (E.SYS_FCNTL, 0, 3), (E.SYS_FCNTL, 2, 3), (E.SYS_FCNTL, 7, 3),
         (E.SYS_FCNTL, 1, 1), (E.SYS_FCNTL, 9, 1), (E.SYS_FCNTL, 1, 2), (E.SYS_READ, 0, 5), (E.SYS_READ, 3, 5), (E.SYS_WRITE_LINUX, 1, 0x1000), (E.SYS_OPEN, 1, 2),
         (4222, 0xffffffff, 0xfffffffe)]


def levent(code, a0, a1, brk=0x20000, heap=0x30000000, a2=77, shard=2, clk=300, seed=0):
    """The flattened LinuxEvent of a Linux syscall with the registers BRK, HEAP and $a2 holding brk, heap and a2."""
    rng = np.random.default_rng(seed)
    prev = lambda: (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))      # noqa: E731
    e = np.zeros(1, dtype=E.LINUX_EVENT)[0]
    v0, a3, new_heap = E.linux_syscall(code, a0, a1, brk=brk, heap=heap, a2=a2)
    e["shard"], e["clk"], e["a0"], e["a1"], e["v0"], e["syscall_code"] = shard, clk, a0, a1, v0, code
    if code == E.SYS_BRK:
        e["read_record"] = (brk, shard, clk) + prev()
    if code == E.SYS_WRITE_LINUX:
        e["read_record"] = (a2, shard, clk) + prev()
    e["a3_record"] = (a3, shard, clk, int(rng.integers(0, 1 << 32))) + prev()
    if new_heap is not None:
        e["heap_record"] = (new_heap, shard, clk, heap) + prev()
    return e


def some_events():
    return np.array([levent(*c, clk=300 + 10 * i, seed=i) for i, c in enumerate(CASES)])


def test_the_handlers():
    f = E.linux_syscall
    assert f(E.SYS_BRK, 5, 0, brk=9) == (9, 0, None) and f(E.SYS_BRK, 12, 0, brk=9) == (12, 0, None)
    assert f(E.SYS_MMAP, 0, 0x1001, heap=0x100) == (0x100, 0, 0x2100) and f(E.SYS_MMAP2, 0, 0x3000, heap=0x100) == (0x100, 0, 0x3100) and f(E.SYS_MMAP, 0x77, 1) == (0x77, 0, None)
    assert f(E.SYS_FCNTL, 2, 1) == (2, 0, None) and f(E.SYS_FCNTL, 2, 3) == (1, 0, None) and f(E.SYS_FCNTL, 0, 3) == (0, 0, None) and f(E.SYS_FCNTL, 5, 3) == (0xffffffff, 9, None)
    assert f(E.SYS_READ, 0, 0) == (0, 0, None) and f(E.SYS_READ, 1, 0) == (0xffffffff, 9, None) and f(E.SYS_WRITE_LINUX, 1, 2, a2=40) == (40, 0, None)
    assert f(E.SYS_CLONE, 1, 1) == (1, 0, None) and f(4222, 1, 1) == (0, 0, None)


def test_sys_linux_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    evs = some_events()
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_sys_linux(evs, -1, counts)
    assert t.shape == (32, E.SYS_LINUX_WIDTH)
    tc = F.from_monty(t)
    assert air.debug_constraints(chips.record_sys_linux_constraints().b, tc) == []
    # the $a3 write of every call; brk: the read, two comparisons (one when a0 = brk), four range checks; mmap: four, and with a0 = 0 the heap write,
    # the size, the addition's three words; write: the read
    want = 2 * len(evs) + (2 + 2 + 4) + (2 + 2 + 4) + (2 + 1 + 4) + 4 * (4 + 2 + 2 + 6) + 4 + 2
    assert counts.sum() == want
    for i, (code, a0, a1) in enumerate(CASES):
        assert sum(int(tc[i, 11 + k]) << (8 * k) for k in range(4)) == int(evs[i]["v0"])
    chip = chips.record_sys_linux_chip(5)
    assert not chip.local_only and chip.main_width + 4 * chip.perm_ext_width + 8 == ref[chip.name] == 187
    for field, value in (("v0", 3), ("a0", 0x31000)):
        forged = evs.copy()
        forged[field][0] = value
        with pytest.raises(RuntimeError, match="does not return"):
            oracle.tracegen_sys_linux(forged)
    forged = evs.copy()
    forged["heap_record"]["value"][3] += 0x1000
    with pytest.raises(RuntimeError, match="rounded size"):
        oracle.tracegen_sys_linux(forged)


def test_every_sys_linux_column_is_bound(oracle):
    evs = some_events()
    t = F.from_monty(oracle.tracegen_sys_linux(evs))
    rec, chip = chips.record_sys_linux_constraints(), chips.record_sys_linux_chip(5)
    inverses = {c for c in (41, 43, 45, 47, 49, 51, 53, 55, 58, 60, 62, 64, 66, 79)}      # IsZeroOperation inverses: free when the operand is zero (is_zero.rs:33-49)
    # what a branch does not use is unconstrained on its rows, as in the reference: sweep a row of every branch and require each column to be
    # bound on at least one of them (the mmap-only, brk-only and inorout columns are bound only where their branch runs)
    holes = windowed_sweep(rec, chip, t, (1, 3, 6, 7, 8, 11, 14, 17, 18, 19))
    assert [h for h in holes if h not in inverses] == [], holes


def linux_machine():
    calls = ((E.SYS_BRK, 0x40000, 0), (E.SYS_MMAP, 0, 0x2345), (E.SYS_MMAP2, 0, 0x1000), (E.SYS_MMAP, 0x9000, 0x10), (E.SYS_FCNTL, 1, 3), (E.SYS_FCNTL, 8, 1), (E.SYS_READ, 0, 4),
             (E.SYS_READ, 2, 4), (E.SYS_WRITE_LINUX, 1, 0x5000, 33), (E.SYS_CLONE, 0, 0), (E.SYS_OPEN, 3, 3), (E.SYS_BRK, 0x10, 0))
    return M.run_machine(900, seed=14, shard_cycles=2048, linux_calls=calls)


def test_machine_with_linux_calls_is_coherent(oracle):
    """Linux syscalls inside a run: the CPU writes their results to $v0, the SyscallInstrs chip sends the call and its result, the SyscallCore /
    SyscallPrecompile tables and the Global chip carry both to the precompile shard, where SysLinux receives them."""
    m = linux_machine()
    pre = [s.record.linux for s in m.shards if s.kind == "precompile"]
    assert len(pre) == 1 and len(pre[0]) == 12
    assert [int(x) for x in pre[0]["v0"]] == [0x40000, 0, 0x3000, 0x9000, 1, 0xffffffff, 0, 0xffffffff, 33, 1, 0, 0x10]      # the heap starts at 0; brk only reads register BRK (0)
    assert [int(x) for x in pre[0]["a3_record"]["value"]] == [0, 0, 0, 0, 0, 9, 0, 9, 0, 0, 0, 0]
    shards = check_machine_airs(oracle, m)
    assert "SysLinux" in {c.name for cs in shards for c in cs}
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_sys_linux_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_sys_linux against the restated generate_trace, bit for bit, with the byte lookups: a call of every branch, one, none, 400
    random ones in a fixed table; a wrong result and a wrong heap are errors."""
    from ziren_amd import lib
    evs = some_events()
    rng = np.random.default_rng(9)
    codes = [E.SYS_BRK, E.SYS_MMAP, E.SYS_MMAP2, E.SYS_CLONE, E.SYS_FCNTL, E.SYS_READ, E.SYS_WRITE_LINUX, E.SYS_OPEN, 4338]
    many = np.array([levent(codes[i % len(codes)], int(rng.integers(0, 4)) if i % 3 else int(rng.integers(0, 1 << 32)), int(rng.integers(0, 5)) if i % 2 else int(rng.integers(0, 1 << 32)),
                            brk=int(rng.integers(0, 1 << 32)), heap=int(rng.integers(0, 1 << 31)), a2=int(rng.integers(0, 1 << 32)), clk=100 + 7 * i, seed=i) for i in range(400)])
    for ev, fixed in ((evs, -1), (evs[:1], -1), (evs[:0], -1), (many, 9)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_sys_linux(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_sys_linux(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    forged = evs.copy()
    forged["v0"][0] = 3
    with pytest.raises(lib.ZkmError, match="does not return"):
        hip_ctx.tracegen_sys_linux(forged)
    forged = evs.copy()
    forged["heap_record"]["value"][3] += 0x1000
    with pytest.raises(lib.ZkmError, match="does not return"):
        hip_ctx.tracegen_sys_linux(forged)


@pytest.mark.gpu
def test_gpu_machine_with_linux_calls_proves_and_verifies(hip_ctx, oracle):
    m = linux_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
