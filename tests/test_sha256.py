"""The SHA-256 precompiles (crates/core/machine/src/syscall/precompiles/sha256/): ShaExtend (48 rows per call) and ShaCompress (80 rows per
call), fully in the reference tree — columns, AIR and trace generation are transcribed with file:line, pinned by the reference's cost table
(15 936 and 40 480 per call: column counts, lookup counts and constraint degree all enter) and by SHA-256 known answers (hashlib) for the
values the rows carry. Restated in the oracle (generate_sha_extend / generate_sha_compress), recorded as AIRs (chips._sha_extend /
_sha_compress), built on the device (zkm_tracegen_sha_extend / _sha_compress) and proven inside a machine whose program calls SHA_EXTEND
and SHA_COMPRESS the way the reference's own test programs do (extend/mod.rs:44-61, compress/mod.rs:52-78)."""
import hashlib
import json
import os

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_air_completeness import windowed_sweep
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine


def sha_events(w16, h, shard=2, clk=500, w_ptr=0x400000, h_ptr=0x500000, seed=0):
    """One SHA-256 block as the two syscalls record it (syscalls/precompiles/sha256/extend.rs:16-74, compress.rs:34-118): the ShaExtendEvent
    (w[16..64] one word per cycle from clk), then the ShaCompressEvent at a later clk on the same schedule; previous accesses of the words
    in this and in earlier shards. Returns (extend event, compress event, the state written back)."""
    rng = np.random.default_rng(seed)
    last = {}
    mem = {w_ptr + 4 * i: w16[i] for i in range(16)}
    mem.update({h_ptr + 4 * i: h[i] for i in range(8)})

    def prev(addr):
        if addr in last:
            return last[addr]
        return (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))

    def rd(addr, ts):
        p = prev(addr)
        last[addr] = (shard, ts)
        return (mem.get(addr, 0), shard, ts, p[0], p[1])

    def wr(addr, ts, v):
        p = prev(addr)
        last[addr] = (shard, ts)
        out = (v, shard, ts, mem.get(addr, 0), p[0], p[1])
        mem[addr] = v
        return out

    ext = np.zeros(1, dtype=E.SHA_EXTEND_EVENT)[0]
    ext["shard"], ext["clk"], ext["w_ptr"] = shard, clk, w_ptr
    w = E.sha_extend(w16)
    for j in range(48):
        i, ts = 16 + j, clk + j
        ext["w_i_minus_15_reads"][j] = rd(w_ptr + 4 * (i - 15), ts)
        ext["w_i_minus_2_reads"][j] = rd(w_ptr + 4 * (i - 2), ts)
        ext["w_i_minus_16_reads"][j] = rd(w_ptr + 4 * (i - 16), ts)
        ext["w_i_minus_7_reads"][j] = rd(w_ptr + 4 * (i - 7), ts)
        ext["w_i_writes"][j] = wr(w_ptr + 4 * i, ts, w[i])
    clk2 = clk + 48 + 5
    cmp_ = np.zeros(1, dtype=E.SHA_COMPRESS_EVENT)[0]
    cmp_["shard"], cmp_["clk"], cmp_["w_ptr"], cmp_["h_ptr"] = shard, clk2, w_ptr, h_ptr
    for i in range(8):
        cmp_["h_read_records"][i] = rd(h_ptr + 4 * i, clk2)
    for i in range(64):
        cmp_["w_i_read_records"][i] = rd(w_ptr + 4 * i, clk2)
    out = E.sha_compress(h, w)
    for i in range(8):
        cmp_["h_write_records"][i] = wr(h_ptr + 4 * i, clk2 + 1, out[i])
    return ext, cmp_, out


def one_block_message(msg):
    m = bytearray(msg) + b"\x80"
    m += bytes((56 - len(m)) % 64) + (8 * len(msg)).to_bytes(8, "big")
    assert len(m) == 64
    return [int.from_bytes(m[4 * i:4 * i + 4], "big") for i in range(16)]


def two_blocks():
    """SHA-256("abc") from the initial state, and a block of random words on a random state."""
    e1, c1, out = sha_events(one_block_message(b"abc"), E.SHA256_IV)
    rng = np.random.default_rng(1)
    e2, c2, _ = sha_events([int(x) for x in rng.integers(0, 1 << 32, 16)], [int(x) for x in rng.integers(0, 1 << 32, 8)], clk=900, w_ptr=0x410000,
                           h_ptr=0x510000, seed=3)
    return np.array([e1, e2]), np.array([c1, c2]), out


def test_sha256_events_hash_like_hashlib():
    for msg in (b"", b"abc", b"The quick brown fox jumps over the lazy dog"):
        _, _, out = sha_events(one_block_message(msg), E.SHA256_IV)
        assert b"".join(x.to_bytes(4, "big") for x in out).hex() == hashlib.sha256(msg).hexdigest()


def test_sha_rows_satisfy_the_airs_and_cost_what_the_reference_says(oracle):
    ext, cmp_, out = two_blocks()
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    for name, ev, gen, rec_c, rec_chip, rows, n_lookups in (
            ("ShaExtend", ext, oracle.tracegen_sha_extend, chips.record_sha_extend_constraints, chips.record_sha_extend_chip, 48, 71),
            ("ShaCompress", cmp_, oracle.tracegen_sha_compress, chips.record_sha_compress_constraints, chips.record_sha_compress_chip, 80, 115)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        t = gen(ev, -1, counts)             # the oracle refuses events whose write records are not what their reads give
        assert t.shape == (128 if rows == 48 else 256, rec_c().b.main_width)
        assert air.debug_constraints(rec_c().b, F.from_monty(t)) == [], name
        chip = rec_chip(8)
        assert len(chip.sends) + len(chip.receives) == n_lookups
        # MipsAir::get_chips_and_costs scales a precompile's row cost by its rows per call (mips/mod.rs:220,224)
        assert rows * (chip.main_width + 4 * chip.perm_ext_width + 8) == ref[name]
    # ShaExtend: 5 accesses x 2, 6 shift / rotate x 4, 4 xor x 4, 10 range checks; ShaCompress round: 2 + 6 x 4 + 7 x 4 + 5 x 4 + 2 + 12 + 3 x 6
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    oracle.tracegen_sha_extend(ext, -1, counts)
    assert counts.sum() == 2 * 48 * (10 + 24 + 16 + 10)
    counts[:] = 0
    oracle.tracegen_sha_compress(cmp_, -1, counts)
    assert counts.sum() == 2 * (8 * 2 + 64 * (2 + 24 + 28 + 20 + 2 + 12 + 18) + 8 * (2 + 6))
    # the state the second chip's last rows write back is SHA-256("abc")
    tc = F.from_monty(oracle.tracegen_sha_compress(cmp_))
    got = [int(sum(int(tc[72 + i, 23 + 4 + k]) << (8 * k) for k in range(4))) for i in range(8)]
    assert got == out and b"".join(x.to_bytes(4, "big") for x in out).hex() == hashlib.sha256(b"abc").hexdigest()
    forged = ext.copy()
    forged["w_i_writes"][1, 20]["value"] ^= 4
    with pytest.raises(RuntimeError, match="schedule word"):
        oracle.tracegen_sha_extend(forged)
    forged = cmp_.copy()
    forged["h_write_records"][0, 3]["value"] ^= 1
    with pytest.raises(RuntimeError, match="state"):
        oracle.tracegen_sha_compress(forged)


# column -> why nothing in the reference's eval reads it
SHA_EXTEND_FREE = {}
SHA_COMPRESS_FREE = {}


def test_every_sha_column_is_bound(oracle):
    """The completeness sweep (test_air_completeness.windowed_sweep) for the two chips: a first row of a call, rows of each 16-row lap /
    each phase, the last row of a call, the row before the padding."""
    ext, cmp_, _ = two_blocks()
    t = F.from_monty(oracle.tracegen_sha_extend(ext))
    holes = windowed_sweep(chips.record_sha_extend_constraints(), chips.record_sha_extend_chip(7), t, (48, 49, 63, 64, 80, 95, 47, 20))
    assert [c for c in holes if c not in SHA_EXTEND_FREE] == [], holes
    t = F.from_monty(oracle.tracegen_sha_compress(cmp_))
    holes = windowed_sweep(chips.record_sha_compress_constraints(), chips.record_sha_compress_chip(8), t, (80, 83, 87, 88, 100, 151, 152, 155, 159, 79, 9))
    assert [c for c in holes if c not in SHA_COMPRESS_FREE] == [], holes


def sha_machine():
    return M.run_machine(2500, seed=9, shard_cycles=1024, sha_calls=2, keccak_calls=1)


def test_machine_with_sha_calls_is_coherent(oracle):
    """A run that compresses two blocks (SHA_EXTEND then SHA_COMPRESS each, 48 and 1 extra cycles) and hashes one message with the Keccak
    precompile: CPU shards, one precompile shard per syscall code, the memory shard. Every chip's constraints hold, every shard's lookups
    cancel, the global digests sum to zero, and the state each compress call left in memory is hashlib's SHA-256 compression of its block."""
    m = sha_machine()
    kinds = [s.kind for s in m.shards]
    assert kinds[-4:] == ["precompile", "precompile", "precompile", "memory"] and set(kinds[:-4]) == {"cpu"}
    by_chip = {}
    for s in m.shards:
        if s.kind == "precompile":
            for name in ("keccak_sponge", "sha_extend", "sha_compress"):
                if len(getattr(s.record, name)):
                    by_chip[name] = s.record
    assert len(by_chip["sha_extend"].sha_extend) == 2 and len(by_chip["sha_compress"].sha_compress) == 2
    for ext, cmp_ in zip(by_chip["sha_extend"].sha_extend, by_chip["sha_compress"].sha_compress):
        w16 = [int(ext["w_i_minus_16_reads"][j]["value"]) for j in range(16)]       # row j reads w[j]
        assert [int(x) for x in cmp_["w_i_read_records"]["value"]] == E.sha_extend(w16)
        assert [int(x) for x in cmp_["h_write_records"]["value"]] == E.sha_compress(E.SHA256_IV, E.sha_extend(w16))
        assert int(cmp_["w_ptr"]) == int(ext["w_ptr"]) and int(cmp_["clk"]) > int(ext["clk"]) + 48
    shards = check_machine_airs(oracle, m)
    names = [{c.name for c in cs} for cs in shards]
    assert {"SyscallPrecompile", "ShaExtend", "MemoryLocal", "Global", "Byte", "Program"} in names
    assert {"SyscallPrecompile", "ShaCompress", "MemoryLocal", "Global", "Byte", "Program"} in names
    d = global_digests(shards)
    assert len(d) == len(m.shards) and oracle.global_digest_sum(d + [ZERO_DIGEST])[1]


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_sha_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_sha_extend / _sha_compress against the restated generate_trace, bit for bit, with the byte lookups they record: the
    hand-made calls, a run's calls, one call, none, 150 calls in a fixed table; forged write records are errors."""
    from ziren_amd import lib
    ext, cmp_, _ = two_blocks()
    m = sha_machine()
    run_ext = [s.record.sha_extend for s in m.shards if s.kind == "precompile" and len(s.record.sha_extend)][0]
    run_cmp = [s.record.sha_compress for s in m.shards if s.kind == "precompile" and len(s.record.sha_compress)][0]
    many = [sha_events([int(x) for x in np.random.default_rng(i).integers(0, 1 << 32, 16)], E.SHA256_IV, clk=100 + 60 * i, seed=i) for i in range(150)]
    many_ext, many_cmp = np.array([x[0] for x in many]), np.array([x[1] for x in many])
    for gen_o, gen_d, cases in ((oracle.tracegen_sha_extend, hip_ctx.tracegen_sha_extend, ((ext, -1), (run_ext, -1), (ext[:1], -1), (ext[:0], -1), (many_ext, 13))),
                                (oracle.tracegen_sha_compress, hip_ctx.tracegen_sha_compress, ((cmp_, -1), (run_cmp, -1), (cmp_[:1], -1), (cmp_[:0], -1), (many_cmp, 14)))):
        for ev, fixed in cases:
            counts = np.zeros((1 << 16, 10), dtype=np.uint32)
            want = gen_o(ev, fixed, counts)
            blu = hip_ctx.byte_lookups()
            born = gen_d(ev, fixed, blu)
            mults = hip_ctx.tracegen_byte_mults(blu)
            assert (born.height, born.width) == want.shape
            got = born.to_host()
            assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
            assert np.array_equal(F.from_monty(mults.to_host()), counts)
            born.free(); mults.free(); blu.free()
    forged = ext.copy()
    forged["w_i_writes"][1, 20]["value"] ^= 4
    with pytest.raises(lib.ZkmError, match="schedule word"):
        hip_ctx.tracegen_sha_extend(forged)
    forged = cmp_.copy()
    forged["h_write_records"][0, 3]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="state"):
        hip_ctx.tracegen_sha_compress(forged)


@pytest.mark.gpu
def test_gpu_machine_with_sha_calls_proves_and_verifies(hip_ctx, oracle):
    """The run above through the HIP prover: every shard — CPU shards, the Keccak, ShaExtend and ShaCompress precompile shards, the memory
    shard — from device-born traces, each proof bit-identical to the oracle's, accepted by the restated machine verifier."""
    m = sha_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
