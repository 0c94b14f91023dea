"""The KeccakSponge precompile (BASELINE config 4's workload: examples/keccak-precompile): the chip
crates/core/machine/src/syscall/precompiles/keccak_sponge/ with the round columns of p3-keccak-air, restated in the oracle
(oracle/tracegen.hpp generate_keccak_sponge), recorded as an AIR (ziren_amd/chips.py _keccak_sponge / _keccak_air), built on the device
(zkm_tracegen_keccak_sponge) and proven inside a machine whose program calls KECCAK_SPONGE the way the guest library's keccak256 does.

What pins it: the reference's cost table (24 rows x 4259 columns-equivalents per call block, which fixes the 2633 columns of KeccakCols
and the 357 lookups), Keccak-256 known answers for the values the rows carry, and the machine itself — the chip's syscall, memory and
byte lookups cancel against the reference-transcribed SyscallPrecompile, MemoryLocal and Byte chips. p3-keccak-air is not under
/root/reference; its constraint list is restated from the published crate and is otherwise unpinned."""
import json
import os

import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_machine import ZERO_DIGEST, check_machine_airs, global_digests, gpu_prove_machine

# Keccak-256 of b"abc" and of the empty string's neighbour cases, from the Keccak team's test vectors (also what the guest library's
# keccak256 returns: crates/zkvm/lib/src/keccak256.rs)
KECCAK256_ABC = "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
KECCAK256_QUICK_BROWN_FOX = "4d741b6f1eb29cb2a9b9911c82f56fa8d73b04959d3d9d222895df6c0b28aa15"   # "The quick brown fox jumps over the lazy dog"


def sponge_blocks(words, shard=3, clk=1000, input_addr=0x200000, output_addr=0x300000, seed=0):
    """The blocks of one KECCAK_SPONGE call on `words` (a multiple of 36) as KeccakSpongeSyscall::execute records them, with previous
    accesses in this and in earlier shards. Returns (blocks, final state lanes)."""
    rng = np.random.default_rng(seed)
    nb = len(words) // 36
    blocks = np.zeros(nb, dtype=E.KECCAK_SPONGE_BLOCK)
    st = [0] * 25
    for i in range(nb):
        b = blocks[i]
        b["shard"], b["clk"], b["input_addr"], b["output_addr"], b["input_len_u32s"], b["block_index"] = shard, clk, input_addr, output_addr, len(words), i
        blk = words[36 * i:36 * i + 36]
        for k in range(18):
            st[k] ^= blk[2 * k] | (blk[2 * k + 1] << 32)
        for k in range(25):
            b["xored_state"][2 * k], b["xored_state"][2 * k + 1] = st[k] & 0xffffffff, st[k] >> 32
        for j in range(36):
            r = b["input_read_records"][j]
            r["value"], r["shard"], r["timestamp"] = blk[j], shard, clk
            r["prev_shard"], r["prev_timestamp"] = (shard, int(rng.integers(0, clk))) if rng.random() < 0.7 else (int(rng.integers(0, shard)), int(rng.integers(0, 1 << 20)))
        st = E.keccak_f(st)
        if i == 0:
            r = b["input_length_record"]
            r["value"], r["shard"], r["timestamp"], r["prev_shard"], r["prev_timestamp"] = len(words), shard, clk, shard, clk - 7
        if i == nb - 1:
            for j in range(16):
                r = b["output_write_records"][j]
                r["value"], r["shard"], r["timestamp"] = (st[j // 2] >> (32 * (j & 1))) & 0xffffffff, shard, clk + 1
                r["prev_value"], r["prev_shard"], r["prev_timestamp"] = int(rng.integers(0, 1 << 32)), shard, clk - 3
    return blocks, st


def digest_hex(st):
    return b"".join(int(x).to_bytes(8, "little") for x in st[:4]).hex()


def two_calls():
    """A three-block call, then a one-block call: rows 0-71 and 72-95 of a 128-row trace."""
    b1, _ = sponge_blocks(E.keccak256_words(bytes(range(200)) * 2), clk=2000, input_addr=0x210000, output_addr=0x310000, seed=1)
    b2, st = sponge_blocks(E.keccak256_words(b"abc"))
    assert len(b1) == 3 and len(b2) == 1
    return np.concatenate([b1, b2]), st


def test_keccak_f_known_answers():
    for msg, want in ((b"abc", KECCAK256_ABC), (b"The quick brown fox jumps over the lazy dog", KECCAK256_QUICK_BROWN_FOX)):
        _, st = sponge_blocks(E.keccak256_words(msg))
        assert digest_hex(st) == want
    # 135 bytes: both padding bits share the last byte of the block; 136 bytes: a whole extra block of padding
    assert len(E.keccak256_words(b"x" * 135)) == 36 and E.keccak256_words(b"x" * 135)[33] >> 24 == 0x81
    assert len(E.keccak256_words(b"x" * 136)) == 72


def test_keccak_sponge_rows_satisfy_the_air_and_cost_what_the_reference_says(oracle):
    blocks, st = two_calls()
    assert digest_hex(st) == KECCAK256_ABC
    counts = np.zeros((1 << 16, 10), dtype=np.uint32)
    t = oracle.tracegen_keccak_sponge(blocks, -1, counts)      # the oracle refuses rows whose output records are not the squeezed state
    assert t.shape == (128, E.KECCAK_SPONGE_WIDTH)
    # per block: 36 reads x 2 range checks + 36 x 4 XOR lookups; per call: 2 for the length word + 16 x 2 for the output
    assert counts.sum() == 4 * (72 + 144) + 2 * (2 + 32)
    rec = chips.record_keccak_sponge_constraints()
    tc = F.from_monty(t)
    assert air.debug_constraints(rec.b, tc) == []
    # the output words of the second call are Keccak-256("abc"), as the chip's last real row holds them (output_mem value bytes)
    out = tc[95, 3323:3531].reshape(16, 13)[:, 4:8]
    assert bytes(int(x) for x in out[:8].reshape(-1)).hex() == KECCAK256_ABC
    # rows past the last block: rounds of the permutation of the zero state, the flags rotating on from the real rows
    assert [int(np.argmax(tc[i, :24])) for i in (96, 97, 127)] == [0, 1, 127 % 24] and not tc[96:, 2633:].any()
    chip = chips.record_keccak_sponge_chip(7)
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    perm_width = air.local_permutation_trace_width(len(chip.sends) + len(chip.receives), 2)
    assert (len(chip.sends), len(chip.receives), perm_width, chip.log_quotient_degree) == (303, 54, 180, 1)
    # MipsAir::costs multiplies a precompile's row cost by its rows per event (mips/mod.rs:588-595: KeccakSponge => 24)
    assert 24 * (chip.main_width + 4 * perm_width + 4 * 2) == ref["KeccakSponge"] == 102216
    # a call cut short, blocks out of order, a forged output: errors, not rows
    with pytest.raises(RuntimeError, match="cut short|chain"):
        oracle.tracegen_keccak_sponge(blocks[:2])
    with pytest.raises(RuntimeError, match="chain"):
        oracle.tracegen_keccak_sponge(blocks[[0, 2, 1, 3]])
    forged = blocks.copy()
    forged["output_write_records"][3, 5]["value"] ^= 1
    with pytest.raises(RuntimeError, match="squeezed"):
        oracle.tracegen_keccak_sponge(forged)


# (column) -> why nothing reads it
KECCAK_FREE = {}


def test_every_keccak_sponge_column_is_bound(oracle):
    """The completeness sweep of tests/test_air_completeness.py for this chip: every one of the 3531 columns, changed on one of six rows
    (first row of a call, of a later block, a middle round, the last round of a block that is absorbed on, of a call's last block, of the
    whole table), must be noticed by a constraint or by the chip's lookups. The constraints only look at a row and its successor, so all
    3531 variants of a row are evaluated in one pass over a stack of three-row windows (test_air_completeness.windowed_sweep)."""
    from test_air_completeness import windowed_sweep
    blocks, _ = two_calls()
    t = F.from_monty(oracle.tracegen_keccak_sponge(blocks))
    holes = windowed_sweep(chips.record_keccak_sponge_constraints(), chips.record_keccak_sponge_chip(7), t, (72, 24, 31, 23, 71, 95))
    assert [c for c in holes if c not in KECCAK_FREE] == [], holes


def keccak_machine():
    return M.run_machine(3000, seed=5, shard_cycles=1024, poseidon2_calls=1, keccak_calls=2)


def test_machine_with_keccak_calls_is_coherent(oracle):
    """A run that hashes two messages with the precompile: the CPU shards (the call costs one extra cycle, which the Cpu chip's clk
    transition and the SyscallInstrs chip both check), one precompile shard per syscall code (ExecutionRecord::split), the memory shard.
    Every chip's constraints hold, every shard's lookups cancel, and the seven global digests sum to zero."""
    m = keccak_machine()
    kinds = [s.kind for s in m.shards]
    assert kinds == ["cpu"] * (len(kinds) - 3) + ["precompile", "precompile", "memory"]
    kec = m.shards[-2].record
    assert len(kec.precompile_syscall) == 2 and len(kec.poseidon2_permute) == 0 and 2 <= len(kec.keccak_sponge) <= 6
    assert int(kec.precompile_syscall["syscall_id"][0]) == E.SYS_KECCAK_SPONGE & 0xffff
    # the digest the run left in memory is Keccak-256 of the message it stored: recomputed from the blocks' own input words
    for call in range(2):
        first = [i for i, b in enumerate(kec.keccak_sponge) if b["block_index"] == 0][call]
        nb = int(kec.keccak_sponge[first]["input_len_u32s"]) // 36
        st = [0] * 25
        for b in kec.keccak_sponge[first:first + nb]:
            for i in range(18):
                st[i] ^= int(b["input_read_records"][2 * i]["value"]) | (int(b["input_read_records"][2 * i + 1]["value"]) << 32)
            st = E.keccak_f(st)
        out = kec.keccak_sponge[first + nb - 1]["output_write_records"]["value"]
        assert [int(x) for x in out] == [(st[j // 2] >> (32 * (j & 1))) & 0xffffffff for j in range(16)]
    shards = check_machine_airs(oracle, m)
    assert {c.name for c in shards[-2]} == {"SyscallPrecompile", "KeccakSponge", "MemoryLocal", "Global", "Byte", "Program"}
    d = global_digests(shards)
    assert len(d) == len(m.shards) and oracle.global_digest_sum(d + [ZERO_DIGEST])[1]
    assert not oracle.global_digest_sum(d[:-2] + d[-1:] + [ZERO_DIGEST])[1]       # without the keccak shard the messages do not cancel


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_keccak_sponge_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_keccak_sponge against the restated generate_trace, bit for bit, with the byte lookups it records: the two hand-made
    calls, the calls of a machine run, one block, no block, and 170 calls in a fixed 2^13-row table; broken inputs are errors."""
    from ziren_amd import lib
    blocks, _ = two_calls()
    run = keccak_machine().shards[-2].record.keccak_sponge
    many = np.concatenate([sponge_blocks(E.keccak256_words(bytes([i]) * (i + 1)), clk=100 + 10 * i, seed=i)[0] for i in range(170)])
    for ev, fixed in ((blocks, -1), (run, -1), (blocks[3:], -1), (blocks[:0], -1), (many, 13)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_keccak_sponge(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_keccak_sponge(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape
        got = born.to_host()
        assert np.array_equal(got, want), (len(ev), np.argwhere(got != want)[:5])
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
    for broken, why in ((blocks[:2], "chain"), (blocks[[0, 2, 1, 3]], "chain")):
        with pytest.raises(lib.ZkmError, match=why):
            hip_ctx.tracegen_keccak_sponge(broken)
    forged = blocks.copy()
    forged["output_write_records"][3, 5]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="squeezed"):
        hip_ctx.tracegen_keccak_sponge(forged)
    forged = blocks.copy()
    forged["input_len_u32s"][3] = 35
    with pytest.raises(lib.ZkmError, match="multiple of 36"):
        hip_ctx.tracegen_keccak_sponge(forged)


@pytest.mark.gpu
def test_gpu_machine_with_keccak_calls_proves_and_verifies(hip_ctx, oracle):
    """BASELINE config 4's workload at test size (examples/keccak-precompile: a guest that hashes with the KECCAK_SPONGE precompile,
    several shards): every shard — CPU shards, the Poseidon2 and Keccak precompile shards, the memory shard — proven on the GPU from
    device-born traces, bit-identical to the oracle's proofs, and accepted by the restated machine verifier."""
    m = keccak_machine()
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
    assert ML.verify_machine(oracle, opk, oshards[:-2] + oshards[-1:], proofs[:-2] + proofs[-1:], fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is not None
