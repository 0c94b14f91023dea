"""A whole machine over several shards: CPU shards of one program, the shard of its deferred
precompile events (POSEIDON2_PERMUTE: SyscallInstrs -> SyscallCore -> global table -> SyscallPrecompile -> Poseidon2Permute), and the
memory shard (MemoryGlobalInit / MemoryGlobalFinalize). Every lookup of every shard is exchanged between real chips, every cross-shard
message is sent once and received once (the global digests sum to the zero digest with no stand-in on any side), and the proofs pass
the restated ZKMProver::verify + StarkMachine::verify (tests/machine_lib.py). The reference's fibonacci guest (BASELINE config 1) runs
as a hand-assembled MIPS program."""
import numpy as np
import pytest

from ziren_amd import abi, air, chips, events as E, field as F, miniexec as M, synth

import machine_lib as ML
from test_chip_airs import lookup_tally

PC_BASE = 0x1000
ZERO_DIGEST = F.to_monty(np.array(chips.SEPTIC_START_X + chips.SEPTIC_START_Y, dtype=np.uint64)).astype(np.uint32)


from ziren_amd.fibfast import fib, fibonacci_program  # noqa: E402,F401  (the guest lives with its closed-form shard generator)


def check_machine_airs(oracle, m):
    """Per shard: every constraint of every included chip vanishes on the oracle's rows (with the shard's public values), and the shard's
    lookups cancel between its chips. Returns the shards' chip lists."""
    all_shards = []
    byte_prep, prog_prep = oracle.tracegen_byte_table(), oracle.tracegen_program(0, m.shards[0].record.cpu, m.program, m.pc_base, ML.log2_rows(len(m.program)))
    for k, sh in enumerate(m.shards):
        cs = ML.build_shard(ML.Oracle(oracle), m, k)
        cs[-2].prep_trace, cs[-1].prep_trace = byte_prep, prog_prep
        pv = F.from_monty(ML.shard_public_values(sh)).astype(np.uint64)
        for c in cs[:-2]:
            rec = {"Cpu": chips.record_cpu_constraints, "SyscallInstrs": chips.record_syscall_instrs_constraints, "Global": chips.record_global_constraints,
                   "MemoryGlobalInit": lambda: chips.record_memory_global_constraints(False), "MemoryGlobalFinalize": lambda: chips.record_memory_global_constraints(True),
                   "SyscallCore": lambda: chips.record_syscall_table_constraints(False), "SyscallPrecompile": lambda: chips.record_syscall_table_constraints(True),
                   "Poseidon2Permute": chips.record_poseidon2_permute_constraints, "KeccakSponge": chips.record_keccak_sponge_constraints,
                   "ShaExtend": chips.record_sha_extend_constraints, "ShaCompress": chips.record_sha_compress_constraints,
                   "EdAddAssign": chips.record_ed_add_constraints, "EdDecompress": chips.record_ed_decompress_constraints,
                   "Uint256MulMod": chips.record_uint256_mul_constraints, "U256XU2048Mul": chips.record_u256x2048_mul_constraints,
                   "BooleanCircuitGarble": chips.record_boolean_circuit_garble_constraints, "SysLinux": chips.record_sys_linux_constraints}.get(c.name)
            if rec is None and ("FpOpAssign" in c.name or "Fp2" in c.name):
                field = "Bn254" if c.name.startswith("Bn254") else "Bls12381"
                kind = "fp" if "FpOp" in c.name else "fp2_mul" if "Fp2Mul" in c.name else "fp2_addsub"
                rec = lambda field=field, kind=kind: chips.record_fp_tower_constraints(field, kind)      # noqa: E731
            if rec is None and c.name.endswith("Decompress"):
                rec = lambda curve=c.name.replace("Decompress", ""): chips.record_weierstrass_decompress_constraints(curve)      # noqa: E731
            if rec is None and c.name.endswith(("AddAssign", "DoubleAssign")):
                curve, double = c.name.replace("DoubleAssign", "").replace("AddAssign", ""), c.name.endswith("DoubleAssign")
                rec = lambda curve=curve, double=double: chips.record_weierstrass_constraints(curve, double)      # noqa: E731
            if rec is not None:
                assert air.debug_constraints(rec().b, F.from_monty(c.trace), public_values=pv) == [], (k, c.name)
        assert not any(lookup_tally(cs).values()), (k, sh.kind)
        all_shards.append(cs)
    return all_shards


def global_digests(shard_chips):
    return [c.trace[-1, 85:] for cs in shard_chips for c in cs if c.name == "Global"]


def test_machine_run_is_coherent(oracle):
    m = M.run_machine(2600, seed=4, shard_cycles=1024, poseidon2_calls=3)
    kinds = [s.kind for s in m.shards]
    assert kinds == ["cpu", "cpu", "cpu", "precompile", "memory"] and [s.pv["shard"] for s in m.shards] == [1, 2, 3, 4, 5]
    assert [s.pv["execution_shard"] for s in m.shards] == [1, 2, 3, 3, 3]
    for a, b in zip(m.shards, m.shards[1:]):
        assert b.pv["start_pc"] == a.pv["next_pc"]
    assert m.shards[-1].pv["next_pc"] == 0 and all(int(s.record.cpu["clk"][0]) == 0 for s in m.shards[:3])
    pre = m.shards[3].record
    assert len(pre.poseidon2_permute) == 3 and len(pre.memory_local) == 48
    # an address the precompile touched has its CPU access chain closed before and reopened after (SyscallContext::postprocess)
    mem = m.shards[4].record
    assert list(mem.memory_init["addr"]) == sorted(mem.memory_init["addr"]) and mem.memory_init["addr"][0] == 0
    assert (mem.memory_init["shard"] == 1).all() and (mem.memory_init["timestamp"] == 1).all()
    shards = check_machine_airs(oracle, m)
    names = [{c.name for c in cs} for cs in shards]
    assert "SyscallCore" in names[0] | names[1] | names[2] and names[3] == {"SyscallPrecompile", "Poseidon2Permute", "MemoryLocal", "Global", "Byte", "Program"}
    assert names[4] == {"MemoryGlobalInit", "MemoryGlobalFinalize", "Global", "Byte", "Program"}
    # StarkMachine::verify's last check with no stand-in: every message is sent once and received once across the five shards
    d = global_digests(shards)
    assert len(d) == 5 and oracle.global_digest_sum(d + [ZERO_DIGEST])[1]
    for drop in range(5):
        assert not oracle.global_digest_sum(d[:drop] + d[drop + 1:] + [ZERO_DIGEST])[1]


def test_new_chips_catch_corrupted_cells(oracle):
    m = M.run_machine(900, seed=8, shard_cycles=512, poseidon2_calls=1)
    mem, pre = m.shards[-1], m.shards[-2]
    pv = F.from_monty(ML.shard_public_values(mem)).astype(np.uint64)
    for finalize, ev in ((False, mem.record.memory_init), (True, mem.record.memory_finalize)):
        rec = chips.record_memory_global_constraints(finalize)
        t = F.from_monty(oracle.tracegen_memory_global(ev, 0))
        assert air.debug_constraints(rec.b, t, public_values=pv) == []
        n = len(ev)
        for col, rows in ((2, {3}), (3 + 2, {3}), (35 + 1, {2, 3}), (73 + 4, None), (105, None), (106, {3, 2}), (110, {3})):
            bad = t.copy()
            bad[3, col] = (int(bad[3, col]) + 1) % F.P if col != 73 + 4 else 2      # a value bit flipped 0 -> 1 is only caught by the lookup argument
            hit = {r for _, r in air.debug_constraints(rec.b, bad, public_values=pv)}
            assert hit and (rows is None or hit <= rows | {2, 3}), (finalize, col, hit)
        bad_pv = pv.copy()
        bad_pv[(141 if finalize else 77) + 2] ^= 1                       # a different last address is claimed
        assert {r for _, r in air.debug_constraints(rec.b, t, public_values=bad_pv)} == {n - 1}
        # a second chunk starts from the previous shard's last address
        t2 = F.from_monty(oracle.tracegen_memory_global(ev[5:], int(np.sort(ev["addr"])[4])))
        pv2 = pv.copy()
        for i in range(32):
            pv2[(109 if finalize else 45) + i] = (int(np.sort(ev["addr"])[4]) >> i) & 1
        assert air.debug_constraints(rec.b, t2, public_values=pv2) == []
        assert air.debug_constraints(rec.b, t2, public_values=pv) != []           # ... and not from zero
    for precompile, ev in ((False, np.concatenate([s.record.syscall for s in m.shards[:-2]])), (True, pre.record.precompile_syscall)):
        rec = chips.record_syscall_table_constraints(precompile)
        t = F.from_monty(oracle.tracegen_syscall(ev, precompile))
        assert t[:, 10].sum() == 1 and air.debug_constraints(rec.b, t) == []
        for col in (9, 10, 7):
            bad = t.copy()
            bad[0, col] = (int(bad[0, col]) + 1) % F.P if col != 9 else 2      # is_linux 0 -> 1 alone is caught by the SyscallResult lookup
            assert {r for _, r in air.debug_constraints(rec.b, bad)} == {0}, col
    rec = chips.record_poseidon2_permute_constraints()
    t = F.from_monty(oracle.tracegen_poseidon2_permute(pre.record.poseidon2_permute))
    assert air.debug_constraints(rec.b, t) == []
    for col in (0, 130, 160, 200, 305, 313, 316, 320, 324, 524, 748, 972):
        bad = t.copy()
        bad[0, col] = (int(bad[0, col]) + 1) % F.P
        assert {r for _, r in air.debug_constraints(rec.b, bad)} == {0}, col
    costs = {"MemoryGlobalInit": chips.record_memory_global_chip(False, 10), "MemoryGlobalFinalize": chips.record_memory_global_chip(True, 10),
             "SyscallCore": chips.record_syscall_table_chip(False, 10), "SyscallPrecompile": chips.record_syscall_table_chip(True, 10),
             "Poseidon2Permute": chips.record_poseidon2_permute_chip(10)}
    import json, os
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mips_costs.json")))["costs"]
    for name, c in costs.items():
        assert c.main_width + 4 * c.perm_ext_width + 8 == ref[name], name


def test_fibonacci_program_runs(oracle):
    """BASELINE config 1's program: n = 1000, committed words = (n, fib(n-1 .. n) mod 7919) as Python computes them; four CPU shards."""
    n = 1000
    m = M.run_machine(program=fibonacci_program(n), shard_cycles=2048)
    a, b = fib(n)
    cpu = [s for s in m.shards if s.kind == "cpu"]
    assert len(cpu) == 3 and sum(len(s.record.cpu) for s in cpu) == 5 + 6 * n + 35
    assert m.shards[0].pv["committed_value_digest"] == [n, a, b, 0, 0, 0, 0, 0] and all(s.pv["committed_value_digest"][1] == a for s in m.shards)
    assert [s.kind for s in m.shards] == ["cpu", "cpu", "cpu", "memory"]
    shards = check_machine_airs(oracle, m)
    assert oracle.global_digest_sum(global_digests(shards) + [ZERO_DIGEST])[1]
    # the program table is fetched 1000 times at the loop's instructions
    assert int(F.from_monty(shards[0][-1].trace).max()) > 300


def prove_machine(oracle, m, shards, fri):
    byte_prep, prog_prep = shards[0][-2].prep_trace, shards[0][-1].prep_trace
    igcs = ZERO_DIGEST
    opk = oracle.Pk([byte_prep, prog_prep], [0, 0], F.to_monty(m.pc_base), igcs, fri.log_blowup)
    start = oracle.new_challenger()
    opk.observe_into(start)
    proofs = [oracle.prove_shard(opk, cs, [c.trace for c in cs], ML.shard_public_values(sh), fri, synth.NUM_PV_ELTS, start.copy())[0]
              for cs, sh in zip(shards, m.shards)]
    return opk, proofs


def test_oracle_proves_and_verifies_a_machine(oracle):
    """The restated ZKMProver::verify + StarkMachine::verify accept the oracle's proofs of all the shards of a run with a precompile call,
    and reject: a missing shard, shards out of order, a shard proven for another shard number, a wrong claimed init address."""
    m = M.run_machine(700, seed=6, shard_cycles=400, poseidon2_calls=1)
    shards = check_machine_airs(oracle, m)
    fri = abi.FriConfig(1, 84, 16)
    opk, proofs = prove_machine(oracle, m, shards, fri)
    ok = lambda sc, pf: ML.verify_machine(oracle, opk, sc, pf, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST)   # noqa: E731
    assert ok(shards, proofs) is None
    assert ok(shards[:-1], proofs[:-1]) == "global cumulative sum is not zero"
    assert ok(shards[:-2] + shards[-1:], proofs[:-2] + proofs[-1:]) is not None
    assert ok(shards[1:], proofs[1:]) is not None and ok([shards[1], shards[0]] + shards[2:], [proofs[1], proofs[0]] + proofs[2:]) is not None
    # a memory shard proven with another claimed last address fails inside its own proof (MemoryGlobalInit's boundary constraint)
    start = oracle.new_challenger()
    opk.observe_into(start)
    bad_pv = dict(m.shards[-1].pv, last_init_addr=m.shards[-1].pv["last_init_addr"] ^ 4)
    bad, _ = oracle.prove_shard(opk, shards[-1], [c.trace for c in shards[-1]], M.public_values(bad_pv), fri, synth.NUM_PV_ELTS, start.copy())
    assert ok(shards, proofs[:-1] + [bad]) == f"shard {len(proofs)}: invalid shard proof (code {oracle.verify_shard(opk, shards[-1], fri, synth.NUM_PV_ELTS, start.copy(), bad)})"


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_new_chip_tracegen_matches_oracle(hip_ctx, oracle):
    """zkm_tracegen_memory_global / _syscall / _poseidon2_permute against the restated generate_trace, bit for bit, with the byte lookups
    they record; unsorted memory events are sorted as the reference sorts them; bad inputs are errors, not silent rows."""
    from ziren_amd import lib
    m = M.run_machine(3000, seed=12, shard_cycles=1 << 20, poseidon2_calls=5)
    mem, pre, cpu = m.shards[-1].record, m.shards[-2].record, m.shards[0].record
    rng = np.random.default_rng(3)
    for ev in (mem.memory_init, mem.memory_finalize):
        for sub, prev, fixed in ((ev, 0, -1), (ev[7:], int(ev["addr"][6]), -1), (ev[:1], 0, -1), (ev[rng.permutation(len(ev))], 0, 9), (ev[:0], 0, -1)):
            if len(sub) == 0:
                born = hip_ctx.tracegen_memory_global(sub, prev, fixed)
                assert born.height == 16 and not born.to_host().any()
                born.free()
                continue
            want = oracle.tracegen_memory_global(sub, prev, fixed)
            born = hip_ctx.tracegen_memory_global(sub, prev, fixed)
            assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), (len(sub), prev)
            born.free()
        with pytest.raises(lib.ZkmError, match="strictly increasing"):
            hip_ctx.tracegen_memory_global(ev[3:], int(ev["addr"][5]))
        with pytest.raises(lib.ZkmError, match="strictly increasing"):
            hip_ctx.tracegen_memory_global(np.concatenate([ev[:4], ev[3:6]]), 0)
    for precompile, ev in ((False, cpu.syscall), (True, pre.precompile_syscall), (False, cpu.syscall[:0]), (True, np.tile(pre.precompile_syscall, 300))):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_syscall(ev, precompile, -1, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_syscall(ev, precompile, -1, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), (precompile, len(ev))
        assert np.array_equal(F.from_monty(mults.to_host()), counts)
        born.free(); mults.free(); blu.free()
        if len(ev):
            # the same events already in HBM (zkm_events_upload_async): SyscallCore's filter runs on the device — with the shape's height
            # the kept count never reaches the host, without one it costs a round trip — and the rows are the same
            for fixed in (-1, int(np.log2(want.shape[0])), int(np.log2(want.shape[0])) + 2):
                dev = hip_ctx.events_upload_async(np.ascontiguousarray(ev))
                blu = hip_ctx.byte_lookups()
                born = hip_ctx.tracegen_syscall(dev, precompile, fixed, blu)
                mults = hip_ctx.tracegen_byte_mults(blu)
                want_f = want if fixed < 0 else oracle.tracegen_syscall(ev, precompile, fixed)
                assert (born.height, born.width) == want_f.shape and np.array_equal(born.to_host(), want_f), (precompile, len(ev), fixed)
                assert np.array_equal(F.from_monty(mults.to_host()), counts)
                born.free(); mults.free(); blu.free(); dev.free()
    if len(cpu.syscall) > 4:
        dev = hip_ctx.events_upload_async(np.ascontiguousarray(cpu.syscall))
        with pytest.raises(lib.ZkmError, match="too small"):
            hip_ctx.tracegen_syscall(dev, False, 2)         # five kept events do not fit four rows: found by the device-side filter
        dev.free()
    assert int(F.from_monty(oracle.tracegen_syscall(cpu.syscall, False))[:, 10].sum()) == 5      # of 14 syscalls only the precompile calls go to the table
    for ev, fixed in ((pre.poseidon2_permute, -1), (pre.poseidon2_permute[:1], -1), (pre.poseidon2_permute[:0], -1), (np.tile(pre.poseidon2_permute, 70), 9)):
        counts = np.zeros((1 << 16, 10), dtype=np.uint32)
        want = oracle.tracegen_poseidon2_permute(ev, fixed, counts)
        blu = hip_ctx.byte_lookups()
        born = hip_ctx.tracegen_poseidon2_permute(ev, fixed, blu)
        mults = hip_ctx.tracegen_byte_mults(blu)
        assert (born.height, born.width) == want.shape and np.array_equal(born.to_host(), want), len(ev)
        assert np.array_equal(F.from_monty(mults.to_host()), counts) and counts.sum() == 32 * len(ev)
        born.free(); mults.free(); blu.free()
    bad = pre.poseidon2_permute.copy()
    bad["state_records"][2, 5]["value"] ^= 1
    with pytest.raises(lib.ZkmError, match="permutation of the pre-state"):
        hip_ctx.tracegen_poseidon2_permute(bad)


def gpu_prove_machine(hip_ctx, oracle, m, fri):
    """Every shard of the run: traces born on the device (equal to the oracle's rows), proven by the HIP prover and by the oracle."""
    from ziren_amd import prover
    oshards = check_machine_airs(oracle, m)
    byte_prep, prog_prep = oshards[0][-2].prep_trace, oshards[0][-1].prep_trace
    pc_start = F.to_monty(m.pc_base)
    opk = oracle.Pk([byte_prep, prog_prep], [0, 0], pc_start, ZERO_DIGEST, fri.log_blowup)
    pk = None
    proofs, oproofs = [], []
    for k, (ocs, sh) in enumerate(zip(oshards, m.shards)):
        dev = ML.Device(hip_ctx)
        dcs = ML.build_shard(dev, m, k)
        assert [c.name for c in dcs] == [c.name for c in ocs]
        for d, o in zip(dcs, ocs):
            assert np.array_equal(d.trace.to_host(), o.trace), (k, d.name)
        hp = prover.HipProver(ocs, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
        hp.specialize_quotient_kernels(ocs)
        if pk is None:
            pk = hp.setup([hip_ctx.tracegen_byte_table(), hip_ctx.tracegen_program(m.program, m.pc_base, ocs[-1].log_height)], [0, 0], pc_start, ZERO_DIGEST)
            assert np.array_equal(pk.commit, opk.commitment())
        ch = prover.new_challenger()
        pk.observe_into(ch)
        och = oracle.new_challenger()
        opk.observe_into(och)
        pvs = ML.shard_public_values(sh)
        proofs.append(hp.prove_shard(pk, pvs, [c.trace for c in dcs], ch).copy())
        oproofs.append(oracle.prove_shard(opk, ocs, [c.trace for c in ocs], pvs, fri, synth.NUM_PV_ELTS, och)[0])
        assert np.array_equal(proofs[-1], oproofs[-1]), f"shard {k + 1}: GPU proof differs from the oracle's"
        for c in dcs:
            c.trace.free()
        dev.blu.free()
    return opk, oshards, proofs


@pytest.mark.gpu
@pytest.mark.parametrize("prefetch", [False, True])
def test_gpu_one_call_trace_generation_of_every_shard_kind(hip_ctx, oracle, prefetch):
    """zkm_tracegen_shard over whole machines (every chip of a CPU shard: jumps, memory and misc instructions, syscalls with their Core table,
    Global with the syscalls' messages; the deferred shards' syscall tables and Global next to their own entry points): every trace equal
    to the oracle's rows, from host events and from events copied ahead (nothing read on the host: SyscallCore filtered on the device at
    the height the shard fixes)."""
    for m in (M.run_machine(5000, seed=21, shard_cycles=2048, poseidon2_calls=4), M.run_machine(program=fibonacci_program(150), shard_cycles=512)):
        for k in range(len(m.shards)):
            ocs = ML.build_shard(ML.Oracle(oracle), m, k)
            dev = ML.DeviceOneCall(hip_ctx, prefetch)
            dcs = ML.build_shard(dev, m, k)
            dev.flush(dcs)
            assert [c.name for c in dcs] == [c.name for c in ocs]
            for d, o in zip(dcs, ocs):
                assert np.array_equal(d.trace.to_host(), o.trace), (m.shards[k].kind, k, d.name, prefetch)
                d.trace.free()
            dev.blu.free()


@pytest.mark.gpu
def test_gpu_machine_with_precompile_proves_and_verifies(hip_ctx, oracle):
    """BASELINE config 4's shape at test size (a program with precompile calls, several shards): CPU shards, the precompile shard and the
    memory shard proven on the GPU from device-born traces, bit-identical to the oracle's proofs, accepted by the restated machine
    verifier; without the memory shard the global digests do not cancel."""
    m = M.run_machine(5000, seed=21, shard_cycles=2048, poseidon2_calls=4)
    assert [s.kind for s in m.shards] == ["cpu", "cpu", "cpu", "precompile", "memory"]
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    ok = lambda sc, pf: ML.verify_machine(oracle, opk, sc, pf, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST)   # noqa: E731
    assert ok(oshards, proofs) is None
    assert ok(oshards[:-1], proofs[:-1]) == "global cumulative sum is not zero"
    assert ok(oshards[:3] + oshards[4:], proofs[:3] + proofs[4:]) is not None


@pytest.mark.gpu
def test_gpu_fibonacci_n1000(hip_ctx, oracle):
    """BASELINE config 1 (examples/fibonacci, n = 1000) through the HIP prover: the hand-assembled guest, three CPU shards and the memory
    shard, every proof bit-identical to the oracle's, the machine verifies, and the committed digest is (n, fib(n - 1), fib(n)) mod 7919
    as Python computes it."""
    n = 1000
    m = M.run_machine(program=fibonacci_program(n), shard_cycles=2048)
    fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, fri)
    assert ML.verify_machine(oracle, opk, oshards, proofs, fri, synth.NUM_PV_ELTS, m.pc_base, ZERO_DIGEST) is None
    a, b = fib(n)
    pv = F.from_monty(ML.decode_shard_proof(proofs[-1])["public_values"])
    words = [int(sum(int(pv[4 * i + k]) << (8 * k) for k in range(4))) for i in range(8)]
    assert words == [n, a, b, 0, 0, 0, 0, 0]


@pytest.mark.gpu
def test_gpu_core_proofs_feed_a_compress_shaped_shard(hip_ctx, oracle):
    """BASELINE config 5's shape (core + recursion) as a two-level DAG at test size: level 1 = the core shard proofs of the fibonacci run
    (proven on the GPU, gathered in shard order through the farm's interface); level 2 = one shard of the compress machine's nine chips
    whose program witnesses the level-1 commitments (24 words per core proof), absorbs them with Poseidon2 and commits the resulting digest
    as its public value — proven under the compress prover's FRI configuration (`InnerSC::default()`: log_blowup 1, 84 queries;
    crates/prover/src/lib.rs:192, kb31_poseidon2.rs:203-213), bit-identical
    to the oracle's proof and accepted by the restated verifier. The digest is recomputed independently from the gathered proofs; a
    different core proof gives a different digest. (The real reduce programs come from the reference's recursion compiler; the chips,
    FRI configuration and data flow are the reference's, the program is a stand-in.)"""
    from ziren_amd import farm, prover, recursion as R
    from test_recursion_chips import balanced_shard, recursion_public_values
    m = M.run_machine(program=fibonacci_program(150), shard_cycles=512)
    core_fri = abi.FriConfig(1, 84, 16)
    opk, oshards, proofs = gpu_prove_machine(hip_ctx, oracle, m, core_fri)
    f = farm.Farm()
    gathered = f.gather_proofs(list(range(len(proofs))), proofs, len(proofs))
    inputs = [int(w) for p in gathered for w in F.from_monty(p[:24])]

    def digest_of(words):
        st = np.zeros(16, dtype=np.uint64)
        words = list(words) + [0] * (-len(words) % 8)
        for k in range(0, len(words), 8):
            st[:8] = words[k:k + 8]
            st = F.from_monty(oracle.poseidon2_permute_batch(F.to_monty(st.reshape(1, 16)))[0]).astype(np.uint64)
        return [int(x) for x in st[:8]]

    recs, streams = balanced_shard(300, 200, 40, seed=77, n_var=64, n_select=40, n_poseidon2=10, oracle=oracle, n_exp=10, n_batch_fri=12,
                                   commit_public_values=True, inputs=inputs)
    digest = streams.pop()
    assert digest == digest_of(inputs)
    other = list(inputs)
    other[30] ^= 1
    assert digest_of(other) != digest
    fri = abi.FriConfig(1, 84, 16)
    pv = recursion_public_values(digest)
    igcs = F.to_monty(np.zeros(14, dtype=np.uint64))
    for i, r in enumerate(recs):
        r.prep_index = i
    hp = prover.HipProver(recs, fri, synth.NUM_PV_ELTS, ctx=hip_ctx)
    hp.specialize_quotient_kernels(recs)
    preps = [hip_ctx.tracegen_flat(ins, r.prep_trace.shape[1], r.log_height) for (ins, _), r in zip(streams, recs)]
    lo = [int(r.local_only) for r in recs]
    pk = hp.setup(preps, lo, F.to_monty(0), igcs)
    ropk = oracle.Pk([r.prep_trace for r in recs], lo, F.to_monty(0), igcs, fri.log_blowup)
    assert np.array_equal(pk.commit, ropk.commitment())
    ch = prover.new_challenger()
    pk.observe_into(ch)
    start = ch.copy()
    born = [hip_ctx.tracegen_poseidon2_wide(ev, r.log_height) if r.name == "Poseidon2WideDeg3" else
            hip_ctx.tracegen_exp_reverse_bits(*ev, r.log_height) if r.name == "ExpReverseBitsLen" else
            hip_ctx.tracegen_flat(ev, r.trace.shape[1], r.log_height) if ev is not None else hip_ctx.upload(r.trace)
            for (_, ev), r in zip(streams, recs)]
    proof = hp.prove_shard(pk, pv, born, ch).copy()
    och = oracle.new_challenger()
    ropk.observe_into(och)
    oproof, _ = oracle.prove_shard(ropk, recs, [c.trace for c in recs], pv, fri, synth.NUM_PV_ELTS, och)
    assert np.array_equal(proof, oproof)
    assert oracle.verify_shard(ropk, recs, fri, synth.NUM_PV_ELTS, start.copy(), proof) == 0
    assert oracle.verify_shard(ropk, recs, fri, synth.NUM_PV_ELTS, start.copy(), np.concatenate([proof[:-8], proof[-8:] ^ 1])) != 0
    for t in born:
        t.free()
