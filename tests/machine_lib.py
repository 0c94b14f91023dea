"""Test infrastructure: a whole machine run (ziren_amd/miniexec.run_machine) shard by shard — which chips a shard includes
(MachineAir::included), their traces from the oracle or from the device, the shard's global messages — and the machine-level
verifier restated from the reference:

  ZKMProver::verify            crates/prover/src/verify.rs:60-296      shard numbering, pc chaining, exit code, memory
                                                                       init / finalize address chaining, digest chaining
  StarkMachine::verify         crates/stark/src/machine.rs:619-675     every shard proof verifies; the global cumulative sums and the
                                                                       key's initial one add up to the zero digest
"""
import numpy as np

from ziren_amd import air, chips, events as E, field as F, miniexec as M

KIND_MEMORY, KIND_SYSCALL, KIND_SYSCALL_RESULT = 1, 6, 8     # LookupKind (crates/stark/src/lookup/lookup.rs:22-48)


def log2_rows(n):
    h = 16
    while h < n:
        h <<= 1
    return h.bit_length() - 1


def halves(x):
    return x & 0xffff, x >> 16


def syscall_global_events(syscall_events, precompile):
    """SyscallChip::generate_dependencies (syscall/chip.rs:115-187): per event two messages — the arguments as half-words (kind Syscall) and
    the result half-words (kind SyscallResult; zero unless the call is a Linux syscall) — sent by Core, received by Precompile."""
    ev = syscall_events
    if not precompile:
        code = ev["a_record"]["prev_value"]
        ev = ev[(((code >> 16) & 0xff) == 1) | (((code >> 8) & 0xff) != 0)]
    out = np.zeros(2 * len(ev), dtype=M.GLOBAL_LOOKUP_EVENT)
    for i, e in enumerate(ev):
        a1, a2 = halves(int(e["arg1"])), halves(int(e["arg2"]))
        out["message"][2 * i] = [e["shard"], e["clk"], e["syscall_id"], a1[0], a1[1], a2[0], a2[1]]
        linux = (int(e["a_record"]["prev_value"]) >> 8) & 0xff != 0      # a Linux syscall: the result's half-words (chip.rs:134-157)
        res = halves(int(e["a_record"]["value"])) if linux else (0, 0)
        out["message"][2 * i + 1] = [e["shard"], e["clk"], e["syscall_id"], res[0], res[1], 0, 0]
        out["kind"][2 * i], out["kind"][2 * i + 1] = KIND_SYSCALL, KIND_SYSCALL_RESULT
    out["is_receive"] = 1 if precompile else 0
    return out


def memory_global_events(events, finalize):
    """MemoryGlobalChip::generate_dependencies (memory/global.rs:62-99): Initialize sends (0, 0, addr, value bytes), Finalize receives
    (shard, timestamp, addr, value bytes); sorted by address."""
    ev = np.sort(events, order="addr")
    out = np.zeros(len(ev), dtype=M.GLOBAL_LOOKUP_EVENT)
    m = out["message"]
    if finalize:
        m[:, 0], m[:, 1] = ev["shard"], ev["timestamp"]
    m[:, 2] = ev["addr"]
    for k in range(4):
        m[:, 3 + k] = (ev["value"] >> (8 * k)) & 0xff
    out["is_receive"] = 1 if finalize else 0
    out["kind"] = KIND_MEMORY
    return out


class Oracle:
    """Trace provider: the CPU restatement (tests/oracle_lib)."""

    def __init__(self, O):
        self.O = O
        self.counts = np.zeros((1 << 16, 10), dtype=np.uint32)

    def trace(self, what, *a):
        O, c = self.O, self.counts
        if what == "cpu":
            ev, prog, pc_base, shard, lh = a
            return O.tracegen_cpu(ev, prog, pc_base, shard, lh, c)
        if what == "alu":
            return O.tracegen_alu(a[0], a[1], a[2])      # the ALU chips' byte lookups are counted from the events (byte_mults)
        fn = {"syscall_instrs": O.tracegen_syscall_instrs, "jump": O.tracegen_jump, "mov_cond": O.tracegen_mov_cond, "memory_local": O.tracegen_memory_local}
        if what in fn:
            return fn[what](a[0], a[1])
        fn = {"branch": O.tracegen_branch, "memory_instrs": O.tracegen_memory_instrs, "misc_instrs": O.tracegen_misc_instrs, "mul": O.tracegen_mul,
              "divrem": O.tracegen_divrem, "global": O.tracegen_global, "poseidon2_permute": O.tracegen_poseidon2_permute,
              "keccak_sponge": O.tracegen_keccak_sponge, "sha_extend": O.tracegen_sha_extend, "sha_compress": O.tracegen_sha_compress,
              "ed_add": O.tracegen_ed_add, "ed_decompress": O.tracegen_ed_decompress, "uint256_mul": O.tracegen_uint256_mul, "u256x2048_mul": O.tracegen_u256x2048_mul,
              "garble": O.tracegen_boolean_circuit_garble, "linux": O.tracegen_sys_linux}
        if what in fn:
            return fn[what](a[0], a[1], c)
        if what == "weierstrass":
            return O.tracegen_weierstrass(a[0], a[1], a[2], a[3], c)
        if what == "weierstrass_decompress":
            return O.tracegen_weierstrass_decompress(a[0], a[1], a[2], c)
        if what == "fp_tower":
            return O.tracegen_fp_tower(a[0], a[1], a[2], a[3], c)
        if what == "syscall_table":
            return O.tracegen_syscall(a[0], a[1], a[2], c)
        if what == "memory_global":
            return O.tracegen_memory_global(a[0], a[1], a[2])
        raise KeyError(what)

    def byte_trace(self, alu_streams):
        return self.O.tracegen_byte_mults(alu_streams, self.counts)

    def program_mults(self, cpu, prog, pc_base, lh):
        return self.O.tracegen_program(1, cpu, prog, pc_base, lh)


class Device:
    """Trace provider: the HIP library (traces stay on the device; .to_host() for comparison)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.blu = ctx.byte_lookups()

    def trace(self, what, *a):
        ctx, blu = self.ctx, self.blu
        if what == "cpu":
            ev, prog, pc_base, shard, lh = a
            return ctx.tracegen_cpu(ev, prog, pc_base, shard, lh, blu)
        if what == "alu":
            return ctx.tracegen_alu(a[0], a[1], a[2], blu)
        fn = {"syscall_instrs": ctx.tracegen_syscall_instrs, "jump": ctx.tracegen_jump, "mov_cond": ctx.tracegen_mov_cond, "memory_local": ctx.tracegen_memory_local}
        if what in fn:
            return fn[what](a[0], a[1])
        fn = {"branch": ctx.tracegen_branch, "memory_instrs": ctx.tracegen_memory_instrs, "misc_instrs": ctx.tracegen_misc_instrs, "mul": ctx.tracegen_mul,
              "divrem": ctx.tracegen_divrem, "global": ctx.tracegen_global, "poseidon2_permute": ctx.tracegen_poseidon2_permute,
              "keccak_sponge": ctx.tracegen_keccak_sponge, "sha_extend": ctx.tracegen_sha_extend, "sha_compress": ctx.tracegen_sha_compress,
              "ed_add": ctx.tracegen_ed_add, "ed_decompress": ctx.tracegen_ed_decompress, "uint256_mul": ctx.tracegen_uint256_mul, "u256x2048_mul": ctx.tracegen_u256x2048_mul,
              "garble": ctx.tracegen_boolean_circuit_garble, "linux": ctx.tracegen_sys_linux}
        if what in fn:
            return fn[what](a[0], a[1], blu)
        if what == "weierstrass":
            return ctx.tracegen_weierstrass(a[0], a[1], a[2], a[3], blu)
        if what == "weierstrass_decompress":
            return ctx.tracegen_weierstrass_decompress(a[0], a[1], a[2], blu)
        if what == "fp_tower":
            return ctx.tracegen_fp_tower(a[0], a[1], a[2], a[3], blu)
        if what == "syscall_table":
            return ctx.tracegen_syscall(a[0], a[1], a[2], blu)
        if what == "memory_global":
            return ctx.tracegen_memory_global(a[0], a[1], a[2])
        raise KeyError(what)

    def byte_trace(self, alu_streams):
        return self.ctx.tracegen_byte_mults(self.blu)

    def program_mults(self, cpu, prog, pc_base, lh):
        return self.ctx.tracegen_program_mults(cpu, len(prog), pc_base, lh)


class _Pending:
    def __init__(self, i):
        self.i = i


class DeviceOneCall(Device):
    """Trace provider: the HIP library with a CPU shard's generators queued in ONE call (zkm_tracegen_shard) — what a farm lane does. `trace`
    hands back placeholders; `flush(chips)` makes the call (optionally from events copied ahead with zkm_events_upload_async, so that
    nothing is read on the host) and puts the device matrices in their place. Chips the shard call does not cover (the precompiles, the
    memory tables) go through their own entry points at once, counting into the same byte-lookup table."""

    def __init__(self, ctx, prefetch=False):
        super().__init__(ctx)
        self.items, self.prefetch = [], prefetch

    def _defer(self, item):
        self.items.append(item)
        return _Pending(len(self.items) - 1)

    def trace(self, what, *a):
        from ziren_amd import abi
        if what == "cpu":
            ev, prog, pc_base, shard, lh = a
            return self._defer((abi.TG_CPU, ev, lh, {"program": prog, "pc_base": pc_base, "shard": shard}))
        if what == "alu":
            return self._defer((abi.TG_ALU, a[1], a[2], {"chip": a[0]}))
        simple = {"syscall_instrs": abi.TG_SYSCALL_INSTRS, "jump": abi.TG_JUMP, "mov_cond": abi.TG_MOV_COND, "memory_local": abi.TG_MEMORY_LOCAL,
                  "branch": abi.TG_BRANCH, "memory_instrs": abi.TG_MEMORY_INSTRS, "misc_instrs": abi.TG_MISC_INSTRS, "mul": abi.TG_MUL, "divrem": abi.TG_DIVREM,
                  "global": abi.TG_GLOBAL}
        if what in simple:
            return self._defer((simple[what], a[0], a[1], None))
        if what == "syscall_table":
            return self._defer((abi.TG_SYSCALL_PRECOMPILE if a[1] else abi.TG_SYSCALL_CORE, a[0], a[2], None))
        return super().trace(what, *a)

    def byte_trace(self, alu_streams):
        from ziren_amd import abi
        return self._defer((abi.TG_BYTE_MULTS, None, 16, None))

    def program_mults(self, cpu, prog, pc_base, lh):
        from ziren_amd import abi
        if not any(it[0] == abi.TG_CPU for it in self.items):
            return super().program_mults(cpu, prog, pc_base, lh)
        return self._defer((abi.TG_PROGRAM_MULTS, None, lh, None))

    def flush(self, chip_list):
        items = self.items
        held = []
        if self.prefetch:
            items = []
            for kind, ev, lh, extra in self.items:
                if ev is not None and len(ev):
                    ev = self.ctx.events_upload_async(np.ascontiguousarray(ev))
                    held.append(ev)
                items.append((kind, ev, lh, extra))
        born = self.ctx.tracegen_shard(items, self.blu) if items else []
        for d in held:
            d.free()
        for c in chip_list:
            if isinstance(c.trace, _Pending):
                c.trace = born[c.trace.i]
        self.items = []


def build_shard(src, machine, k, shape=None):
    """The chips shard k includes (MachineAir::included: a chip with no events is left out; Program and Byte are always in) with their
    traces from `src`. Returns the RecordedChips in machine order, Byte and Program last (prep indices 0 and 1 of the key).
    `shape` ({MipsAirId name: log2 height}, a cpu shard only): the record's fixed shape (CoreShapeConfig::fix_shape) — a chip is in when the
    shape names it, events or not, and every trace is padded to the shape's height."""
    sh = machine.shards[k]
    prog, pc_base, shard_no = machine.program, machine.pc_base, sh.pv["shard"]
    rec = sh.record
    out, alu_streams = [], []
    glob = []

    def add(chip, trace):
        chip.trace = trace
        out.append(chip)

    def height(name, rows):
        """log2 height of a chip, or None when the shard leaves it out."""
        if shape is None:
            return log2_rows(rows) if rows else None
        if name not in shape:
            assert not rows, (name, rows)
            return None
        assert rows <= (1 << shape[name]), (name, rows)
        return shape[name]

    assert shape is None or sh.kind == "cpu"
    if sh.kind == "cpu":
        rec = M.add_dependencies(rec)
        lh = height("Cpu", len(rec.cpu))
        add(chips.record_cpu_chip(lh), src.trace("cpu", rec.cpu, prog, pc_base, shard_no, lh))
        for chip in sorted(E.CHIP_NAMES):
            ev = rec.alu[chip]
            lh = height(E.CHIP_NAMES[chip], len(ev))
            if lh is not None:
                add(chips.record_chip(chip, lh), src.trace("alu", chip, ev, lh))
                alu_streams.append((chip, ev))
        for name, air, ev, record in (("syscall_instrs", "SyscallInstrs", rec.syscall, chips.record_syscall_instrs_chip), ("jump", "Jump", rec.jump, chips.record_jump_chip),
                                      ("mov_cond", "MovCond", rec.mov_cond, chips.record_mov_cond_chip), ("branch", "Branch", rec.branch, chips.record_branch_chip),
                                      ("memory_instrs", "MemoryInstrs", rec.mem_instr, chips.record_memory_instrs_chip),
                                      ("misc_instrs", "MiscInstrs", rec.misc, chips.record_misc_instrs_chip),
                                      ("mul", "Mul", rec.mul, chips.record_mul_chip), ("divrem", "DivRem", rec.divrem, chips.record_divrem_chip)):
            lh = height(air, len(ev))
            if lh is not None:
                add(record(lh), src.trace(name, ev, lh))
        to_table = syscall_global_events(rec.syscall, False)
        lh = height("SyscallCore", len(to_table) // 2)
        if lh is not None:
            add(chips.record_syscall_table_chip(False, lh), src.trace("syscall_table", rec.syscall, False, lh))
            if len(to_table):
                glob.append(to_table)
    elif sh.kind == "precompile":
        lh = log2_rows(len(rec.precompile_syscall))
        add(chips.record_syscall_table_chip(True, lh), src.trace("syscall_table", rec.precompile_syscall, True, lh))
        glob.append(syscall_global_events(rec.precompile_syscall, True))
        if len(rec.poseidon2_permute):
            lh = log2_rows(len(rec.poseidon2_permute))
            add(chips.record_poseidon2_permute_chip(lh), src.trace("poseidon2_permute", rec.poseidon2_permute, lh))
        if len(rec.keccak_sponge):
            lh = log2_rows(24 * len(rec.keccak_sponge))
            add(chips.record_keccak_sponge_chip(lh), src.trace("keccak_sponge", rec.keccak_sponge, lh))
        if len(rec.sha_extend):
            lh = log2_rows(48 * len(rec.sha_extend))
            add(chips.record_sha_extend_chip(lh), src.trace("sha_extend", rec.sha_extend, lh))
        if len(rec.sha_compress):
            lh = log2_rows(80 * len(rec.sha_compress))
            add(chips.record_sha_compress_chip(lh), src.trace("sha_compress", rec.sha_compress, lh))
        if len(rec.ed_add):
            lh = log2_rows(len(rec.ed_add))
            add(chips.record_ed_add_chip(lh), src.trace("ed_add", rec.ed_add, lh))
        if len(rec.ed_decompress):
            lh = log2_rows(len(rec.ed_decompress))
            add(chips.record_ed_decompress_chip(lh), src.trace("ed_decompress", rec.ed_decompress, lh))
        if len(getattr(rec, "uint256_mul", ())):
            lh = log2_rows(len(rec.uint256_mul))
            add(chips.record_uint256_mul_chip(lh), src.trace("uint256_mul", rec.uint256_mul, lh))
        if len(getattr(rec, "u256x2048_mul", ())):
            lh = log2_rows(len(rec.u256x2048_mul))
            add(chips.record_u256x2048_mul_chip(lh), src.trace("u256x2048_mul", rec.u256x2048_mul, lh))
        if len(getattr(rec, "garble", ())):
            lh = log2_rows(len(rec.garble))
            add(chips.record_boolean_circuit_garble_chip(lh), src.trace("garble", rec.garble, lh))
        if len(getattr(rec, "linux", ())):
            lh = log2_rows(len(rec.linux))
            add(chips.record_sys_linux_chip(lh), src.trace("linux", rec.linux, lh))
        if rec.weierstrass is not None:
            kind, ev = rec.weierstrass
            curve, double = kind.split("_")[0], kind.endswith("_double")
            lh = log2_rows(len(ev))
            add(chips.record_weierstrass_chip(curve, double, lh), src.trace("weierstrass", curve, double, ev, lh))
        if getattr(rec, "weierstrass_decompress", None) is not None:
            curve, ev = rec.weierstrass_decompress
            lh = log2_rows(len(ev))
            add(chips.record_weierstrass_decompress_chip(curve, lh), src.trace("weierstrass_decompress", curve, ev, lh))
        if getattr(rec, "fp_tower", None) is not None:
            kind_key, ev = rec.fp_tower
            field, kind = kind_key.split("_", 1)
            lh = log2_rows(len(ev))
            add(chips.record_fp_tower_chip(field, kind, lh), src.trace("fp_tower", field, kind, ev, lh))
    else:
        for finalize, ev, prev in ((False, rec.memory_init, sh.pv["previous_init_addr"]), (True, rec.memory_finalize, sh.pv["previous_finalize_addr"])):
            if len(ev):
                lh = log2_rows(len(ev))
                add(chips.record_memory_global_chip(finalize, lh), src.trace("memory_global", ev, prev, lh))
                glob.append(memory_global_events(ev, finalize))
    if sh.kind != "memory" and (len(rec.memory_local) or (shape is not None and "MemoryLocal" in shape)):
        rows = -(-len(rec.memory_local) // M.MEMORY_LOCAL_ENTRIES_PER_ROW)
        lh = log2_rows(rows) if shape is None else height("MemoryLocal", rows)
        add(chips.record_memory_local_chip(lh), src.trace("memory_local", rec.memory_local, lh))
        glob.insert(0, M.global_lookup_events(rec.memory_local))
    ge = np.concatenate(glob) if glob else np.zeros(0, dtype=M.GLOBAL_LOOKUP_EVENT)
    if len(ge) or (shape is not None and "Global" in shape):
        lh = log2_rows(len(ge)) if shape is None else height("Global", len(ge))
        add(chips.record_global_chip(lh), src.trace("global", ge, lh))
    byte = chips.record_byte_chip(prep_index=0)
    byte.trace = src.byte_trace(alu_streams)
    plh = log2_rows(len(prog)) if shape is None else shape["Program"]
    program = chips.record_program_chip(plh, prep_index=1)
    program.trace = src.program_mults(rec.cpu, prog, pc_base, plh)
    return out + [byte, program]


def shard_public_values(sh):
    return M.public_values(sh.pv)


# ---- the machine-level verifier -------------------------------------------------------------------------------------------------------

def _pv(proof_pv):
    v = F.from_monty(np.asarray(proof_pv, dtype=np.uint32))
    bits = lambda base: int(sum(int(v[base + i]) << i for i in range(32)))   # noqa: E731
    return {"digest": [int(x) for x in v[0:32]], "deferred": [int(x) for x in v[32:40]], "start_pc": int(v[40]), "next_pc": int(v[41]), "exit_code": int(v[42]),
            "shard": int(v[43]), "execution_shard": int(v[44]), "prev_init": bits(45), "last_init": bits(77), "prev_fin": bits(109), "last_fin": bits(141)}


def verify_public_values(shards, vk_pc_start):
    """ZKMProver::verify (crates/prover/src/verify.rs:60-290) up to the call of StarkMachine::verify. `shards`: per shard proof
    (public values as Montgomery words, set of chip names). Returns None or the reference's error string."""
    if not shards:
        return "empty proof"
    pvs = [(_pv(p), names) for p, names in shards]
    if "Cpu" not in pvs[0][1]:
        return "missing cpu in first shard"
    for i, (pv, _) in enumerate(pvs):                               # :92-104
        if pv["shard"] != i + 1:
            return "shard index should be the previous shard index + 1 and start at 1"
    ex = 0
    for pv, names in pvs:                                           # :114-128
        if "Cpu" in names:
            ex += 1
            if pv["execution_shard"] != ex:
                return "execution shard index should be the previous execution shard index + 1 if cpu exists and start at 1"
    prev_next = 0
    for i, (pv, names) in enumerate(pvs):                           # :141-169
        if i == 0 and pv["start_pc"] != vk_pc_start:
            return "start_pc != vk.start_pc"
        if i != 0 and pv["start_pc"] != prev_next:
            return "start_pc != next_pc_prev"
        if "Cpu" not in names and pv["start_pc"] != pv["next_pc"]:
            return "start_pc != next_pc for a non-cpu shard"
        if "Cpu" in names and pv["start_pc"] == 0:
            return "start_pc == 0"
        if i == len(pvs) - 1 and pv["next_pc"] != 0:
            return "next_pc != 0: execution should have halted"
        prev_next = pv["next_pc"]
    if any(pv["exit_code"] != 0 for pv, _ in pvs):                  # :174-183
        return "exit_code != 0"
    last_init = last_fin = 0
    for pv, names in pvs:                                           # :199-231
        if pv["prev_init"] != last_init:
            return "previous_init_addr_bits != last_init_addr_bits_prev"
        if pv["prev_fin"] != last_fin:
            return "previous_finalize_addr_bits != last_finalize_addr_bits_prev"
        if "MemoryGlobalInit" not in names and pv["prev_init"] != pv["last_init"]:
            return "previous_init_addr_bits != last_init_addr_bits"
        if "MemoryGlobalFinalize" not in names and pv["prev_fin"] != pv["last_fin"]:
            return "previous_finalize_addr_bits != last_finalize_addr_bits"
        last_init, last_fin = pv["last_init"], pv["last_fin"]
    prev_digest, prev_deferred = [0] * 32, [0] * 8
    for pv, names in pvs:                                           # :251-283
        if any(prev_digest) and pv["digest"] != prev_digest:
            return "committed_value_digest != committed_value_digest_prev"
        if any(prev_deferred) and pv["deferred"] != prev_deferred:
            return "deferred_proofs_digest != deferred_proofs_digest_prev"
        if "Cpu" not in names and (pv["digest"] != prev_digest or pv["deferred"] != prev_deferred):
            return "digests changed in a non-cpu shard"
        prev_digest, prev_deferred = pv["digest"], pv["deferred"]
    if len(pvs) > 1 << 16:
        return "too many shards"
    return None


def decode_shard_proof(stream):
    """The fields of a flat ShardProof stream the machine-level checks read: per chip the caller index, log degree
    and global cumulative sum, and the public values."""
    w = np.asarray(stream, dtype=np.uint32)
    pos = 24
    n_chips = int(w[pos]); pos += 1
    out = {"chips": []}
    for _ in range(n_chips):
        idx, logd = int(w[pos]), int(w[pos + 1]); pos += 2
        for _ in range(3):                      # preprocessed, main, permutation: width, local[width x 4], next[width x 4]
            width = int(w[pos]); pos += 1 + 8 * width
        n_chunks = int(w[pos]); pos += 1 + 16 * n_chunks
        gcs = w[pos:pos + 14].copy(); pos += 14 + 4
        out["chips"].append({"index": idx, "log_degree": logd, "global_cumulative_sum": gcs})
    n_pv = int(w[-232]) if len(w) > 232 else 0
    assert n_pv == 231, "the stream ends with n_public_values = 231 and the values"
    out["public_values"] = w[-231:].copy()
    return out


def verify_machine(O, opk, shard_chip_lists, proofs, fri, num_pv_elts, vk_pc_start, initial_global_cumulative_sum):
    """ZKMProver::verify + StarkMachine::verify: public-value checks, every shard proof through the restated shard verifier (the shard's
    challenger = the key's challenger after observing the shard's public values happens inside verify_shard), and the digest sum."""
    decoded = [decode_shard_proof(p) for p in proofs]
    shards = [(d["public_values"], {chips_[c["index"]].name for c in d["chips"]}) for d, chips_ in zip(decoded, shard_chip_lists)]
    err = verify_public_values(shards, vk_pc_start)
    if err:
        return err
    start = O.new_challenger()
    opk.observe_into(start)
    for k, (p, chips_) in enumerate(zip(proofs, shard_chip_lists)):
        rc = O.verify_shard(opk, chips_, fri, num_pv_elts, start.copy(), p)
        if rc != 0:
            return f"shard {k + 1}: invalid shard proof (code {rc})"
    digests = []
    for d in decoded:           # ShardProof::global_cumulative_sum: the sum over the shard's chips (only Global's is not the zero digest)
        digests += [c["global_cumulative_sum"] for c in d["chips"]]
    digests.append(np.asarray(initial_global_cumulative_sum, dtype=np.uint32))
    if not O.global_digest_sum(digests)[1]:
        return "global cumulative sum is not zero"
    return None
