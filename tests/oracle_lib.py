"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE (see oracle/oracle_capi.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from ziren_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_last_error.restype = C.c_char_p
        L.orc_challenger_sample.restype = C.c_uint32
        L.orc_challenger_sample_bits.restype = C.c_uint32
        L.orc_challenger_grind.restype = C.c_uint32
        L.orc_two_adic_generator.restype = C.c_uint32
        L.orc_tracegen_alu_width.restype = C.c_size_t
        L.orc_tracegen_alu_check.restype = C.c_long
        # the oracle's OpenMP loops are fine-grained: beyond ~16 threads they get slower (256-thread hosts: 60x)
        L.orc_set_num_threads(min(16, os.cpu_count() or 1))
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())


def _ptr_array(mats):
    arr = (abi.u32p * len(mats))()
    for i, m in enumerate(mats):
        assert m.dtype == np.uint32 and m.flags["C_CONTIGUOUS"]
        arr[i] = abi.as_u32p(m)
    return arr


def _sizes(vals):
    return (C.c_size_t * len(vals))(*vals)


def poseidon2_permute_batch(states):
    s = np.ascontiguousarray(states, dtype=np.uint32).copy()
    lib().orc_poseidon2_permute_batch(abi.as_u32p(s), C.c_size_t(s.shape[0]))
    return s


def hash_slice(v):
    v = np.ascontiguousarray(v, dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint32)
    lib().orc_hash(abi.as_u32p(v), C.c_size_t(len(v)), abi.as_u32p(out))
    return out


def compress(l, r):
    out = np.zeros(8, dtype=np.uint32)
    lib().orc_compress(abi.as_u32p(np.ascontiguousarray(l, dtype=np.uint32)),
                       abi.as_u32p(np.ascontiguousarray(r, dtype=np.uint32)), abi.as_u32p(out))
    return out


def coset_lde_batch(mat, log_blowup, lde_shift):
    mat = np.ascontiguousarray(mat, dtype=np.uint32)
    h, w = mat.shape
    out = np.zeros((h << log_blowup, w), dtype=np.uint32)
    _check(lib().orc_coset_lde_batch(abi.as_u32p(mat), C.c_size_t(h), C.c_size_t(w), C.c_uint32(log_blowup),
                                     C.c_uint32(lde_shift), abi.as_u32p(out)))
    return out


def pcs_commit(mats, log_blowup, domain_shifts=None, want_ldes=False, want_layers=False):
    mats = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
    root = np.zeros(8, dtype=np.uint32)
    hs, ws = [m.shape[0] for m in mats], [m.shape[1] for m in mats]
    ldes = np.zeros(sum((h << log_blowup) * w for h, w in zip(hs, ws)), dtype=np.uint32) if want_ldes else None
    maxh = max(hs) << log_blowup
    layers = np.zeros((2 * maxh - 1) * 8, dtype=np.uint32) if want_layers else None
    sh = None
    if domain_shifts is not None:
        sh = np.ascontiguousarray(domain_shifts, dtype=np.uint32)
    _check(lib().orc_pcs_commit(C.c_size_t(len(mats)), _ptr_array(mats), _sizes(hs), _sizes(ws),
                                abi.as_u32p(sh) if sh is not None else None, C.c_uint32(log_blowup),
                                abi.as_u32p(root), abi.as_u32p(ldes) if want_ldes else None,
                                abi.as_u32p(layers) if want_layers else None))
    out_ldes = None
    if want_ldes:
        out_ldes, pos = [], 0
        for h, w in zip(hs, ws):
            H = h << log_blowup
            out_ldes.append(ldes[pos:pos + H * w].reshape(H, w))
            pos += H * w
    return root, out_ldes, layers


def pcs_open_batch(mats, log_blowup, index, domain_shifts=None):
    mats = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
    hs, ws = [m.shape[0] for m in mats], [m.shape[1] for m in mats]
    values = np.zeros(sum(ws), dtype=np.uint32)
    logmax = (max(hs) << log_blowup).bit_length() - 1
    proof = np.zeros(logmax * 8, dtype=np.uint32)
    ok = C.c_int(0)
    sh = np.ascontiguousarray(domain_shifts, dtype=np.uint32) if domain_shifts is not None else None
    _check(lib().orc_pcs_open_batch(C.c_size_t(len(mats)), _ptr_array(mats), _sizes(hs), _sizes(ws),
                                    abi.as_u32p(sh) if sh is not None else None, C.c_uint32(log_blowup),
                                    C.c_size_t(index), abi.as_u32p(values), abi.as_u32p(proof), C.byref(ok)))
    return values, proof.reshape(logmax, 8), bool(ok.value)


def mmcs_verify_batch(root, heights, widths, index, values, proof):
    root = np.ascontiguousarray(root, dtype=np.uint32)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    proof = np.ascontiguousarray(proof, dtype=np.uint32)
    ok = C.c_int(0)
    _check(lib().orc_mmcs_verify_batch(abi.as_u32p(root), C.c_size_t(len(heights)), _sizes(heights), _sizes(widths),
                                       C.c_size_t(index), abi.as_u32p(values), abi.as_u32p(proof),
                                       C.c_size_t(proof.size // 8), C.byref(ok)))
    return bool(ok.value)


class Pk:
    def __init__(self, prep_traces, local_only, pc_start, igcs, log_blowup):
        prep = [np.ascontiguousarray(m, dtype=np.uint32) for m in prep_traces]
        self.h = C.c_void_p()
        lo = np.ascontiguousarray(local_only, dtype=np.uint32) if len(prep) else np.zeros(1, dtype=np.uint32)
        ig = np.ascontiguousarray(igcs, dtype=np.uint32)
        _check(lib().orc_pk_setup(C.c_size_t(len(prep)), _ptr_array(prep), _sizes([m.shape[0] for m in prep]),
                                  _sizes([m.shape[1] for m in prep]), abi.as_u32p(lo), C.c_uint32(pc_start),
                                  abi.as_u32p(ig), C.c_uint32(log_blowup), C.byref(self.h)))

    def commitment(self):
        out = np.zeros(8, dtype=np.uint32)
        lib().orc_pk_commitment(self.h, abi.as_u32p(out))
        return out

    def observe_into(self, ch):
        lib().orc_pk_observe_into(self.h, C.byref(ch))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pk_free(self.h)
            self.h = None


def new_challenger():
    c = abi.Challenger()
    lib().orc_challenger_init(C.byref(c))
    return c


def challenger_observe(c, vals):
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    lib().orc_challenger_observe(C.byref(c), abi.as_u32p(v), C.c_size_t(len(v)))


def prove_shard(pk, chips, traces, public_values, fri, num_pv_elts, challenger, proof_cap=1 << 24):
    descs, keep = abi.make_chip_descs(chips)
    traces = [np.ascontiguousarray(t, dtype=np.uint32) for t in traces]
    pv = np.ascontiguousarray(public_values, dtype=np.uint32)
    out = np.zeros(proof_cap, dtype=np.uint32)
    plen = C.c_size_t(0)
    timings = (C.c_double * 2)()
    _check(lib().orc_prove_shard(pk.h, C.c_size_t(len(chips)), descs, _ptr_array(traces),
                                 _sizes([t.shape[0] for t in traces]), abi.as_u32p(pv), C.c_size_t(len(pv)),
                                 C.byref(fri), C.c_uint32(num_pv_elts), C.byref(challenger), abi.as_u32p(out),
                                 C.c_size_t(proof_cap), C.byref(plen), timings))
    return out[:plen.value].copy(), (timings[0], timings[1])


def permutation_trace(chip, main, prep, alpha, beta):
    """generate_permutation_trace of one chip (oracle/stark.hpp): main / prep row-major Montgomery, alpha / beta four Montgomery words each.
    Returns (height x 4 perm_ext_width, the cumulative sum)."""
    descs, keep = abi.make_chip_descs([chip])
    main = np.ascontiguousarray(main, dtype=np.uint32)
    prep = None if prep is None else np.ascontiguousarray(prep, dtype=np.uint32)
    ch = np.ascontiguousarray(list(alpha) + list(beta), dtype=np.uint32)
    out = np.zeros((main.shape[0], 4 * chip.perm_ext_width), dtype=np.uint32)
    total = np.zeros(4, dtype=np.uint32)
    _check(lib().orc_permutation_trace(descs, abi.as_u32p(main), abi.as_u32p(prep) if prep is not None else None, C.c_size_t(main.shape[0]), abi.as_u32p(ch),
                                       abi.as_u32p(out), C.c_size_t(out.size), abi.as_u32p(total)))
    return out, total


def verify_shard(pk, chips, fri, num_pv_elts, challenger, proof):
    descs, keep = abi.make_chip_descs(chips)
    proof = np.ascontiguousarray(proof, dtype=np.uint32)
    verdict = C.c_int(-1)
    _check(lib().orc_verify_shard(pk.h, C.c_size_t(len(chips)), descs, C.byref(fri), C.c_uint32(num_pv_elts),
                                  C.byref(challenger), abi.as_u32p(proof), C.c_size_t(len(proof)),
                                  C.byref(verdict)))
    return verdict.value


def tracegen_alu(chip, alu_events, fixed_log2_rows=-1):
    """Row-major Montgomery trace of an ALU chip (oracle/tracegen.hpp)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(alu_events, dtype=E.ALU_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    w = lib().orc_tracegen_alu_width(C.c_int(chip))
    out = np.zeros((rows.value, w), dtype=np.uint32)
    _check(lib().orc_tracegen_alu(C.c_int(chip), C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows),
                                  abi.as_u32p(out), C.c_size_t(out.size)))
    return out


def tracegen_alu_check(chip, alu_events):
    """Index of the first event whose row breaks one of the reference's in-line identities, or -1."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(alu_events, dtype=E.ALU_EVENT)
    return int(lib().orc_tracegen_alu_check(C.c_int(chip), C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev))))


def tracegen_byte_table():
    out = np.zeros((1 << 16, 12), dtype=np.uint32)
    _check(lib().orc_tracegen_byte_table(abi.as_u32p(out)))
    return out


def tracegen_byte_mults(streams, extra_counts=None):
    """streams: [(chip, alu_events)]; extra_counts: optional (65536, 10) plain counts."""
    from ziren_amd import events as E
    evs = [np.ascontiguousarray(ev, dtype=E.ALU_EVENT) for _, ev in streams]
    chips = (C.c_int * len(streams))(*[c for c, _ in streams])
    ptrs = (C.c_void_p * len(streams))(*[ev.ctypes.data for ev in evs])
    ns = _sizes([len(ev) for ev in evs])
    out = np.zeros((1 << 16, 10), dtype=np.uint32)
    ex = np.ascontiguousarray(extra_counts, dtype=np.uint32) if extra_counts is not None else None
    _check(lib().orc_tracegen_byte_mults(C.c_size_t(len(streams)), chips, ptrs, ns,
                                         abi.as_u32p(ex) if ex is not None else None, abi.as_u32p(out)))
    return out


def tracegen_jump(jump_events, fixed_log2_rows=-1):
    from ziren_amd import events as E
    ev = np.ascontiguousarray(jump_events, dtype=E.JUMP_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.JUMP_WIDTH), dtype=np.uint32)
    _check(lib().orc_tracegen_jump(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                   C.c_size_t(out.size)))
    return out


def tracegen_mov_cond(events, fixed_log2_rows=-1):
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.MOV_COND_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.MOV_COND_WIDTH), dtype=np.uint32)
    _check(lib().orc_tracegen_mov_cond(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                       C.c_size_t(out.size)))
    return out


def tracegen_branch(events, fixed_log2_rows=-1, byte_counts=None):
    """Branch chip rows; byte_counts (optional (65536, 10) uint32 array) accumulates the rows' byte lookups in place."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.BRANCH_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.BRANCH_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_branch(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                     C.c_size_t(out.size), bc))
    return out


def tracegen_mul(events, fixed_log2_rows=-1, byte_counts=None):
    """Mul chip rows from CompAluEvents; byte_counts as for tracegen_branch."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.COMP_ALU_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.MUL_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_mul(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                  C.c_size_t(out.size), bc))
    return out


def tracegen_divrem(events, fixed_log2_rows=-1, byte_counts=None):
    """DivRem chip rows from CompAluEvents; byte_counts as for tracegen_branch."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.COMP_ALU_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.DIVREM_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_divrem(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                     C.c_size_t(out.size), bc))
    return out


def tracegen_cpu(events, program, pc_base, shard, fixed_log2_rows=-1, byte_counts=None):
    """Cpu chip rows from CpuEventFfi records and the program (miniexec.CPU_EVENT / INSTRUCTION)."""
    from ziren_amd import miniexec as M
    ev = np.ascontiguousarray(events, dtype=M.CPU_EVENT)
    prog = np.ascontiguousarray(program, dtype=M.INSTRUCTION)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, M.CPU_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_cpu(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_void_p(prog.ctypes.data), C.c_size_t(len(prog)),
                                  C.c_uint32(pc_base), C.c_uint32(shard), C.c_int(fixed_log2_rows), abi.as_u32p(out), C.c_size_t(out.size), bc))
    return out


def tracegen_program(which, events, program, pc_base, fixed_log2_rows=-1):
    """Program chip: which = 0 the preprocessed (pc, instruction) table, 1 the multiplicity column."""
    from ziren_amd import miniexec as M
    ev = np.ascontiguousarray(events, dtype=M.CPU_EVENT)
    prog = np.ascontiguousarray(program, dtype=M.INSTRUCTION)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(prog)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, M.PROGRAM_PREP_WIDTH if which == 0 else 1), dtype=np.uint32)
    _check(lib().orc_tracegen_program(C.c_int(which), C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_void_p(prog.ctypes.data),
                                      C.c_size_t(len(prog)), C.c_uint32(pc_base), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                      C.c_size_t(out.size)))
    return out


def tracegen_memory_local(events, fixed_log2_rows=-1):
    """MemoryLocal chip rows from MemoryLocalEvents (miniexec.MEMORY_LOCAL_EVENT), four per row."""
    from ziren_amd import miniexec as M
    ev = np.ascontiguousarray(events, dtype=M.MEMORY_LOCAL_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_memory_local(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), None, C.c_size_t(0), C.byref(rows)))
    out = np.zeros((rows.value, M.MEMORY_LOCAL_WIDTH), dtype=np.uint32)
    _check(lib().orc_tracegen_memory_local(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                           C.c_size_t(out.size), C.byref(rows)))
    return out


def tracegen_memory_instrs(events, fixed_log2_rows=-1, byte_counts=None):
    """MemoryInstructions chip rows from MemInstrEvents; byte_counts as for tracegen_branch."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.MEM_INSTR_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.MEMORY_INSTRS_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_memory_instrs(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                            C.c_size_t(out.size), bc))
    return out


def _rows_then_fill(fn, width, *args, tail=()):
    rows = C.c_size_t()
    _check(fn(*args, None, C.c_size_t(0), C.byref(rows), *([None] * len(tail))))
    out = np.zeros((rows.value, width), dtype=np.uint32)
    _check(fn(*args, abi.as_u32p(out), C.c_size_t(out.size), C.byref(rows), *tail))
    return out


def tracegen_memory_global(events, previous_addr=0, fixed_log2_rows=-1):
    """MemoryGlobalInit / MemoryGlobalFinalize rows from MemoryInitializeFinalizeEvents (events.MEMORY_INIT_FINALIZE_EVENT); previous_addr =
    the last address of the previous shard's chip (public values)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.MEMORY_INIT_FINALIZE_EVENT)
    return _rows_then_fill(lib().orc_tracegen_memory_global, E.MEMORY_GLOBAL_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_uint32(previous_addr), C.c_int(fixed_log2_rows))


def tracegen_syscall(events, precompile, fixed_log2_rows=-1, byte_counts=None):
    """SyscallCore (precompile False: filtered as the chip filters) / SyscallPrecompile rows from SyscallEvents."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.SYSCALL_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_syscall, E.SYSCALL_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(int(precompile)),
                           C.c_int(fixed_log2_rows), tail=bc)


def tracegen_poseidon2_permute(events, fixed_log2_rows=-1, byte_counts=None):
    """Poseidon2Permute precompile rows from flattened Poseidon2PermuteEvents (events.POSEIDON2_PERMUTE_EVENT)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.POSEIDON2_PERMUTE_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_poseidon2_permute, E.POSEIDON2_PERMUTE_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(fixed_log2_rows), tail=bc)


def tracegen_keccak_sponge(blocks, fixed_log2_rows=-1, byte_counts=None):
    """KeccakSponge precompile rows from KeccakSpongeEvents cut into 36-word blocks (events.KECCAK_SPONGE_BLOCK), 24 rows per block."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(blocks, dtype=E.KECCAK_SPONGE_BLOCK)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_keccak_sponge, E.KECCAK_SPONGE_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(fixed_log2_rows), tail=bc)


def tracegen_sha_extend(events, fixed_log2_rows=-1, byte_counts=None):
    """ShaExtend precompile rows from flattened ShaExtendEvents (events.SHA_EXTEND_EVENT), 48 rows per event."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.SHA_EXTEND_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_sha_extend, E.SHA_EXTEND_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(fixed_log2_rows), tail=bc)


def tracegen_sha_compress(events, fixed_log2_rows=-1, byte_counts=None):
    """ShaCompress precompile rows from flattened ShaCompressEvents (events.SHA_COMPRESS_EVENT), 80 rows per event."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.SHA_COMPRESS_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_sha_compress, E.SHA_COMPRESS_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(fixed_log2_rows), tail=bc)


def tracegen_ed_add(events, fixed_log2_rows=-1, byte_counts=None):
    """EdAddAssign precompile rows from flattened EllipticCurveAddEvents (events.ED_ADD_EVENT), one row per event."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.ED_ADD_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_ed_add, E.ED_ADD_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_ed_decompress(events, fixed_log2_rows=-1, byte_counts=None):
    """EdDecompress precompile rows from flattened EdDecompressEvents (events.ED_DECOMPRESS_EVENT), one row per event."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.ED_DECOMPRESS_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_ed_decompress, E.ED_DECOMPRESS_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_weierstrass(curve, double, events, fixed_log2_rows=-1, byte_counts=None):
    """<Curve>AddAssign / <Curve>DoubleAssign rows from flattened EllipticCurveAddEvents / EllipticCurveDoubleEvents (events.weierstrass_event_dtypes)."""
    from ziren_amd import events as E
    c = E.WEIERSTRASS_CURVES[curve]
    dt = E.weierstrass_event_dtypes(curve)[1 if double else 0]
    ev = np.ascontiguousarray(events, dtype=dt)
    n = c["n_limbs"]
    mod = (C.c_uint8 * n)(*c["p"].to_bytes(n, "little"))
    a = (C.c_uint8 * n)(*c["a"].to_bytes(n, "little"))
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_weierstrass, E.weierstrass_widths(curve)[1 if double else 0], C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(int(double)), C.c_int(n), mod, a, C.c_uint32(c["witness_offset"]), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_uint256_mul(events, fixed_log2_rows=-1, byte_counts=None):
    """Uint256MulMod rows from flattened Uint256MulEvents (events.UINT256_MUL_EVENT)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.UINT256_MUL_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_uint256_mul, E.UINT256_MUL_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_u256x2048_mul(events, fixed_log2_rows=-1, byte_counts=None):
    """U256XU2048Mul rows from flattened U256xU2048MulEvents (events.U256X2048_MUL_EVENT)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.U256X2048_MUL_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_u256x2048_mul, E.U256X2048_MUL_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_boolean_circuit_garble(rows, fixed_log2_rows=-1, byte_counts=None):
    """BooleanCircuitGarble rows from row records (events.GARBLE_ROW: header rows and gate rows)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(rows, dtype=E.GARBLE_ROW)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_boolean_circuit_garble, E.GARBLE_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_sys_linux(events, fixed_log2_rows=-1, byte_counts=None):
    """SysLinux rows from flattened LinuxEvents (events.LINUX_EVENT)."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.LINUX_EVENT)
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_sys_linux, E.SYS_LINUX_WIDTH, C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), tail=bc)


def tracegen_weierstrass_decompress(curve, events, fixed_log2_rows=-1, byte_counts=None):
    """<Curve>Decompress rows from flattened EllipticCurveDecompressEvents (events.weierstrass_decompress_event_dtype)."""
    from ziren_amd import events as E
    c = E.WEIERSTRASS_CURVES[curve]
    ev = np.ascontiguousarray(events, dtype=E.weierstrass_decompress_event_dtype(curve))
    n = c["n_limbs"]
    as_bytes = lambda v: (C.c_uint8 * n)(*v.to_bytes(n, "little"))      # noqa: E731
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_weierstrass_decompress, E.weierstrass_decompress_width(curve), C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(n), as_bytes(c["p"]), as_bytes(c["a"]), as_bytes(c["b"]), as_bytes(c["generator"][0]), C.c_uint32(c["witness_offset"]),
                           C.c_int(int(E.WEIERSTRASS_DECOMPRESS[curve]["lexicographic"])), C.c_int(fixed_log2_rows), tail=bc)


FP_TOWER_KINDS = {"fp": 0, "fp2_addsub": 1, "fp2_mul": 2}


def tracegen_fp_tower(field, kind, events, fixed_log2_rows=-1, byte_counts=None):
    """<Field>FpOpAssign / Fp2AddSubAssign / Fp2MulAssign rows from flattened FpOpEvents / Fp2AddSubEvents / Fp2MulEvents (events.fp_tower_event_dtype)."""
    from ziren_amd import events as E
    c = E.WEIERSTRASS_CURVES[field]
    ev = np.ascontiguousarray(events, dtype=E.fp_tower_event_dtype(field, kind))
    n = c["n_limbs"]
    mod = (C.c_uint8 * n)(*c["p"].to_bytes(n, "little"))
    bc = (abi.as_u32p(byte_counts),) if byte_counts is not None else (None,)
    return _rows_then_fill(lib().orc_tracegen_fp_tower, E.fp_tower_width(field, kind), C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)),
                           C.c_int(FP_TOWER_KINDS[kind]), C.c_int(n), mod, C.c_uint32(c["witness_offset"]), C.c_int(fixed_log2_rows), tail=bc)


def septic_known_answers(a, b):
    """((z^i)^p, (z^i)^(p^2) for i = 1..6, a * b, normalised sqrt(a^2)) in the septic extension, canonical words."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    out = np.zeros(98, dtype=np.uint32)
    _check(lib().orc_septic_known_answers(abi.as_u32p(a), abi.as_u32p(b), abi.as_u32p(out)))
    return out[:42].reshape(6, 7), out[42:84].reshape(6, 7), out[84:91], out[91:]


def global_digest_sum(digests):
    """Machine::verify's check over global_cumulative_sums ((n, 14) Montgomery words): (their SepticDigest sum, is it the zero digest)."""
    d = np.ascontiguousarray(digests, dtype=np.uint32).reshape(-1, 14)
    out = np.zeros(14, dtype=np.uint32)
    z = C.c_int(0)
    _check(lib().orc_global_digest_sum(abi.as_u32p(d), C.c_size_t(len(d)), abi.as_u32p(out), C.byref(z)))
    return out, bool(z.value)


def tracegen_global(events, fixed_log2_rows=-1, byte_counts=None):
    """Global chip rows from GlobalLookupEvents; byte_counts as for tracegen_branch."""
    from ziren_amd import miniexec as M
    ev = np.ascontiguousarray(events, dtype=M.GLOBAL_LOOKUP_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, M.GLOBAL_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_global(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                     C.c_size_t(out.size), bc))
    return out


def tracegen_poseidon2_wide(events, fixed_log2_rows=-1):
    """Recursion Poseidon2Wide (degree 3) rows from events of 32 Montgomery words (input[16], output[16])."""
    from ziren_amd import recursion as R
    ev = np.ascontiguousarray(events, dtype=np.uint32).reshape(-1, 32)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, R.POSEIDON2_WIDE_WIDTH), dtype=np.uint32)
    _check(lib().orc_tracegen_poseidon2_wide(abi.as_u32p(ev), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out), C.c_size_t(out.size)))
    return out


def tracegen_syscall_instrs(events, fixed_log2_rows=-1):
    """SyscallInstrs chip rows from SyscallEvents."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.SYSCALL_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.SYSCALL_INSTRS_WIDTH), dtype=np.uint32)
    _check(lib().orc_tracegen_syscall_instrs(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                             C.c_size_t(out.size)))
    return out


def tracegen_misc_instrs(events, fixed_log2_rows=-1, byte_counts=None):
    """MiscInstrs chip rows from MiscEvents; byte_counts as for tracegen_branch."""
    from ziren_amd import events as E
    ev = np.ascontiguousarray(events, dtype=E.MISC_EVENT)
    rows = C.c_size_t()
    _check(lib().orc_tracegen_alu_rows(C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), C.byref(rows)))
    out = np.zeros((rows.value, E.MISC_INSTRS_WIDTH), dtype=np.uint32)
    bc = abi.as_u32p(byte_counts) if byte_counts is not None else None
    _check(lib().orc_tracegen_misc_instrs(C.c_void_p(ev.ctypes.data), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), abi.as_u32p(out),
                                          C.c_size_t(out.size), bc))
    return out


def tracegen_exp_reverse_bits(bases, bits, offsets, fixed_log2_rows=-1):
    """Recursion ExpReverseBitsLen rows: bases (n), all exponent bits end to end, offsets (n + 1); Montgomery words."""
    bases, bits = np.ascontiguousarray(bases, dtype=np.uint32), np.ascontiguousarray(bits, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    rows = C.c_size_t()
    n = len(bases)
    args = (abi.as_u32p(bases), abi.as_u32p(bits), abi.as_u32p(offsets), C.c_size_t(n), C.c_int(fixed_log2_rows))
    _check(lib().orc_tracegen_exp_reverse_bits(*args, None, C.c_size_t(0), C.byref(rows)))
    out = np.zeros((rows.value, 7), dtype=np.uint32)
    _check(lib().orc_tracegen_exp_reverse_bits(*args, abi.as_u32p(out), C.c_size_t(out.size), C.byref(rows)))
    return out


def tracegen_poseidon2_skinny(events, fixed_log2_rows=-1):
    """Recursion Poseidon2Skinny rows (eleven per event) from events of 32 Montgomery words (input[16], output[16])."""
    ev = np.ascontiguousarray(events, dtype=np.uint32).reshape(-1, 32)
    rows = C.c_size_t()
    args = (abi.as_u32p(ev), C.c_size_t(len(ev)), C.c_int(fixed_log2_rows))
    _check(lib().orc_tracegen_poseidon2_skinny(*args, None, C.c_size_t(0), C.byref(rows)))
    out = np.zeros((rows.value, 28), dtype=np.uint32)
    _check(lib().orc_tracegen_poseidon2_skinny(*args, abi.as_u32p(out), C.c_size_t(out.size), C.byref(rows)))
    return out
