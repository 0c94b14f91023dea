"""HipProver — host-side mirror of Ziren's `MachineProver` for the MI355X back-end.

Mirrors the trait of crates/stark/src/prover.rs:30-184 (names, argument meaning, error
behaviour): `setup` / `pk_to_device`, `commit(record, traces) -> ShardMainData`,
`open(pk, data, challenger) -> ShardProof`, `prove`. In production the host is Ziren's Rust
SDK binding the same C ABI; this Python layer exists so parity tests and the
benchmark read like the reference's own `run_test::<P>` (crates/core/machine/src/utils/prove.rs:614).

Everything numeric runs in libzkm_hip.so on the GPU; this module only marshals.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import abi, lib


class DeviceMatrix:
    """`MachineProver::DeviceMatrix`: a trace resident in HBM, column-major."""

    def __init__(self, ctx: "Context", handle, height, width):
        self.ctx, self.h, self.height, self.width = ctx, handle, height, width

    def wait(self):
        lib.check(lib.load().zkm_matrix_wait(self.ctx.h, self.h))

    def to_host(self) -> np.ndarray:
        out = np.empty((self.height, self.width), dtype=np.uint32)
        lib.check(lib.load().zkm_matrix_download(self.ctx.h, self.h, abi.as_u32p(out)))
        return out

    def free(self):
        if self.h and self.ctx.h:             # a closed context has already released everything it owned
            lib.load().zkm_matrix_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ByteLookups:
    """`record.byte_lookups` on the device (zkm_byte_lookups)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    def free(self):
        if self.h and self.ctx.h:
            lib.load().zkm_byte_lookups_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceEvents:
    """Executor events copied to the device ahead of their trace generation (zkm_events_upload_async). Pass it to a core-shard
    `Context.tracegen_*` call in place of the numpy array; `host` (page-locked for a true asynchronous copy) is kept alive with it."""

    def __init__(self, ctx, ptr, host):
        self.ctx, self.ptr, self.host, self.dtype = ctx, ptr, host, host.dtype

    def __len__(self):
        return len(self.host)

    def free(self):
        if self.ptr and self.ctx.h:           # a closed context has already released everything it owned
            lib.load().zkm_events_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _evptr(events, dtype):
    """(address, count, keep-alive) of an event vector: a numpy array (host pointer) or a DeviceEvents (device address)."""
    if isinstance(events, DeviceEvents):
        if events.dtype != dtype:
            raise TypeError(f"events of dtype {events.dtype} handed to a chip that reads {dtype}")
        return C.c_void_p(events.ptr if len(events) else None), len(events), events
    ev = events if (isinstance(events, np.ndarray) and events.dtype == dtype and events.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(events, dtype=dtype)
    return C.c_void_p(ev.ctypes.data if len(ev) else None), len(ev), ev


class Context:
    """One per GPU (`zkm_ctx`)."""

    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        lib.check(lib.load().zkm_ctx_create(C.c_int(device), C.byref(self.h)))
        self.device = device

    def upload(self, host_row_major: np.ndarray) -> DeviceMatrix:
        m = np.ascontiguousarray(host_row_major, dtype=np.uint32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_matrix_upload(self.h, abi.as_u32p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]),
                                               C.byref(h)))
        return DeviceMatrix(self, h, m.shape[0], m.shape[1])

    def upload_async(self, host_row_major: np.ndarray) -> DeviceMatrix:
        """zkm_matrix_upload_async: queue the copy and return; consumers wait for the matrix on the device. The array
        (page-locked for real overlap: host_alloc) must stay alive and unchanged until the matrix has been consumed."""
        m = host_row_major
        assert m.dtype == np.uint32 and m.flags["C_CONTIGUOUS"]
        h = C.c_void_p()
        lib.check(lib.load().zkm_matrix_upload_async(self.h, abi.as_u32p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]),
                                                     C.byref(h)))
        dm = DeviceMatrix(self, h, m.shape[0], m.shape[1])
        dm._host = m
        return dm

    def events_upload_async(self, events: np.ndarray) -> DeviceEvents:
        """zkm_events_upload_async: queue the copy of an event vector (page-locked: host_alloc / fibfast.DeviceShard.pin) on the DMA stream."""
        assert events.flags["C_CONTIGUOUS"]
        d = C.c_void_p()
        lib.check(lib.load().zkm_events_upload_async(self.h, C.c_void_p(events.ctypes.data if len(events) else None), C.c_size_t(events.nbytes), C.byref(d)))
        return DeviceEvents(self, d.value, events)

    def permutation_trace(self, chip, main: DeviceMatrix, prep: Optional[DeviceMatrix], alpha, beta):
        """generate_permutation_trace of one chip on the device (zkm_permutation_trace, a test entry point): returns (DeviceMatrix of
        height x 4 perm_ext_width, the cumulative sum as four Montgomery words)."""
        descs, keep = abi.make_chip_descs([chip])
        ch = np.ascontiguousarray(list(alpha) + list(beta), dtype=np.uint32)
        total = np.zeros(4, dtype=np.uint32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_permutation_trace(self.h, descs, main.h, prep.h if prep is not None else None, abi.as_u32p(ch), C.byref(h), abi.as_u32p(total)))
        return self._born(h), total

    def byte_lookups(self) -> "ByteLookups":
        """An empty `record.byte_lookups` on the device (zkm_byte_lookups_create)."""
        h = C.c_void_p()
        lib.check(lib.load().zkm_byte_lookups_create(self.h, C.byref(h)))
        return ByteLookups(self, h)

    def tracegen_alu(self, chip: int, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of an ALU chip on the device (zkm_tracegen_alu); `events` has dtype events.ALU_EVENT.
        With `blu`, the chip's `generate_dependencies` runs in the same pass: its byte lookups are counted into it."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.ALU_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_alu(self.h, C.c_int(chip), p_ev, C.c_size_t(n_ev), C.c_int(fixed_log2_rows), blu.h if blu is not None else None,
                                              C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_jump(self, events: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the Jump chip on the device (zkm_tracegen_jump); dtype events.JUMP_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.JUMP_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_jump(self.h, p_ev, C.c_size_t(n_ev),
                                               C.c_int(fixed_log2_rows), C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_branch(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the Branch chip on the device (zkm_tracegen_branch); dtype events.BRANCH_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.BRANCH_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_branch(self.h, p_ev, C.c_size_t(n_ev),
                                                 C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_mul(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` (+ `generate_dependencies` into `blu`) of the Mul chip on the device (zkm_tracegen_mul); dtype
        events.COMP_ALU_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.COMP_ALU_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_mul(self.h, p_ev, C.c_size_t(n_ev),
                                              C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_divrem(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` (which also records the byte lookups, into `blu`) of the DivRem chip on the device
        (zkm_tracegen_divrem); dtype events.COMP_ALU_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.COMP_ALU_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_divrem(self.h, p_ev, C.c_size_t(n_ev),
                                                 C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def _born(self, h) -> DeviceMatrix:
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_cpu(self, events: np.ndarray, program: np.ndarray, pc_base: int, shard: int, fixed_log2_rows: int = -1,
                     blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` + `generate_dependencies` of the Cpu chip on the device (zkm_tracegen_cpu); dtypes
        miniexec.CPU_EVENT (CpuEventFfi) and miniexec.INSTRUCTION (InstructionFfi)."""
        from . import miniexec as _m
        p_ev, n_ev, _keep = _evptr(events, _m.CPU_EVENT)
        prog = np.ascontiguousarray(program, dtype=_m.INSTRUCTION)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_cpu(self.h, p_ev, C.c_size_t(n_ev),
                                              C.c_void_p(prog.ctypes.data if len(prog) else None), C.c_size_t(len(prog)),
                                              C.c_uint32(pc_base), C.c_uint32(shard), C.c_int(fixed_log2_rows),
                                              blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_memory_instrs(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the MemoryInstructions chip on the device (zkm_tracegen_memory_instrs); dtype events.MEM_INSTR_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.MEM_INSTR_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_memory_instrs(self.h, p_ev, C.c_size_t(n_ev),
                                                        C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_poseidon2_wide(self, events: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the recursion Poseidon2Wide chip (degree 3) on the device (zkm_tracegen_poseidon2_wide); events:
        32 Montgomery words each (input[16], output[16])."""
        ev = np.ascontiguousarray(events, dtype=np.uint32).reshape(-1, 32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_poseidon2_wide(self.h, abi.as_u32p(ev) if len(ev) else None, C.c_size_t(len(ev)),
                                                         C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_syscall_instrs(self, events: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the SyscallInstrs chip on the device (zkm_tracegen_syscall_instrs); dtype events.SYSCALL_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.SYSCALL_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_syscall_instrs(self.h, p_ev, C.c_size_t(n_ev),
                                                         C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_syscall(self, events: np.ndarray, precompile: bool, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the SyscallCore / SyscallPrecompile tables on the device (zkm_tracegen_syscall); dtype events.SYSCALL_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.SYSCALL_EVENT)      # a DeviceEvents: SyscallCore's filter runs on the device
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_syscall(self.h, p_ev, C.c_size_t(n_ev), C.c_int(int(precompile)),
                                                  C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_memory_global(self, events: np.ndarray, previous_addr: int = 0, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of MemoryGlobalInit / MemoryGlobalFinalize on the device (zkm_tracegen_memory_global); dtype
        events.MEMORY_INIT_FINALIZE_EVENT; previous_addr = the previous shard's last address (public values)."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.MEMORY_INIT_FINALIZE_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_memory_global(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                        C.c_uint32(previous_addr), C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_poseidon2_permute(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the Poseidon2Permute precompile on the device (zkm_tracegen_poseidon2_permute); dtype events.POSEIDON2_PERMUTE_EVENT."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.POSEIDON2_PERMUTE_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_poseidon2_permute(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                            C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_keccak_sponge(self, blocks: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the KeccakSponge precompile on the device (zkm_tracegen_keccak_sponge); dtype events.KECCAK_SPONGE_BLOCK, the
        calls' 36-word blocks in order, 24 rows each."""
        from . import events as _ev
        ev = np.ascontiguousarray(blocks, dtype=_ev.KECCAK_SPONGE_BLOCK)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_keccak_sponge(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                        C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_sha_extend(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the ShaExtend precompile on the device (zkm_tracegen_sha_extend); dtype events.SHA_EXTEND_EVENT, 48 rows each."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.SHA_EXTEND_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_sha_extend(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                     C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_sha_compress(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the ShaCompress precompile on the device (zkm_tracegen_sha_compress); dtype events.SHA_COMPRESS_EVENT, 80 rows each."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.SHA_COMPRESS_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_sha_compress(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                       C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_ed_add(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the EdAddAssign precompile on the device (zkm_tracegen_ed_add); dtype events.ED_ADD_EVENT, one row each."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.ED_ADD_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_ed_add(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                 C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_ed_decompress(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the EdDecompress precompile on the device (zkm_tracegen_ed_decompress); dtype events.ED_DECOMPRESS_EVENT."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.ED_DECOMPRESS_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_ed_decompress(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                        C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_weierstrass(self, curve: str, double: bool, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of <Curve>AddAssign / <Curve>DoubleAssign on the device (zkm_tracegen_weierstrass_add / _double); curve one of
        events.WEIERSTRASS_CURVES, dtype events.weierstrass_event_dtypes(curve)."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.weierstrass_event_dtypes(curve)[1 if double else 0])
        h = C.c_void_p()
        fn = lib.load().zkm_tracegen_weierstrass_double if double else lib.load().zkm_tracegen_weierstrass_add
        lib.check(fn(self.h, C.c_int(_ev.WEIERSTRASS_CURVES[curve]["index"]), C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                     C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_weierstrass_decompress(self, curve: str, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of Secp256k1Decompress / Secp256r1Decompress / Bls12381Decompress on the device (zkm_tracegen_weierstrass_decompress);
        dtype events.weierstrass_decompress_event_dtype(curve)."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.weierstrass_decompress_event_dtype(curve))
        if curve not in _ev.WEIERSTRASS_DECOMPRESS:
            raise ValueError("no decompress chip for " + curve)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_weierstrass_decompress(self.h, C.c_int(_ev.WEIERSTRASS_CURVES[curve]["index"]), C.c_void_p(ev.ctypes.data if len(ev) else None),
                                                                 C.c_size_t(len(ev)), C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_uint256_mul(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the Uint256MulMod precompile on the device (zkm_tracegen_uint256_mul); dtype events.UINT256_MUL_EVENT, one row each."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.UINT256_MUL_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_uint256_mul(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                      C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_u256x2048_mul(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the U256XU2048Mul precompile on the device (zkm_tracegen_u256x2048_mul); dtype events.U256X2048_MUL_EVENT, one row each."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.U256X2048_MUL_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_u256x2048_mul(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                        C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_boolean_circuit_garble(self, rows: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the BooleanCircuitGarble precompile on the device (zkm_tracegen_boolean_circuit_garble); dtype events.GARBLE_ROW, one
        record per row (a header row and a row per gate for every call)."""
        from . import events as _ev
        ev = np.ascontiguousarray(rows, dtype=_ev.GARBLE_ROW)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_boolean_circuit_garble(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                                 C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_sys_linux(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the SysLinux chip on the device (zkm_tracegen_sys_linux); dtype events.LINUX_EVENT, one row each."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.LINUX_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_sys_linux(self.h, C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                                                    C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_fp_tower(self, field: str, kind: str, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of <Field>FpOpAssign / Fp2AddSubAssign / Fp2MulAssign on the device (zkm_tracegen_fp_op / _fp2_addsub / _fp2_mul); field
        "Bn254" or "Bls12381", kind "fp" / "fp2_addsub" / "fp2_mul", dtype events.fp_tower_event_dtype(field, kind)."""
        from . import events as _ev
        ev = np.ascontiguousarray(events, dtype=_ev.fp_tower_event_dtype(field, kind))
        h = C.c_void_p()
        L = lib.load()
        fn = {"fp": L.zkm_tracegen_fp_op, "fp2_addsub": L.zkm_tracegen_fp2_addsub, "fp2_mul": L.zkm_tracegen_fp2_mul}[kind]
        lib.check(fn(self.h, C.c_int(_ev.WEIERSTRASS_CURVES[field]["index"]), C.c_void_p(ev.ctypes.data if len(ev) else None), C.c_size_t(len(ev)),
                     C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_misc_instrs(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the MiscInstrs chip on the device (zkm_tracegen_misc_instrs); dtype events.MISC_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.MISC_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_misc_instrs(self.h, p_ev, C.c_size_t(n_ev),
                                                      C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_poseidon2_skinny(self, events: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the recursion Poseidon2Skinny chip on the device (zkm_tracegen_poseidon2_skinny)."""
        ev = np.ascontiguousarray(events, dtype=np.uint32).reshape(-1, 32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_poseidon2_skinny(self.h, abi.as_u32p(ev) if len(ev) else None, C.c_size_t(len(ev)),
                                                           C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_exp_reverse_bits(self, bases: np.ndarray, bits: np.ndarray, offsets: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the recursion ExpReverseBitsLen chip on the device (zkm_tracegen_exp_reverse_bits)."""
        bases, bits = np.ascontiguousarray(bases, dtype=np.uint32), np.ascontiguousarray(bits, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_exp_reverse_bits(self.h, abi.as_u32p(bases) if len(bases) else None, abi.as_u32p(bits) if len(bits) else None,
                                                           abi.as_u32p(offsets), C.c_size_t(len(bases)), C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_memory_local(self, events: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the MemoryLocal chip on the device (zkm_tracegen_memory_local); dtype miniexec.MEMORY_LOCAL_EVENT."""
        from . import miniexec as _m
        p_ev, n_ev, _keep = _evptr(events, _m.MEMORY_LOCAL_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_memory_local(self.h, p_ev, C.c_size_t(n_ev),
                                                       C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_global(self, events: np.ndarray, fixed_log2_rows: int = -1, blu: "ByteLookups" = None) -> DeviceMatrix:
        """`generate_trace` of the Global chip on the device (zkm_tracegen_global); dtype miniexec.GLOBAL_LOOKUP_EVENT. The
        U16Range lookups of the messages' first words are counted into `blu`."""
        from . import miniexec as _m
        p_ev, n_ev, _keep = _evptr(events, _m.GLOBAL_LOOKUP_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_global(self.h, p_ev, C.c_size_t(n_ev),
                                                 C.c_int(fixed_log2_rows), blu.h if blu is not None else None, C.byref(h)))
        return self._born(h)

    def tracegen_cpu_and_program(self, events: np.ndarray, program: np.ndarray, pc_base: int, shard: int, fixed_log2_rows: int = -1,
                                 program_fixed_log2_rows: int = -1, blu: "ByteLookups" = None):
        """The Cpu trace and, from the same upload of the events, the Program chip's multiplicity trace
        (zkm_tracegen_cpu_and_program). Returns (cpu, program_mults)."""
        from . import miniexec as _m
        p_ev, n_ev, _keep = _evptr(events, _m.CPU_EVENT)
        prog = np.ascontiguousarray(program, dtype=_m.INSTRUCTION)
        h, hp = C.c_void_p(), C.c_void_p()
        lib.check(lib.load().zkm_tracegen_cpu_and_program(
            self.h, p_ev, C.c_size_t(n_ev), C.c_void_p(prog.ctypes.data if len(prog) else None),
            C.c_size_t(len(prog)), C.c_uint32(pc_base), C.c_uint32(shard), C.c_int(fixed_log2_rows), C.c_int(program_fixed_log2_rows),
            blu.h if blu is not None else None, C.byref(h), C.byref(hp)))
        return self._born(h), self._born(hp)

    def tracegen_program(self, program: np.ndarray, pc_base: int, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """The Program chip's preprocessed table on the device (zkm_tracegen_program)."""
        from . import miniexec as _m
        prog = np.ascontiguousarray(program, dtype=_m.INSTRUCTION)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_program(self.h, C.c_void_p(prog.ctypes.data if len(prog) else None), C.c_size_t(len(prog)),
                                                  C.c_uint32(pc_base), C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_program_mults(self, events: np.ndarray, n_instr: int, pc_base: int, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """The Program chip's multiplicity trace on the device (zkm_tracegen_program_mults)."""
        from . import miniexec as _m
        p_ev, n_ev, _keep = _evptr(events, _m.CPU_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_program_mults(self.h, p_ev, C.c_size_t(n_ev),
                                                        C.c_size_t(n_instr), C.c_uint32(pc_base), C.c_int(fixed_log2_rows), C.byref(h)))
        return self._born(h)

    def tracegen_mov_cond(self, events: np.ndarray, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """`generate_trace` of the MovCond chip on the device (zkm_tracegen_mov_cond); dtype events.MOV_COND_EVENT."""
        from . import events as _ev
        p_ev, n_ev, _keep = _evptr(events, _ev.MOV_COND_EVENT)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_mov_cond(self.h, p_ev, C.c_size_t(n_ev),
                                                   C.c_int(fixed_log2_rows), C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_flat(self, words: np.ndarray, width: int, fixed_log2_rows: int = -1) -> DeviceMatrix:
        """Padded trace of a chip whose rows are its records end to end (zkm_tracegen_flat): recursion BaseAlu / ExtAlu."""
        w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1)
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_flat(self.h, abi.as_u32p(w) if len(w) else None, C.c_size_t(len(w)), C.c_size_t(width),
                                               C.c_int(fixed_log2_rows), C.byref(h)))
        L = lib.load()
        return DeviceMatrix(self, h, int(L.zkm_matrix_height(h)), int(L.zkm_matrix_width(h)))

    def tracegen_byte_table(self) -> DeviceMatrix:
        """`ByteChip::trace()`: the Byte chip's 65536 x 12 preprocessed table, generated on the device."""
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_byte_table(self.h, C.byref(h)))
        return DeviceMatrix(self, h, 1 << 16, 12)

    def tracegen_shard(self, items, blu: "ByteLookups" = None):
        """generate_traces of a core shard in one call (zkm_tracegen_shard): `items` = [(kind, events, fixed_log2_rows, extra)] with kind one
        of abi.TG_*, events a numpy array or a DeviceEvents (None for TG_BYTE_MULTS / TG_PROGRAM_MULTS), extra = {"chip": alu chip} for
        TG_ALU, {"program", "pc_base", "shard"} for TG_CPU. Every generator is queued behind the copy of its events and the call
        synchronises once. Returns one DeviceMatrix per item."""
        from . import events as _ev, miniexec as _m
        dtypes = {abi.TG_ALU: _ev.ALU_EVENT, abi.TG_CPU: _m.CPU_EVENT, abi.TG_BRANCH: _ev.JUMP_EVENT, abi.TG_JUMP: _ev.JUMP_EVENT,
                  abi.TG_MOV_COND: _ev.MOV_COND_EVENT, abi.TG_MUL: _ev.COMP_ALU_EVENT, abi.TG_DIVREM: _ev.COMP_ALU_EVENT, abi.TG_MEMORY_INSTRS: _ev.MEM_INSTR_EVENT,
                  abi.TG_MISC_INSTRS: _ev.MISC_EVENT, abi.TG_SYSCALL_INSTRS: _ev.SYSCALL_EVENT, abi.TG_SYSCALL_CORE: _ev.SYSCALL_EVENT,
                  abi.TG_SYSCALL_PRECOMPILE: _ev.SYSCALL_EVENT, abi.TG_MEMORY_LOCAL: _m.MEMORY_LOCAL_EVENT, abi.TG_GLOBAL: _m.GLOBAL_LOOKUP_EVENT}
        descs = (abi.TracegenDesc * len(items))()
        keep = []
        for d, (kind, events, lh, extra) in zip(descs, items):
            d.kind, d.fixed_log2_rows, d.chip = kind, lh, -1
            if kind in dtypes:
                p_ev, n_ev, k = _evptr(events, dtypes[kind])
                keep.append(k)
                d.events, d.n_events = p_ev.value, n_ev
            if kind in (abi.TG_FLAT, abi.TG_POSEIDON2_WIDE, abi.TG_EXP_REVERSE_BITS):
                # the recursion machine's chips: plain words. TG_FLAT: extra = {"width"}; TG_POSEIDON2_WIDE: 32 words per permutation;
                # TG_EXP_REVERSE_BITS: one buffer [bases | offsets | bits], extra = {"n": instructions, "rows": offsets[n]}
                if isinstance(events, DeviceEvents):
                    p_ev, n_words, k = C.c_void_p(events.ptr), len(events.host), events
                else:
                    k = np.ascontiguousarray(events if events is not None else np.zeros(0, dtype=np.uint32), dtype=np.uint32).reshape(-1)
                    p_ev, n_words = C.c_void_p(k.ctypes.data if len(k) else None), len(k)
                keep.append(k)
                d.events = p_ev.value
                if kind == abi.TG_FLAT:
                    d.n_events, d.chip = n_words, extra["width"]
                elif kind == abi.TG_POSEIDON2_WIDE:
                    d.n_events = n_words // 32
                else:
                    d.n_events, d.n_instr = extra["n"], extra["rows"]
            if kind == abi.TG_ALU:
                d.chip = extra["chip"]
            if kind == abi.TG_CPU:
                prog = np.ascontiguousarray(extra["program"], dtype=_m.INSTRUCTION)
                keep.append(prog)
                d.program, d.n_instr, d.pc_base, d.shard = (prog.ctypes.data if len(prog) else None), len(prog), extra["pc_base"], extra["shard"]
        out = (C.c_void_p * len(items))()
        lib.check(lib.load().zkm_tracegen_shard(self.h, descs, C.c_size_t(len(items)), blu.h if blu is not None else None, out))
        return [self._born(C.c_void_p(h)) for h in out]

    def tracegen_byte_mults(self, blu: "ByteLookups", extra_counts=None) -> DeviceMatrix:
        """`ByteChip::generate_trace` over the lookups counted in `blu` (+ optional (65536, 10) host counts)."""
        ex = np.ascontiguousarray(extra_counts, dtype=np.uint32) if extra_counts is not None else None
        h = C.c_void_p()
        lib.check(lib.load().zkm_tracegen_byte_mults(self.h, blu.h, abi.as_u32p(ex) if ex is not None else None, C.byref(h)))
        return DeviceMatrix(self, h, 1 << 16, 10)

    def host_alloc(self, shape) -> np.ndarray:
        """uint32 array in page-locked host memory (zkm_host_alloc); free with host_free."""
        n = int(np.prod(shape))
        p = lib.load().zkm_host_alloc(self.h, C.c_size_t(max(n, 1) * 4))
        if not p:
            raise lib.ZkmError("zkm_host_alloc failed")
        arr = np.ctypeslib.as_array((C.c_uint32 * max(n, 1)).from_address(p))[:n].reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            lib.load().zkm_host_free(self.h, C.c_void_p(p))

    def synchronize(self):
        lib.check(lib.load().zkm_ctx_synchronize(self.h))

    def set_memory_limit(self, nbytes: int):
        """zkm_ctx_set_memory_limit: cap what the context's pool may hold (0: none); over it a call fails with "out of device memory"."""
        lib.check(lib.load().zkm_ctx_set_memory_limit(self.h, C.c_size_t(int(nbytes))))

    def memory_held(self) -> int:
        return int(lib.load().zkm_ctx_memory_held(self.h))

    def trim(self):
        lib.check(lib.load().zkm_ctx_trim(self.h))

    def last_timings(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = lib.load().zkm_ctx_last_timings(self.h, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(min(n, 64))]

    def kernel_timings(self):
        """[(kernel, total ms, launches, compulsory HBM bytes)] of the last commit/open/prove call."""
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        calls = (C.c_uint32 * 64)()
        nbytes = (C.c_double * 64)()
        n = lib.load().zkm_ctx_kernel_timings(self.h, names, ms, calls, nbytes, 64)
        return [(names[i].decode(), float(ms[i]), int(calls[i]), float(nbytes[i])) for i in range(min(n, 64))]

    def close(self):
        if self.h:
            lib.load().zkm_ctx_destroy(self.h)
            self.h = None


def _handles(mats: Sequence[DeviceMatrix]):
    arr = (C.c_void_p * len(mats))()
    for i, m in enumerate(mats):
        arr[i] = m.h
    return arr


class PcsData:
    """`DeviceProverData` (LDEs + Merkle tree)."""

    def __init__(self, ctx, handle, root, mats, log_blowup):
        self.ctx, self.h, self.root, self.log_blowup = ctx, handle, root, log_blowup
        self.shapes = [(m.height << log_blowup, m.width) for m in mats]
        self._keep = list(mats)

    def lde(self, idx) -> np.ndarray:
        out = np.empty(self.shapes[idx], dtype=np.uint32)
        lib.check(lib.load().zkm_pcs_data_get_lde(self.ctx.h, self.h, C.c_size_t(idx), abi.as_u32p(out)))
        return out

    def open_batch(self, index):
        logmax = max(s[0] for s in self.shapes).bit_length() - 1
        values = np.zeros(sum(s[1] for s in self.shapes), dtype=np.uint32)
        proof = np.zeros((logmax, 8), dtype=np.uint32)
        lib.check(lib.load().zkm_pcs_open_batch(self.ctx.h, self.h, C.c_size_t(index), abi.as_u32p(values),
                                                abi.as_u32p(proof)))
        return values, proof

    def free(self):
        if self.h and self.ctx.h:
            lib.load().zkm_pcs_data_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pcs_commit(ctx: Context, mats: Sequence[DeviceMatrix], log_blowup: int, domain_shifts=None) -> PcsData:
    """`Pcs::commit` (prover.rs:277)."""
    root = np.zeros(8, dtype=np.uint32)
    h = C.c_void_p()
    sh = np.ascontiguousarray(domain_shifts, dtype=np.uint32) if domain_shifts is not None else None
    lib.check(lib.load().zkm_pcs_commit(ctx.h, C.c_size_t(len(mats)), _handles(mats),
                                        abi.as_u32p(sh) if sh is not None else None, C.c_uint32(log_blowup),
                                        abi.as_u32p(root), C.byref(h)))
    return PcsData(ctx, h, root, mats, log_blowup)


@dataclass
class ShardMainData:
    """crates/stark/src/types.rs:16-35"""
    handle: C.c_void_p
    main_commit: np.ndarray
    chip_ordering: List[int]
    public_values: np.ndarray
    traces: List[DeviceMatrix]


class ProvingKey:
    """`DeviceProvingKey` (StarkProvingKey, crates/stark/src/machine.rs:58-75)."""

    def __init__(self, ctx, handle, prep):
        self.ctx, self.h, self._prep = ctx, handle, prep

    @property
    def commit(self):
        out = np.zeros(8, dtype=np.uint32)
        lib.load().zkm_pk_commitment(self.h, abi.as_u32p(out))
        return out

    def observe_into(self, challenger: abi.Challenger):
        lib.load().zkm_pk_observe_into(self.h, C.byref(challenger))

    def free(self):
        if self.h and self.ctx.h:
            lib.load().zkm_pk_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def new_challenger() -> abi.Challenger:
    """`config.challenger()`"""
    c = abi.Challenger()
    lib.load().zkm_challenger_init(C.byref(c))
    return c


class HipProver:
    """`MachineProver<KoalaBearPoseidon2, A>` over libzkm_hip.

    `chips` plays the role of the `StarkMachine`: per-chip metadata + recorded constraints
    (objects with the fields of synth.SynChip).
    """

    def __init__(self, chips, fri: abi.FriConfig, num_pv_elts: int, device: int = 0, ctx: Optional[Context] = None,
                 specialize: bool = False):
        self.chips = list(chips)
        self.fri = fri
        self.num_pv_elts = num_pv_elts
        self.ctx = ctx or Context(device)
        self._descs, self._keep = abi.make_chip_descs(self.chips)
        if specialize:
            self.specialize_quotient_kernels()

    def specialize_quotient_kernels(self, chips=None):
        """Compile (or load from the in-tree cache) one quotient kernel per chip AIR and register it; chips without
        one go through the bytecode interpreter."""
        from . import codegen
        for c in (self.chips if chips is None else chips):
            prog = np.ascontiguousarray(c.program, dtype=np.uint32)
            try:
                co = codegen.specialize(prog)
            except Exception as e:  # noqa: BLE001  the generator or hipcc failed on this AIR: the library's bytecode interpreter evaluates it (same arithmetic, slower)
                import warnings
                warnings.warn(f"no specialised quotient kernel for chip {getattr(c, 'name', '?')} ({type(e).__name__}: {e}); the interpreter evaluates it")
                continue
            if co is None:          # too long for a straight-line kernel: the interpreter evaluates it
                continue
            buf = C.create_string_buffer(co, len(co))
            lib.check(lib.load().zkm_ctx_register_quotient_kernel(self.ctx.h, abi.as_u32p(prog), C.c_uint32(len(prog)), buf,
                                                                  C.c_size_t(len(co))))
        if os.environ.get("ZKM_NO_PERM_KERNELS") != "1":
            self.specialize_perm_kernels(chips)

    def specialize_perm_kernels(self, chips=None):
        """The same for the permutation traces: one generated kernel per chip's lookups (zkm_ctx_register_perm_kernel); chips without one
        go through the generic kernel."""
        from . import codegen
        for c in (self.chips if chips is None else chips):
            blob = np.ascontiguousarray(c.lookups_blob, dtype=np.uint32)
            try:
                co = codegen.specialize_perm(blob, c.log_quotient_degree)
            except Exception as e:  # noqa: BLE001
                import warnings
                warnings.warn(f"no specialised permutation kernel for chip {getattr(c, 'name', '?')} ({type(e).__name__}: {e}); the generic kernel runs")
                continue
            if co is None:
                continue
            buf = C.create_string_buffer(co, len(co))
            lib.check(lib.load().zkm_ctx_register_perm_kernel(self.ctx.h, abi.as_u32p(blob), C.c_uint32(len(blob)), C.c_uint32(c.log_quotient_degree),
                                                              buf, C.c_size_t(len(co))))

    # fn setup / pk_to_device (prover.rs:54-66)
    def setup(self, prep_traces: Sequence[np.ndarray], prep_local_only, pc_start, initial_global_cumulative_sum) -> ProvingKey:
        prep = [t if isinstance(t, DeviceMatrix) else self.ctx.upload(t) for t in prep_traces]  # device-born tables pass through
        lo = np.ascontiguousarray(prep_local_only if len(prep) else [0], dtype=np.uint32)
        ig = np.ascontiguousarray(initial_global_cumulative_sum, dtype=np.uint32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_pk_setup(self.ctx.h, C.c_size_t(len(prep)), _handles(prep), abi.as_u32p(lo),
                                          C.c_uint32(int(pc_start)), abi.as_u32p(ig), C.c_uint32(self.fri.log_blowup),
                                          C.byref(h)))
        return ProvingKey(self.ctx, h, prep)

    def upload_traces(self, traces: Sequence[np.ndarray]) -> List[DeviceMatrix]:
        return [self.ctx.upload(t) for t in traces]

    # fn commit (prover.rs:258-292)
    def commit(self, public_values: np.ndarray, traces: Sequence[DeviceMatrix]) -> ShardMainData:
        # a HipProver is built for one shard's chip list (the reference's shard_chips_ordered): traces pair with self.chips one to one
        if len(traces) != len(self.chips):
            raise ValueError(f"commit: {len(traces)} traces for {len(self.chips)} chips — build the prover with the chips this shard includes")
        names = (C.c_char_p * len(self.chips))(*[c.name.encode() for c in self.chips])
        pv = np.ascontiguousarray(public_values, dtype=np.uint32)
        root = np.zeros(8, dtype=np.uint32)
        order = np.zeros(len(self.chips), dtype=np.uint32)
        h = C.c_void_p()
        lib.check(lib.load().zkm_commit(self.ctx.h, C.c_size_t(len(traces)), names, _handles(traces), abi.as_u32p(pv),
                                        C.c_size_t(len(pv)), C.c_uint32(self.fri.log_blowup), abi.as_u32p(root),
                                        abi.as_u32p(order), C.byref(h)))
        return ShardMainData(h, root, [int(x) for x in order], pv, list(traces))

    # fn open (prover.rs:298-653)
    def open(self, pk: ProvingKey, data: ShardMainData, challenger: abi.Challenger, proof_cap: int = 1 << 24) -> np.ndarray:
        out = np.zeros(proof_cap, dtype=np.uint32)
        plen = C.c_size_t(0)
        try:
            lib.check(lib.load().zkm_open(self.ctx.h, pk.h, data.handle, self._descs, C.byref(self.fri),
                                          C.c_uint32(self.num_pv_elts), C.byref(challenger), abi.as_u32p(out),
                                          C.c_size_t(proof_cap), C.byref(plen)))
        finally:
            lib.load().zkm_main_data_free(self.ctx.h, data.handle)  # `open` consumes ShardMainData
            data.handle = None
        return out[:plen.value].copy()

    # commit + open on device-resident traces
    def prove_shard(self, pk: ProvingKey, public_values: np.ndarray, traces: Sequence[DeviceMatrix],
                    challenger: abi.Challenger, proof_cap: int = 1 << 24, out: Optional[np.ndarray] = None) -> np.ndarray:
        pv = np.ascontiguousarray(public_values, dtype=np.uint32)
        if out is None:
            out = np.zeros(proof_cap, dtype=np.uint32)
        plen = C.c_size_t(0)
        lib.check(lib.load().zkm_prove_shard(self.ctx.h, pk.h, C.c_size_t(len(traces)), self._descs, _handles(traces),
                                             abi.as_u32p(pv), C.c_size_t(len(pv)), C.byref(self.fri),
                                             C.c_uint32(self.num_pv_elts), C.byref(challenger), abi.as_u32p(out),
                                             C.c_size_t(len(out)), C.byref(plen)))
        return out[:plen.value]


# fine-grained entry points ------------------------------------------------------------------
def poseidon2_permute_batch(ctx: Context, states: np.ndarray) -> np.ndarray:
    s = np.ascontiguousarray(states, dtype=np.uint32).copy()
    lib.check(lib.load().zkm_poseidon2_permute_batch(ctx.h, abi.as_u32p(s), C.c_size_t(s.shape[0])))
    return s


def poseidon2_permute_batch_int(ctx: Context, states: np.ndarray) -> np.ndarray:
    """The integer-pipe formulation of the same permutation (poseidon2.cuh), kept as a parity entry point."""
    s = np.ascontiguousarray(states, dtype=np.uint32).copy()
    lib.check(lib.load().zkm_poseidon2_permute_batch_int(ctx.h, abi.as_u32p(s), C.c_size_t(s.shape[0])))
    return s


def coset_lde_batch(ctx: Context, mat: np.ndarray, log_blowup: int, lde_shift: int) -> np.ndarray:
    m = np.ascontiguousarray(mat, dtype=np.uint32)
    out = np.empty((m.shape[0] << log_blowup, m.shape[1]), dtype=np.uint32)
    lib.check(lib.load().zkm_coset_lde_batch(ctx.h, abi.as_u32p(m), C.c_size_t(m.shape[0]), C.c_size_t(m.shape[1]),
                                             C.c_uint32(log_blowup), C.c_uint32(int(lde_shift)), abi.as_u32p(out)))
    return out
