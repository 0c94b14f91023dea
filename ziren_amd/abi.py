"""ctypes declarations of include/zkm_hip.h (structs and prototypes)."""
import ctypes as C

import numpy as np

u32p = C.POINTER(C.c_uint32)


class FriConfig(C.Structure):
    """FriConfig, crates/stark/src/kb31_poseidon2.rs:203-213 (core: blowup 2, 84 queries, 16 PoW bits)."""
    _fields_ = [("log_blowup", C.c_uint32), ("num_queries", C.c_uint32), ("proof_of_work_bits", C.c_uint32)]


class Challenger(C.Structure):
    _fields_ = [("sponge_state", C.c_uint32 * 16), ("num_inputs", C.c_uint32),
                ("input_buffer", C.c_uint32 * 16), ("num_outputs", C.c_uint32),
                ("output_buffer", C.c_uint32 * 16)]

    def copy(self):
        c = Challenger()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Challenger))
        return c

    def as_tuple(self):
        return (tuple(self.sponge_state), self.num_inputs, tuple(self.input_buffer[:self.num_inputs]),
                self.num_outputs, tuple(self.output_buffer[:self.num_outputs]))


class ChipDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("main_width", C.c_uint32), ("prep_width", C.c_uint32),
                ("prep_index", C.c_int32), ("log_quotient_degree", C.c_uint32), ("local_only", C.c_uint32),
                ("commit_scope_global", C.c_uint32), ("num_constraints", C.c_uint32),
                ("lookups", u32p), ("lookups_len", C.c_uint32),
                ("program", u32p), ("program_len", C.c_uint32)]


# enum zkm_tracegen_kind
(TG_ALU, TG_CPU, TG_BRANCH, TG_JUMP, TG_MOV_COND, TG_MUL, TG_DIVREM, TG_MEMORY_INSTRS, TG_MISC_INSTRS, TG_SYSCALL_INSTRS, TG_SYSCALL_CORE,
 TG_SYSCALL_PRECOMPILE, TG_MEMORY_LOCAL, TG_GLOBAL, TG_BYTE_MULTS, TG_PROGRAM_MULTS, TG_FLAT, TG_POSEIDON2_WIDE, TG_EXP_REVERSE_BITS) = range(19)


class TracegenDesc(C.Structure):
    """zkm_tracegen_desc: one chip of a zkm_tracegen_shard call."""
    _fields_ = [("kind", C.c_uint32), ("chip", C.c_int32), ("events", C.c_void_p), ("n_events", C.c_size_t), ("fixed_log2_rows", C.c_int32),
                ("no_byte_lookups", C.c_uint32), ("program", C.c_void_p), ("n_instr", C.c_size_t), ("pc_base", C.c_uint32), ("shard", C.c_uint32)]


def as_u32p(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def make_chip_descs(chips):
    """ChipDesc array for a list of synth.SynChip-like objects. Returns (array, keepalive)."""
    arr = (ChipDesc * len(chips))()
    keep = []
    for i, c in enumerate(chips):
        name = c.name.encode()
        lk = np.ascontiguousarray(c.lookups_blob, dtype=np.uint32)
        pg = np.ascontiguousarray(c.program, dtype=np.uint32)
        keep += [name, lk, pg]
        arr[i].name = name
        arr[i].main_width = c.main_width
        arr[i].prep_width = c.prep_width
        arr[i].prep_index = c.prep_index
        arr[i].log_quotient_degree = c.log_quotient_degree
        arr[i].local_only = int(c.local_only)
        arr[i].commit_scope_global = int(c.commit_scope_global)
        arr[i].num_constraints = c.num_constraints
        arr[i].lookups = as_u32p(lk)
        arr[i].lookups_len = len(lk)
        arr[i].program = as_u32p(pg)
        arr[i].program_len = len(pg)
    return arr, keep
