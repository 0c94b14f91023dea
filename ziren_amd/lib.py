"""ctypes loader for libzkm_hip.so (the C ABI of include/zkm_hip.h)."""
import ctypes as C
import importlib.util
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
# ZKM_HIP_LIB: another build of the same library (A/B timing of two kernel versions on one box); never anything but a libzkm_hip build
LIB_PATH = os.environ.get("ZKM_HIP_LIB") or os.path.join(HERE, "libzkm_hip.so")
_LIB = None

# every symbol include/zkm_hip.h declares
EXPORTS = [
    "zkm_last_error", "zkm_build_info", "zkm_ctx_create", "zkm_ctx_destroy", "zkm_ctx_synchronize", "zkm_ctx_trim", "zkm_ctx_set_memory_limit", "zkm_ctx_memory_held", "zkm_ctx_last_timings", "zkm_ctx_kernel_timings", "zkm_ctx_set_kernel_timing", "zkm_ctx_set_kernel_timing_only", "zkm_ctx_set_lde_overlap", "zkm_ctx_set_rows_up_front", "zkm_ctx_set_host_wait", "zkm_ctx_register_quotient_kernel", "zkm_ctx_register_perm_kernel",
    "zkm_host_alloc", "zkm_host_free", "zkm_matrix_upload", "zkm_matrix_upload_async", "zkm_matrix_wait", "zkm_events_upload_async", "zkm_events_free", "zkm_matrix_download", "zkm_matrix_height", "zkm_matrix_width", "zkm_matrix_free",
    "zkm_pcs_commit", "zkm_pcs_data_free", "zkm_pcs_data_get_lde", "zkm_pcs_open_batch",
    "zkm_pk_setup", "zkm_pk_commitment", "zkm_pk_observe_into", "zkm_pk_free",
    "zkm_commit", "zkm_main_data_free", "zkm_open", "zkm_prove_shard",
    "zkm_tracegen_alu_width", "zkm_byte_lookups_create", "zkm_byte_lookups_free", "zkm_tracegen_alu", "zkm_tracegen_jump_width", "zkm_tracegen_jump", "zkm_tracegen_branch_width", "zkm_tracegen_branch", "zkm_tracegen_mov_cond_width", "zkm_tracegen_mov_cond", "zkm_tracegen_mul_width", "zkm_tracegen_mul", "zkm_tracegen_divrem_width", "zkm_tracegen_divrem", "zkm_tracegen_cpu_width", "zkm_tracegen_cpu", "zkm_tracegen_cpu_and_program", "zkm_tracegen_program", "zkm_tracegen_program_mults", "zkm_tracegen_memory_local", "zkm_tracegen_global", "zkm_tracegen_misc_instrs_width", "zkm_tracegen_misc_instrs", "zkm_tracegen_syscall_instrs_width", "zkm_tracegen_syscall_instrs", "zkm_tracegen_syscall", "zkm_tracegen_memory_global", "zkm_tracegen_poseidon2_permute", "zkm_tracegen_keccak_sponge", "zkm_tracegen_sha_extend", "zkm_tracegen_sha_compress", "zkm_tracegen_ed_add", "zkm_tracegen_ed_decompress", "zkm_tracegen_weierstrass_add", "zkm_tracegen_weierstrass_double", "zkm_tracegen_weierstrass_decompress", "zkm_tracegen_uint256_mul", "zkm_tracegen_u256x2048_mul", "zkm_tracegen_boolean_circuit_garble", "zkm_tracegen_sys_linux", "zkm_tracegen_fp_op", "zkm_tracegen_fp2_addsub", "zkm_tracegen_fp2_mul", "zkm_tracegen_poseidon2_wide", "zkm_tracegen_poseidon2_skinny", "zkm_tracegen_exp_reverse_bits", "zkm_tracegen_memory_instrs_width", "zkm_tracegen_memory_instrs", "zkm_tracegen_flat", "zkm_tracegen_byte_table", "zkm_tracegen_byte_mults", "zkm_tracegen_shard",
    "zkm_poseidon2_permute_batch", "zkm_poseidon2_permute_batch_int", "zkm_coset_lde_batch", "zkm_permutation_trace",
    "zkm_challenger_init", "zkm_challenger_observe", "zkm_challenger_sample", "zkm_challenger_sample_bits",
    "zkm_host_poseidon2_permute", "zkm_host_poseidon2_permute_f64", "zkm_host_poseidon2_f64_sponge", "zkm_host_poseidon2_f64_compress_inject", "zkm_host_poseidon2_f64_audit",
    "zkm_host_ext_mul", "zkm_host_ext_inv", "zkm_host_field_mul", "zkm_host_field_inv", "zkm_host_reduce96_bounded", "zkm_host_two_adic_generator",
]


class ZkmError(RuntimeError):
    pass


def _preload_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as the
    system one); if our library pulled in /opt/rocm's copy first, a later `import torch` (bench.py uses
    torch.distributed for N > 1) would load a second runtime and find no GPU. Loading torch's copy first,
    when torch is installed, makes both bind to the same runtime regardless of import order."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the in-tree HIP library. Fails loudly if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.environ.get("ZKM_HIP_LIB"):
        # missing, or built from other sources than the tree's (digest recorded in the binary, ziren_amd/build.py): compile it now if the
        # ROCm toolchain is here; never substitute anything else for it and never run a stale one
        from . import build as _build
        if _build.needs_build():
            try:
                _build.build(verbose=True)
            except Exception as e:  # noqa: BLE001
                raise ZkmError(f"{LIB_PATH} is missing or stale (its digest {_build.recorded_digest()}, the tree's {_build.sources_digest()}) "
                               f"and could not be built ({e}): run `python -m ziren_amd.build`") from e
    elif not os.path.exists(LIB_PATH):
        raise ZkmError(f"ZKM_HIP_LIB={LIB_PATH} does not exist")
    # a proof is ~525 dependent launches; with the kernel arguments in device memory each dispatch is ~1 microsecond shorter (0.6 ms per
    # SYN-22 proof, measured: DESIGN.md section 4). The runtime reads this when it initialises, so it has to be in place before the first HIP call
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.zkm_last_error.restype = C.c_char_p
    if hasattr(L, "zkm_build_info"):
        L.zkm_build_info.restype = C.c_char_p
    L.zkm_host_alloc.restype = C.c_void_p
    L.zkm_matrix_height.restype = C.c_size_t
    if hasattr(L, "zkm_ctx_memory_held"):
        L.zkm_ctx_memory_held.restype = C.c_size_t
        L.zkm_ctx_set_memory_limit.argtypes = [C.c_void_p, C.c_size_t]
    L.zkm_matrix_width.restype = C.c_size_t
    L.zkm_tracegen_alu_width.restype = C.c_size_t
    L.zkm_tracegen_jump_width.restype = C.c_size_t
    L.zkm_tracegen_mov_cond_width.restype = C.c_size_t
    L.zkm_tracegen_branch_width.restype = C.c_size_t
    L.zkm_tracegen_mul_width.restype = C.c_size_t
    L.zkm_tracegen_divrem_width.restype = C.c_size_t
    L.zkm_tracegen_cpu_width.restype = C.c_size_t
    L.zkm_tracegen_misc_instrs_width.restype = C.c_size_t
    L.zkm_tracegen_syscall_instrs_width.restype = C.c_size_t
    L.zkm_tracegen_memory_instrs_width.restype = C.c_size_t
    L.zkm_challenger_sample.restype = C.c_uint32
    L.zkm_challenger_sample_bits.restype = C.c_uint32
    L.zkm_host_field_mul.restype = C.c_uint32
    L.zkm_host_field_inv.restype = C.c_uint32
    if hasattr(L, "zkm_host_reduce96_bounded"):
        L.zkm_host_reduce96_bounded.restype = C.c_uint32
        L.zkm_host_reduce96_bounded.argtypes = [C.c_uint32, C.c_uint64]
    L.zkm_host_two_adic_generator.restype = C.c_uint32
    for name in ("zkm_ctx_destroy", "zkm_host_free", "zkm_events_free", "zkm_byte_lookups_free", "zkm_ctx_set_kernel_timing", "zkm_ctx_set_kernel_timing_only", "zkm_ctx_set_lde_overlap", "zkm_ctx_set_rows_up_front", "zkm_ctx_set_host_wait", "zkm_matrix_free", "zkm_pcs_data_free", "zkm_pk_free", "zkm_main_data_free",
                 "zkm_challenger_init", "zkm_challenger_observe", "zkm_host_poseidon2_permute", "zkm_host_poseidon2_permute_f64", "zkm_host_poseidon2_f64_sponge",
                 "zkm_host_poseidon2_f64_compress_inject", "zkm_host_poseidon2_f64_audit", "zkm_host_ext_mul", "zkm_host_ext_inv"):
        if hasattr(L, name) or "ZKM_HIP_LIB" not in os.environ:   # an older build under A/B comparison may lack the newest entry points
            getattr(L, name).restype = None
    _LIB = L
    return L


def build_info():
    """{"ZKM_SOURCES_DIGEST": ..., "hipcc": ..., "arch": ...} as compiled into the loaded library (zkm_build_info)."""
    L = load()
    if not hasattr(L, "zkm_build_info"):
        return {}
    return dict(kv.split("=", 1) for kv in L.zkm_build_info().decode().split(";") if "=" in kv)


def check_build_identity():
    """The loaded library must have been built from this tree's sources (digest compiled into it against ziren_amd/build.py's digest of
    csrc/ + include/zkm_hip.h): a stale binary is an error, not a silent run. Returns the digest. Another build under A/B comparison
    (ZKM_HIP_LIB) is what the caller asked for and is not checked."""
    from . import build as _build
    have = build_info().get("ZKM_SOURCES_DIGEST")
    if os.environ.get("ZKM_HIP_LIB"):
        return have
    want = _build.sources_digest()
    if have != want:
        raise ZkmError(f"{LIB_PATH} was built from other sources (its digest {have}, the tree's {want}): run `python -m ziren_amd.build`")
    return have


def check(rc):
    if rc != 0:
        raise ZkmError(load().zkm_last_error().decode())
