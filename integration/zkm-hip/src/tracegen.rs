//! `generate_traces` on the device: for every chip the library can build from executor events, the shim hands the
//! shard's events to `zkm_tracegen_*` instead of letting the chip's own `generate_trace` fill a host matrix, and `commit` receives device
//! matrices. A chip that is not listed here (the seven the library does not build: the Weierstrass decompress chips, Uint256Mul,
//! U256x2048Mul, BooleanCircuitGarble, SysLinux) keeps the reference's host path and is uploaded with `zkm_matrix_upload_async`.
//!
//! The event structs with `#[repr(C)]` in the reference (AluEvent, CompAluEvent, BranchEvent, JumpEvent, MovCondEvent, MemInstrEvent, MiscEvent,
//! SyscallEvent, CpuEvent, MemoryLocalEvent, GlobalLookupEvent, MemoryInitializeFinalizeEvent, MemoryReadRecord, MemoryWriteRecord) cross the
//! boundary by pointer cast. Precompile events hold `Vec`s and are flattened into the fixed layouts of `include/zkm_hip.h` first.
//!
//! Byte lookups: the reference's chips add their `ByteLookupEvent`s to the record in `generate_dependencies`, which runs over every chip before
//! proving; the Byte chip's trace therefore stays a host trace built from `record.byte_lookups`, and `commit` passes a null `blu` so that
//! the device generators do not count the same lookups again. (With `generate_dependencies` skipped for the device-built chips, one
//! `ZkmByteLookups` per shard would collect them and `zkm_tracegen_byte_mults` would give the Byte chip's multiplicities — the path the Python
//! mirror and the tests use.)
//!
//! Written against the reference's types; never compiled (no Rust toolchain in the build image) — see INTEGRATION.md.
use std::ptr::null_mut;

use zkm_core_executor::events::{
    EdDecompressEvent, EllipticCurveAddEvent, EllipticCurveDecompressEvent, EllipticCurveDoubleEvent, Fp2AddSubEvent, Fp2MulEvent, FpOpEvent, FieldOperation, KeccakSpongeEvent,
    MemoryReadRecord, MemoryWriteRecord, Poseidon2PermuteEvent, PrecompileEvent, ShaCompressEvent, ShaExtendEvent, SyscallEvent,
};
use zkm_core_executor::{syscalls::SyscallCode, ExecutionRecord};

use crate::{check, ffi, HipMatrix, HipProverError};

/// The chips `device_trace` has an arm for (MachineAir::name). Byte is not among them: its trace is the multiplicities of record.byte_lookups
/// (see `commit`); with device-side counting it would be zkm_tracegen_byte_mults.
pub(crate) const DEVICE_BUILT: &[&str] = &[
    "AddSub", "Bitwise", "Lt", "ShiftLeft", "ShiftRight", "CloClz", "Mul", "DivRem", "Branch", "Jump", "MovCond", "MemoryInstrs", "MiscInstrs", "SyscallInstrs",
    "Global", "MemoryLocal", "Cpu", "Program", "SyscallCore", "SyscallPrecompile", "MemoryGlobalInit", "MemoryGlobalFinalize", "Poseidon2Permute",
    "KeccakSponge", "ShaExtend", "ShaCompress", "EdAddAssign", "EdDecompress", "Secp256k1AddAssign", "Secp256k1DoubleAssign", "Secp256r1AddAssign",
    "Secp256r1DoubleAssign", "Bn254AddAssign", "Bn254DoubleAssign", "Bls12381AddAssign", "Bls12381DoubleAssign", "Bn254FpOpAssign", "Bn254Fp2AddSubAssign",
    "Bn254Fp2MulAssign", "Bls12381FpOpAssign", "Bls12831Fp2AddSubAssign", "Bls12831Fp2MulAssign", "Secp256k1Decompress", "Secp256r1Decompress",
    "Bls12381Decompress", "Uint256MulMod", "U256XU2048Mul", "BooleanCircuitGarble", "SysLinux",
];

/// ZKM_CURVE_* of include/zkm_hip.h
const SECP256K1: i32 = 0;
const SECP256R1: i32 = 1;
const BN254: i32 = 2;
const BLS12381: i32 = 3;

fn rd(r: &MemoryReadRecord) -> ffi::ZkmMemoryReadRecord {
    ffi::ZkmMemoryReadRecord { value: r.value, shard: r.shard, timestamp: r.timestamp, prev_shard: r.prev_shard, prev_timestamp: r.prev_timestamp }
}
fn wr(r: &MemoryWriteRecord) -> ffi::ZkmMemoryWriteRecord {
    ffi::ZkmMemoryWriteRecord {
        value: r.value, shard: r.shard, timestamp: r.timestamp, prev_value: r.prev_value, prev_shard: r.prev_shard, prev_timestamp: r.prev_timestamp,
    }
}
/// A flattened event as the words the C side reads: header words, then write records (6 words each), then read records (5 words each).
fn flat(head: &[u32], writes: &[MemoryWriteRecord], reads: &[MemoryReadRecord]) -> Vec<u32> {
    let mut v = head.to_vec();
    for w in writes { v.extend([w.value, w.shard, w.timestamp, w.prev_value, w.prev_shard, w.prev_timestamp]); }
    for r in reads { v.extend([r.value, r.shard, r.timestamp, r.prev_shard, r.prev_timestamp]); }
    v
}
fn field_op_word(op: FieldOperation) -> u32 {
    match op { FieldOperation::Add => 0, FieldOperation::Mul => 1, FieldOperation::Sub => 2, FieldOperation::Div => 3 }
}

/// One 36-word block of a KeccakSpongeEvent (zkm_keccak_sponge_block): the state after the block is xored in, the block's read records, and —
/// read by the library on the first / last block only — the input-length record and the output write records.
fn keccak_blocks(e: &KeccakSpongeEvent) -> Vec<ffi::ZkmKeccakSpongeBlock> {
    (0..e.num_blocks())
        .map(|k| {
            let mut xored = [0u32; 50];
            for (i, lane) in e.xored_state_list[k].iter().enumerate() {
                xored[2 * i] = *lane as u32;
                xored[2 * i + 1] = (*lane >> 32) as u32;
            }
            ffi::ZkmKeccakSpongeBlock {
                shard: e.shard, clk: e.clk, input_addr: e.input_addr, output_addr: e.output_addr, input_len_u32s: e.input_len_u32s, block_index: k as u32,
                xored_state: xored,
                input_read_records: core::array::from_fn(|j| rd(&e.input_read_records[36 * k + j])),
                input_length_record: rd(&e.input_length_record),
                output_write_records: core::array::from_fn(|j| wr(&e.output_write_records[j])),
            }
        })
        .collect()
}

/// The events of one precompile syscall code, in the order the reference's chip walks them (`get_precompile_events`).
fn precompile<'a>(r: &'a ExecutionRecord, code: SyscallCode) -> impl Iterator<Item = &'a PrecompileEvent> {
    r.precompile_events.get_events(code).into_iter().flatten().map(|(_, e)| e)
}

/// Build `chip`'s main trace on the device from the shard's events; `None` for a chip the library has no generator for.
pub(crate) fn device_trace(
    ctx: *mut ffi::ZkmCtx,
    chip: &str,
    r: &ExecutionRecord,
    fixed: i32,
    blu: *mut ffi::ZkmByteLookups,
) -> Result<Option<HipMatrix>, HipProverError> {
    let mut m: *mut ffi::ZkmMatrix = null_mut();
    macro_rules! repr_c { ($f:ident, $events:expr, $t:ty $(, $extra:expr)*) => {{
        let ev = $events;
        check(unsafe { ffi::$f(ctx, $($extra,)* ev.as_ptr() as *const $t, ev.len(), fixed, blu, &mut m) })?
    }}; }
    macro_rules! repr_c_no_blu { ($f:ident, $events:expr, $t:ty) => {{
        let ev = $events;
        check(unsafe { ffi::$f(ctx, ev.as_ptr() as *const $t, ev.len(), fixed, &mut m) })?
    }}; }
    macro_rules! curve_events { ($code:ident, $variant:ident, $f:ident, $curve:expr, $flatten:expr) => {{
        let words: Vec<u32> = precompile(r, SyscallCode::$code).flat_map(|e| match e { PrecompileEvent::$variant(e) => $flatten(e), _ => unreachable!() }).collect();
        let n = precompile(r, SyscallCode::$code).count();
        check(unsafe { ffi::$f(ctx, $curve, words.as_ptr() as *const _, n, fixed, blu, &mut m) })?
    }}; }
    let add = |e: &EllipticCurveAddEvent| flat(&[e.shard, e.clk, e.p_ptr, e.q_ptr], &e.p_memory_records, &e.q_memory_records);
    let dbl = |e: &EllipticCurveDoubleEvent| flat(&[e.shard, e.clk, e.p_ptr], &e.p_memory_records, &[]);
    // a decompression lists its reads (x) before its writes (y): include/zkm_hip.h, zkm_tracegen_weierstrass_decompress
    let dec = |e: &EllipticCurveDecompressEvent| {
        let mut v = vec![e.shard, e.clk, e.ptr, e.sign_bit as u32];
        v.extend(flat(&[], &[], &e.x_memory_records));
        v.extend(flat(&[], &e.y_memory_records, &[]));
        v
    };
    let fp = |e: &FpOpEvent| flat(&[e.shard, e.clk, e.x_ptr, e.y_ptr, field_op_word(e.op)], &e.x_memory_records, &e.y_memory_records);
    let fp2 = |e: &Fp2AddSubEvent| flat(&[e.shard, e.clk, e.x_ptr, e.y_ptr, field_op_word(e.op)], &e.x_memory_records, &e.y_memory_records);
    let fp2m = |e: &Fp2MulEvent| flat(&[e.shard, e.clk, e.x_ptr, e.y_ptr], &e.x_memory_records, &e.y_memory_records);
    match chip {
        // ---- the chips whose events are #[repr(C)] (crates/core/machine/src/sys.rs:22-72 is the reference's own list of them)
        "AddSub" => repr_c!(zkm_tracegen_alu, &r.add_sub_events, ffi::ZkmAluEvent, 0),
        "Bitwise" => repr_c!(zkm_tracegen_alu, &r.bitwise_events, ffi::ZkmAluEvent, 1),
        "Lt" => repr_c!(zkm_tracegen_alu, &r.lt_events, ffi::ZkmAluEvent, 2),
        "ShiftLeft" => repr_c!(zkm_tracegen_alu, &r.shift_left_events, ffi::ZkmAluEvent, 3),
        "ShiftRight" => repr_c!(zkm_tracegen_alu, &r.shift_right_events, ffi::ZkmAluEvent, 4),
        "CloClz" => repr_c!(zkm_tracegen_alu, &r.cloclz_events, ffi::ZkmAluEvent, 5),
        "Mul" => repr_c!(zkm_tracegen_mul, &r.mul_events, ffi::ZkmCompAluEvent),
        "DivRem" => repr_c!(zkm_tracegen_divrem, &r.divrem_events, ffi::ZkmCompAluEvent),
        "Branch" => repr_c!(zkm_tracegen_branch, &r.branch_events, ffi::ZkmBranchEvent),
        "Jump" => repr_c_no_blu!(zkm_tracegen_jump, &r.jump_events, ffi::ZkmJumpEvent),
        "MovCond" => repr_c_no_blu!(zkm_tracegen_mov_cond, &r.movcond_events, ffi::ZkmMovCondEvent),
        "MemoryInstrs" => repr_c!(zkm_tracegen_memory_instrs, &r.memory_instr_events, ffi::ZkmMemInstrEvent),
        "MiscInstrs" => repr_c!(zkm_tracegen_misc_instrs, &r.misc_events, ffi::ZkmMiscEvent),
        "SyscallInstrs" => repr_c_no_blu!(zkm_tracegen_syscall_instrs, &r.syscall_events, ffi::ZkmSyscallEvent),
        "Global" => repr_c!(zkm_tracegen_global, &r.global_lookup_events, ffi::ZkmGlobalLookupEvent),
        "MemoryLocal" => {
            let ev: Vec<_> = r.get_local_mem_events().cloned().collect();
            check(unsafe { ffi::zkm_tracegen_memory_local(ctx, ev.as_ptr() as *const ffi::ZkmMemoryLocalEvent, ev.len(), fixed, &mut m) })?
        }
        "Cpu" => check(unsafe {
            ffi::zkm_tracegen_cpu(ctx, r.cpu_events.as_ptr() as *const ffi::ZkmCpuEvent, r.cpu_events.len(), r.program.instructions.as_ptr() as *const ffi::ZkmInstruction,
                                  r.program.instructions.len(), r.program.pc_base, r.public_values.shard, fixed, blu, &mut m)
        })?,
        "Program" => check(unsafe {     // the main trace of the Program chip is its multiplicity column; its preprocessed trace is zkm_tracegen_program at setup
            ffi::zkm_tracegen_program_mults(ctx, r.cpu_events.as_ptr() as *const ffi::ZkmCpuEvent, r.cpu_events.len(), r.program.instructions.len(), r.program.pc_base, fixed, &mut m)
        })?,
        "SyscallCore" => check(unsafe {
            ffi::zkm_tracegen_syscall(ctx, r.syscall_events.as_ptr() as *const ffi::ZkmSyscallEvent, r.syscall_events.len(), 0, fixed, blu, &mut m)      // filtered inside, as the chip filters
        })?,
        "SyscallPrecompile" => {
            // a Linux call's code and result reach the table through its syscall event's a_record (include/zkm_hip.h, zkm_tracegen_syscall):
            // the reference reads them off the LinuxEvent (syscall/chip.rs:223-238)
            let ev: Vec<SyscallEvent> = r.precompile_events.all_events().map(|(e, p)| {
                let mut e = *e;
                if let PrecompileEvent::Linux(l) = p { e.a_record.prev_value = l.syscall_code; e.a_record.value = l.v0; }
                e
            }).collect();
            check(unsafe { ffi::zkm_tracegen_syscall(ctx, ev.as_ptr() as *const ffi::ZkmSyscallEvent, ev.len(), 1, fixed, blu, &mut m) })?
        }
        "MemoryGlobalInit" | "MemoryGlobalFinalize" => {
            let (ev, bits) = if chip == "MemoryGlobalInit" {
                (&r.global_memory_initialize_events, &r.public_values.previous_init_addr_bits)
            } else {
                (&r.global_memory_finalize_events, &r.public_values.previous_finalize_addr_bits)
            };
            let previous = bits.iter().enumerate().fold(0u32, |acc, (i, b)| acc | (*b << i));
            check(unsafe { ffi::zkm_tracegen_memory_global(ctx, ev.as_ptr() as *const ffi::ZkmMemoryInitFinalizeEvent, ev.len(), previous, fixed, &mut m) })?
        }
        // ---- precompiles: events flattened into the layouts of include/zkm_hip.h
        "Poseidon2Permute" => {
            let ev: Vec<ffi::ZkmPoseidon2PermuteEvent> = precompile(r, SyscallCode::POSEIDON2_PERMUTE)
                .map(|e| match e {
                    PrecompileEvent::Poseidon2Permute(e) => poseidon2_event(e),
                    _ => unreachable!(),
                })
                .collect();
            check(unsafe { ffi::zkm_tracegen_poseidon2_permute(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "KeccakSponge" => {
            let blocks: Vec<ffi::ZkmKeccakSpongeBlock> = precompile(r, SyscallCode::KECCAK_SPONGE)
                .flat_map(|e| match e { PrecompileEvent::KeccakSponge(e) => keccak_blocks(e), _ => unreachable!() })
                .collect();
            check(unsafe { ffi::zkm_tracegen_keccak_sponge(ctx, blocks.as_ptr(), blocks.len(), fixed, blu, &mut m) })?     // 24 rows per block
        }
        "ShaExtend" => {
            let ev: Vec<ffi::ZkmShaExtendEvent> = precompile(r, SyscallCode::SHA_EXTEND)
                .map(|e| match e { PrecompileEvent::ShaExtend(e) => sha_extend_event(e), _ => unreachable!() })
                .collect();
            check(unsafe { ffi::zkm_tracegen_sha_extend(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "ShaCompress" => {
            let ev: Vec<ffi::ZkmShaCompressEvent> = precompile(r, SyscallCode::SHA_COMPRESS)
                .map(|e| match e { PrecompileEvent::ShaCompress(e) => sha_compress_event(e), _ => unreachable!() })
                .collect();
            check(unsafe { ffi::zkm_tracegen_sha_compress(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "EdAddAssign" => {
            let ev: Vec<ffi::ZkmEdAddEvent> = precompile(r, SyscallCode::ED_ADD)
                .map(|e| match e {
                    PrecompileEvent::EdAdd(e) => ffi::ZkmEdAddEvent {
                        shard: e.shard, clk: e.clk, p_ptr: e.p_ptr, q_ptr: e.q_ptr,
                        p_memory_records: core::array::from_fn(|i| wr(&e.p_memory_records[i])),
                        q_memory_records: core::array::from_fn(|i| rd(&e.q_memory_records[i])),
                    },
                    _ => unreachable!(),
                })
                .collect();
            check(unsafe { ffi::zkm_tracegen_ed_add(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "EdDecompress" => {
            let ev: Vec<ffi::ZkmEdDecompressEvent> = precompile(r, SyscallCode::ED_DECOMPRESS)
                .map(|e| match e { PrecompileEvent::EdDecompress(e) => ed_decompress_event(e), _ => unreachable!() })
                .collect();
            check(unsafe { ffi::zkm_tracegen_ed_decompress(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "Secp256k1AddAssign" => curve_events!(SECP256K1_ADD, Secp256k1Add, zkm_tracegen_weierstrass_add, SECP256K1, add),
        "Secp256k1DoubleAssign" => curve_events!(SECP256K1_DOUBLE, Secp256k1Double, zkm_tracegen_weierstrass_double, SECP256K1, dbl),
        "Secp256r1AddAssign" => curve_events!(SECP256R1_ADD, Secp256r1Add, zkm_tracegen_weierstrass_add, SECP256R1, add),
        "Secp256r1DoubleAssign" => curve_events!(SECP256R1_DOUBLE, Secp256r1Double, zkm_tracegen_weierstrass_double, SECP256R1, dbl),
        "Bn254AddAssign" => curve_events!(BN254_ADD, Bn254Add, zkm_tracegen_weierstrass_add, BN254, add),
        "Bn254DoubleAssign" => curve_events!(BN254_DOUBLE, Bn254Double, zkm_tracegen_weierstrass_double, BN254, dbl),
        "Bls12381AddAssign" => curve_events!(BLS12381_ADD, Bls12381Add, zkm_tracegen_weierstrass_add, BLS12381, add),
        "Bls12381DoubleAssign" => curve_events!(BLS12381_DOUBLE, Bls12381Double, zkm_tracegen_weierstrass_double, BLS12381, dbl),
        "Uint256MulMod" => {
            let ev: Vec<ffi::ZkmUint256MulEvent> = precompile(r, SyscallCode::UINT256_MUL)
                .map(|e| match e {
                    PrecompileEvent::Uint256Mul(e) => ffi::ZkmUint256MulEvent {
                        shard: e.shard, clk: e.clk, x_ptr: e.x_ptr, y_ptr: e.y_ptr,
                        x_memory_records: core::array::from_fn(|i| wr(&e.x_memory_records[i])),
                        y_memory_records: core::array::from_fn(|i| rd(&e.y_memory_records[i])),
                        modulus_memory_records: core::array::from_fn(|i| rd(&e.modulus_memory_records[i])),
                    },
                    _ => unreachable!(),
                })
                .collect();
            check(unsafe { ffi::zkm_tracegen_uint256_mul(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "U256XU2048Mul" => {
            let ev: Vec<ffi::ZkmU256x2048MulEvent> = precompile(r, SyscallCode::U256XU2048_MUL)
                .map(|e| match e {
                    PrecompileEvent::U256xU2048Mul(e) => ffi::ZkmU256x2048MulEvent {
                        shard: e.shard, clk: e.clk, a_ptr: e.a_ptr, b_ptr: e.b_ptr, lo_ptr: e.lo_ptr, hi_ptr: e.hi_ptr,
                        lo_ptr_memory: rd(&e.lo_ptr_memory), hi_ptr_memory: rd(&e.hi_ptr_memory),
                        a_memory_records: core::array::from_fn(|i| rd(&e.a_memory_records[i])),
                        b_memory_records: core::array::from_fn(|i| rd(&e.b_memory_records[i])),
                        lo_memory_records: core::array::from_fn(|i| wr(&e.lo_memory_records[i])),
                        hi_memory_records: core::array::from_fn(|i| wr(&e.hi_memory_records[i])),
                    },
                    _ => unreachable!(),
                })
                .collect();
            check(unsafe { ffi::zkm_tracegen_u256x2048_mul(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "BooleanCircuitGarble" => {
            // one record per row: a header row and a row per gate (include/zkm_hip.h, zkm_garble_row)
            let mut rows: Vec<ffi::ZkmGarbleRow> = Vec::new();
            let none = ffi::ZkmMemoryReadRecord { value: 0, shard: 0, timestamp: 0, prev_shard: 0, prev_timestamp: 0 };
            let no_write = ffi::ZkmMemoryWriteRecord { value: 0, shard: 0, timestamp: 0, prev_value: 0, prev_shard: 0, prev_timestamp: 0 };
            for e in precompile(r, SyscallCode::BOOLEAN_CIRCUIT_GARBLE) {
                let e = match e { PrecompileEvent::BooleanCircuitGarble(e) => e, _ => unreachable!() };
                let n = e.num_gates();
                let mut reads = [none; 17];
                reads[0] = rd(&e.num_gates_read_record);
                for i in 0..4 { reads[1 + i] = rd(&e.delta_read_records[i]); }
                rows.push(ffi::ZkmGarbleRow { shard: e.shard, clk: e.clk, input_address: e.input_addr, output_address: e.output_addr, is_gate: 0, gate_id: 0,
                                              gates_num: n as u32, pre_check: 0, delta: e.delta, reads, write: no_write });
                let mut running = true;
                for g in 0..n {
                    let info = &e.gates_info[17 * g..17 * g + 17];
                    let ok = (0..4).all(|i| info[1 + i] ^ info[5 + i] ^ info[9 + i] ^ (if info[0] != 0 { e.delta[i] } else { 0 }) == info[13 + i]);
                    rows.push(ffi::ZkmGarbleRow {
                        shard: e.shard, clk: e.clk, input_address: e.input_addr + 20 + 68 * g as u32, output_address: e.output_addr, is_gate: 1, gate_id: g as u32,
                        gates_num: n as u32, pre_check: running as u32, delta: e.delta,
                        reads: core::array::from_fn(|i| rd(&e.gates_read_records[17 * g + i])),
                        write: if g + 1 == n { wr(&e.output_write_record) } else { no_write },
                    });
                    running = running && ok;
                }
            }
            check(unsafe { ffi::zkm_tracegen_boolean_circuit_garble(ctx, rows.as_ptr(), rows.len(), fixed, blu, &mut m) })?
        }
        "SysLinux" => {
            let none = ffi::ZkmMemoryReadRecord { value: 0, shard: 0, timestamp: 0, prev_shard: 0, prev_timestamp: 0 };
            let no_write = ffi::ZkmMemoryWriteRecord { value: 0, shard: 0, timestamp: 0, prev_value: 0, prev_shard: 0, prev_timestamp: 0 };
            let ev: Vec<ffi::ZkmLinuxEvent> = precompile(r, SyscallCode::SYS_LINUX)
                .map(|e| match e {
                    PrecompileEvent::Linux(e) => ffi::ZkmLinuxEvent {
                        shard: e.shard, clk: e.clk, a0: e.a0, a1: e.a1, v0: e.v0, syscall_code: e.syscall_code,
                        read_record: e.read_records.first().map(rd).unwrap_or(none),
                        a3_record: wr(&e.write_records[0]),
                        heap_record: e.write_records.get(1).map(wr).unwrap_or(no_write),
                    },
                    _ => unreachable!(),
                })
                .collect();
            check(unsafe { ffi::zkm_tracegen_sys_linux(ctx, ev.as_ptr(), ev.len(), fixed, blu, &mut m) })?
        }
        "Secp256k1Decompress" => curve_events!(SECP256K1_DECOMPRESS, Secp256k1Decompress, zkm_tracegen_weierstrass_decompress, SECP256K1, dec),
        "Secp256r1Decompress" => curve_events!(SECP256R1_DECOMPRESS, Secp256r1Decompress, zkm_tracegen_weierstrass_decompress, SECP256R1, dec),
        "Bls12381Decompress" => curve_events!(BLS12381_DECOMPRESS, Bls12381Decompress, zkm_tracegen_weierstrass_decompress, BLS12381, dec),
        // the three Fp codes of a field are filed under FP_ADD, FP2_ADD and FP2_SUB under FP2_ADD (syscalls/precompiles/fptower/)
        "Bn254FpOpAssign" => curve_events!(BN254_FP_ADD, Bn254Fp, zkm_tracegen_fp_op, BN254, fp),
        "Bn254Fp2AddSubAssign" => curve_events!(BN254_FP2_ADD, Bn254Fp2AddSub, zkm_tracegen_fp2_addsub, BN254, fp2),
        "Bn254Fp2MulAssign" => curve_events!(BN254_FP2_MUL, Bn254Fp2Mul, zkm_tracegen_fp2_mul, BN254, fp2m),
        "Bls12381FpOpAssign" => curve_events!(BLS12381_FP_ADD, Bls12381Fp, zkm_tracegen_fp_op, BLS12381, fp),
        "Bls12831Fp2AddSubAssign" => curve_events!(BLS12381_FP2_ADD, Bls12381Fp2AddSub, zkm_tracegen_fp2_addsub, BLS12381, fp2),      // sic: the reference's chip name
        "Bls12831Fp2MulAssign" => curve_events!(BLS12381_FP2_MUL, Bls12381Fp2Mul, zkm_tracegen_fp2_mul, BLS12381, fp2m),
        _ => return Ok(None),
    };
    let (height, width) = unsafe { (ffi::zkm_matrix_height(m), ffi::zkm_matrix_width(m)) };
    Ok(Some(HipMatrix::from_handle(ctx, m, height, width)))
}

fn poseidon2_event(e: &Poseidon2PermuteEvent) -> ffi::ZkmPoseidon2PermuteEvent {
    // pre_state / post_state are the previous values / values of the sixteen write records (syscalls/precompiles/poseidon2/permute.rs:30-47)
    ffi::ZkmPoseidon2PermuteEvent { shard: e.shard, clk: e.clk, state_addr: e.state_addr, state_records: core::array::from_fn(|i| wr(&e.state_records[i])) }
}
fn sha_extend_event(e: &ShaExtendEvent) -> ffi::ZkmShaExtendEvent {
    ffi::ZkmShaExtendEvent {
        shard: e.shard, clk: e.clk, w_ptr: e.w_ptr,
        w_i_minus_15_reads: core::array::from_fn(|j| rd(&e.w_i_minus_15_reads[j])),
        w_i_minus_2_reads: core::array::from_fn(|j| rd(&e.w_i_minus_2_reads[j])),
        w_i_minus_16_reads: core::array::from_fn(|j| rd(&e.w_i_minus_16_reads[j])),
        w_i_minus_7_reads: core::array::from_fn(|j| rd(&e.w_i_minus_7_reads[j])),
        w_i_writes: core::array::from_fn(|j| wr(&e.w_i_writes[j])),
    }
}
fn sha_compress_event(e: &ShaCompressEvent) -> ffi::ZkmShaCompressEvent {
    // `w` and `h` of the reference's event are the values of the read records
    ffi::ZkmShaCompressEvent {
        shard: e.shard, clk: e.clk, w_ptr: e.w_ptr, h_ptr: e.h_ptr,
        h_read_records: core::array::from_fn(|i| rd(&e.h_read_records[i])),
        w_i_read_records: core::array::from_fn(|i| rd(&e.w_i_read_records[i])),
        h_write_records: core::array::from_fn(|i| wr(&e.h_write_records[i])),
    }
}
fn ed_decompress_event(e: &EdDecompressEvent) -> ffi::ZkmEdDecompressEvent {
    // y_bytes / decompressed_x_bytes are the values of the y read records / x write records
    ffi::ZkmEdDecompressEvent {
        shard: e.shard, clk: e.clk, ptr: e.ptr, sign: e.sign as u32,
        x_memory_records: core::array::from_fn(|i| wr(&e.x_memory_records[i])),
        y_memory_records: core::array::from_fn(|i| rd(&e.y_memory_records[i])),
    }
}
