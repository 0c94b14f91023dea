/*
 * zkm_hip.h — C ABI of libzkm_hip.so, the MI355X (gfx950) shard-prover back-end.
 *
 * This is the drop-in boundary for Ziren's per-shard STARK proving path
 *     MachineProver::commit + MachineProver::open      crates/stark/src/prover.rs:30-184
 *     (CPU impl: commit :258-292, open :298-653)
 * A Rust shim `HipProver<SC, A>: MachineProver<SC, A>` binds these entry
 * points 1:1 to the trait's associated types and methods:
 *
 *     trait item (prover.rs)                    C ABI
 *     ----------------------------------------  ---------------------------------------------
 *     type DeviceMatrix           (:33-36)      zkm_matrix        (column-major, HBM resident)
 *     type DeviceProverData       (:39)         zkm_pcs_data      (LDEs + Merkle digest layers)
 *     type DeviceProvingKey       (:42)         zkm_pk
 *     fn new(machine)             (:48)         zkm_ctx_create
 *     fn setup / pk_to_device     (:54-66)      zkm_pk_setup      (pcs.commit of machine.rs:406-417)
 *     pk.observe_into             machine.rs:79-86   zkm_pk_observe_into
 *     fn commit(record, traces)   (:111-115)    zkm_commit
 *     fn open(pk, data, chal)     (:118-126)    zkm_open
 *     fn prove  (one shard)       (:660-693)    zkm_prove_shard
 *     type Error                  (:45)         int status + zkm_last_error()
 *
 * Data representation: every field element crossing this ABI is a KoalaBear
 * element as Plonky3 holds it in memory — a u32 in Montgomery form, R = 2^32, value < p =
 * 0x7f000001 (crates/core/machine/include/kb31_t.hpp:458-503). Extension elements are 4 such
 * words, low coefficient first (crates/stark/src/air/extension.rs:55-74). Host matrices are
 * row-major (`RowMajorMatrix<KoalaBear>`, prover.rs:225). Digests are 8 words.
 *
 * Threading: a zkm_ctx is bound to one GPU and serialises calls internally; use one context
 * per GPU (prover.rs:30 requires Send + Sync). No function unwinds or calls back.
 * All functions return 0 on success, non-zero on failure (message via zkm_last_error()).
 */
#ifndef ZKM_HIP_H
#define ZKM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKM_DIGEST_ELEMS 8
#define ZKM_EXT_DEGREE 4
#define ZKM_PERM_WIDTH 16
#define ZKM_KOALABEAR_P 0x7f000001u

typedef struct zkm_ctx zkm_ctx;
typedef struct zkm_matrix zkm_matrix;
typedef struct zkm_pcs_data zkm_pcs_data;
typedef struct zkm_pk zkm_pk;
typedef struct zkm_main_data zkm_main_data;

/* FriConfig { log_blowup, num_queries, proof_of_work_bits } — crates/stark/src/kb31_poseidon2.rs:203-213 */
typedef struct {
  uint32_t log_blowup;
  uint32_t num_queries;
  uint32_t proof_of_work_bits;
} zkm_fri_config;

/* DuplexChallenger<KoalaBear, Perm, 16, 8> state; same fields as ChallengerPublicValues,
 * crates/recursion/circuit/src/challenger.rs:62-66,117-150. */
typedef struct {
  uint32_t sponge_state[16];
  uint32_t num_inputs;
  uint32_t input_buffer[16];
  uint32_t num_outputs;
  uint32_t output_buffer[16];
} zkm_challenger;

/*
 * Per-chip metadata `open` reads from `MachineChip` (crates/stark/src/chip.rs:24-175) and
 * `StarkProvingKey` (machine.rs:58-75).
 *
 * lookups: the chip's *Local-scope* sends then receives (permutation.rs:110-116), as u32 words:
 *     n_sends, n_receives,
 *     per lookup: kind (LookupKind as usize, lookup/lookup.rs:22-48), n_values,
 *       then n_values+1 VirtualPairCol (values..., multiplicity), each:
 *         n_terms, constant (Montgomery),
 *         n_terms x { (is_main << 31) | column , weight (Montgomery) }
 *
 * program: the chip's constraints (`Air::eval` of chip.rs:257-276 *including* the permutation
 * constraints of permutation.rs:205-347) recorded as straight-line bytecode; see
 * "Constraint bytecode" below. The Rust shim records it with a symbolic AirBuilder
 * (same mechanism as lookup/builder.rs:14-112).
 */
typedef struct {
  const char* name;
  uint32_t main_width;
  uint32_t prep_width;          /* 0 if the chip has no preprocessed trace */
  int32_t prep_index;           /* index into the pk's preprocessed traces, -1 if none (pk.chip_ordering) */
  uint32_t log_quotient_degree; /* chip.rs:81-89 */
  uint32_t local_only;          /* chip.local_only(): open main/prep at zeta only (prover.rs:526-544) */
  uint32_t commit_scope_global; /* 1: global_cumulative_sum = last 14 cells of the last main row (prover.rs:352-361) */
  uint32_t num_constraints;     /* pk.constraints_map[name] (prover.rs:447-456) */
  const uint32_t* lookups;
  uint32_t lookups_len;
  const uint32_t* program;
  uint32_t program_len;
} zkm_chip_desc;

/*
 * Constraint bytecode. Header: { n_instr, n_ext_regs, n_constraints, n_base_regs }, then n_instr
 * instructions of 2 words: w0 = op | dst<<8 | a<<16 | b<<24 ; w1 = immediate.
 * Two register files: base-field registers and extension-field registers; the opcode says which
 * file each operand lives in ("B": base, "E": extension, "EB": dst and a extension, b base).
 * Loads (dst <- input; imm = column / index; a = row offset 0:local 1:next):
 */
enum {
  ZKM_OP_LD_MAIN = 1,     /* base  main[a][imm]                                  */
  ZKM_OP_LD_PREP = 2,     /* base  preprocessed[a][imm]                          */
  ZKM_OP_LD_PERM = 3,     /* ext   permutation[a][imm] (imm = ext column)        */
  ZKM_OP_LD_CONST = 4,    /* base  constant imm (Montgomery)                     */
  ZKM_OP_LD_PV = 5,       /* base  public_values[imm]                            */
  ZKM_OP_LD_CHALLENGE = 6,/* ext   permutation challenge imm (0: alpha, 1: beta) */
  ZKM_OP_LD_LOCAL_SUM = 7,/* ext   local_cumulative_sum                          */
  ZKM_OP_LD_GLOBAL_SUM = 8,/* base global_cumulative_sum word imm (0-6: x, 7-13: y) */
  ZKM_OP_LD_IS_FIRST = 9, /* base  is_first_row selector                         */
  ZKM_OP_LD_IS_LAST = 10, /* base  is_last_row selector                          */
  ZKM_OP_LD_IS_TRANS = 11,/* base  is_transition selector                        */
  /* arithmetic: dst <- a (op) b. "B" = both base, "E" = both ext, "EB" = a ext, b base */
  ZKM_OP_ADD_B = 16, ZKM_OP_SUB_B = 17, ZKM_OP_MUL_B = 18, ZKM_OP_NEG_B = 19,
  ZKM_OP_ADD_E = 20, ZKM_OP_SUB_E = 21, ZKM_OP_MUL_E = 22, ZKM_OP_NEG_E = 23,
  ZKM_OP_ADD_EB = 24, ZKM_OP_SUB_EB = 25, ZKM_OP_MUL_EB = 26,
  /* constraints, in order (folder.rs:79-102): accumulator += alpha^(C-1-k) * reg a */
  ZKM_OP_ASSERT_B = 32,
  ZKM_OP_ASSERT_E = 33
};

const char* zkm_last_error(void);
/* What this binary was built from: "ZKM_SOURCES_DIGEST=<16 hex>;hipcc=<version>;arch=gfx950". The digest is sha256 over every file of
 * ziren_amd/csrc plus this header, in name order (ziren_amd/build.py sources_digest); the build script compares it with the tree's before
 * deciding not to compile, and a benchmark refuses a library whose digest is not the tree's. */
const char* zkm_build_info(void);

/* ---- context ---------------------------------------------------------------------------- */
int zkm_ctx_create(int device, zkm_ctx** out);
void zkm_ctx_destroy(zkm_ctx* ctx);
/* Waits for everything queued on the context: the compute stream and every zkm_events_upload_async copy. */
int zkm_ctx_synchronize(zkm_ctx* ctx);
/* Device buffers are recycled through an exact-size pool; trim returns the idle ones to the driver, and drops the per-height tables
 * (coset twiddles, quotient selectors, row twiddles) a long-lived prover accumulates: they are rebuilt on the next use of a height. */
int zkm_ctx_trim(zkm_ctx* ctx);
/* A cap on what the context's pool may hold from the driver (handed out + cached), in bytes; 0 = none (the default; ZKM_POOL_LIMIT_MB sets
 * it for every context of the process). A request that would cross it first returns the cached buffers to the driver and then fails
 * with "out of device memory ..." in zkm_last_error(): the failing call gives back what it had taken and the context stays usable —
 * where the reference's prover would abort the process on an allocation failure (crates/stark/src/prover.rs:206-208: callers unwrap()).
 * A real hipMalloc failure is handled the same way. zkm_ctx_memory_held: the bytes the pool holds right now. */
int zkm_ctx_set_memory_limit(zkm_ctx* ctx, size_t bytes);
size_t zkm_ctx_memory_held(zkm_ctx* ctx);
/* per-phase GPU time of the last zkm_commit/zkm_open on this context, in milliseconds (HIP
 * events on the context's stream). names/values arrays of capacity cap; returns the count. */
int zkm_ctx_last_timings(zkm_ctx* ctx, const char** names, float* ms, int cap);
/* per-kernel totals of the same call: HIP-event time on the context's stream, launch count and
 * compulsory HBM bytes (each input/output array of a launch counted once). Returns the count. */
int zkm_ctx_kernel_timings(zkm_ctx* ctx, const char** names, float* ms, uint32_t* calls, double* bytes, int cap);
/* mode 0: off; 1: every launch; 2 (default): only launches whose compulsory bytes are >= 256 KiB */
void zkm_ctx_set_kernel_timing(zkm_ctx* ctx, int mode);
/* mode 3: only launches of the kernel `name` (as zkm_ctx_kernel_timings reports it) are timed. A timed launch costs a few microseconds of
 * dispatch latency; a shard proof is ~500 launches, so timing them all slows it by ~2.5 %. A benchmark times the kernel it reports on. */
void zkm_ctx_set_kernel_timing_only(zkm_ctx* ctx, const char* name);
/* on (default): pcs_commit extends a commit's shorter matrices on the context's side stream, under the leaf hashing of the tallest.
 * off: every kernel of a proof runs alone on the main stream, so per-kernel HIP-event durations add up to the proof's busy time
 * (a measurement mode: the proof is the same words, ~0.7 ms slower). The environment's ZKM_LDE_OVERLAP=0 sets the default to off. */
void zkm_ctx_set_lde_overlap(zkm_ctx* ctx, int on);
/* on (default): a commit whose matrices are all on the device hashes the rows of every shorter height in ONE launch in front of the tree
 * levels (merkle::hash_rows; a level with injection then takes two permutations per node). off: a level's kernel runs its node's whole
 * row sponge itself (rounds 1-5; also what a commit does while a matrix is still crossing PCIe). Same digests either way. The
 * environment's ZKM_ROWS_UP_FRONT=0 sets the default to off. */
void zkm_ctx_set_rows_up_front(zkm_ctx* ctx, int on);
/* How the calling thread waits for this context's GPU work. 0 (default): it spins (hipStreamSynchronize, and the FRI layer roots are
 * watched arriving in page-locked memory) — lowest latency, one host core per context for the length of a proof. 1: it queries the
 * stream between 20 us sleeps (not a kernel-level blocking wait: this runtime has none per stream) — about a millisecond more per proof
 * for a context alone, nothing measurable when two contexts share a GPU, a tenth of the CPU time: what a rank of a many-GPU host with
 * few cores per GPU wants (the reference's GPU opts run one prover thread per device too, crates/stark/src/opts.rs:83-110). While such a
 * wait sleeps, the calling thread's timer slack is 1 us (PR_SET_TIMERSLACK); it is put back to what the caller had when the wait ends.
 * The environment's ZKM_HOST_WAIT=blocking makes 1 the default. */
void zkm_ctx_set_host_wait(zkm_ctx* ctx, int blocking);
/* Register a chip-specialised quotient kernel: a gfx950 code object exporting
 * `zkm_quotient_specialized(stark::QuotientArgs)` generated from exactly these program words
 * (ziren_amd/codegen.py; the Rust shim does this once per chip AIR). zkm_open uses it for chips whose
 * program matches and the bytecode interpreter otherwise; both compute the same values. A long program
 * (KeccakSponge: 114 324 instructions) is cut into several kernels, handed over as one container: "ZKMQPART",
 * u32 count, u32 zero, count x u64 lengths, the code objects; the first stores its share of the quotient
 * values, the others add theirs. The (first) code object may also export `zkm_quotient_uniforms(stark::QuotientArgs)`: one
 * wavefront that computes the chip's wave-uniform values (powers of the permutation challenges, constants times challenges) into
 * QuotientArgs::uniforms; zkm_open launches it in front of the chip's kernel(s), which then read the table instead of every
 * wavefront repeating that arithmetic (csrc/quotient_args.cuh). Registering a program again replaces its kernels. */
int zkm_ctx_register_quotient_kernel(zkm_ctx* ctx, const uint32_t* program, uint32_t program_len,
                                     const void* code_object, size_t code_object_len);

/* The same for a chip's permutation trace: a gfx950 code object exporting `zkm_perm_rows_specialized(stark::PermArgs)` generated from
 * exactly these lookups words and this log_quotient_degree (ziren_amd/codegen.py emit_perm_source). zkm_open / zkm_permutation_trace use
 * it for chips whose lookups match and the generic kernel (which walks the blob) otherwise; both compute the same values. */
int zkm_ctx_register_perm_kernel(zkm_ctx* ctx, const uint32_t* lookups, uint32_t lookups_len, uint32_t log_quotient_degree,
                                 const void* code_object, size_t code_object_len);

/* ---- DeviceMatrix ----------------------------------------------------------------------- */
/* Page-locked host memory for trace buffers (the shim's trace generation writes rows straight into it):
 * zkm_matrix_upload from such a buffer is pure DMA at PCIe rate. Pageable buffers work too, slower. */
void* zkm_host_alloc(zkm_ctx* ctx, size_t bytes);
void zkm_host_free(zkm_ctx* ctx, void* p);
/* Upload a row-major host matrix (height a power of two) and lay it out column-major in HBM. Words are Montgomery-form field elements
 * as Plonky3 keeps them (< p = 0x7f000001); a word >= p cannot occur in a RowMajorMatrix<KoalaBear> and is reduced mod p on the way in
 * (same for zkm_matrix_upload_async and zkm_tracegen_flat), so the device always holds canonical words — the kernels' accumulator
 * bounds assume it. */
int zkm_matrix_upload(zkm_ctx* ctx, const uint32_t* host_row_major, size_t height, size_t width,
                      zkm_matrix** out);
/* The same without waiting: the copy and the transposition are queued on the context's upload streams and the call
 * returns at once. Every consumer (zkm_commit, zkm_prove_shard, zkm_pk_setup, zkm_pcs_commit, zkm_matrix_download)
 * waits for the matrix on the device, in stream order, right before its first use — so when a shard's traces are
 * queued tallest first and then proved, the uploads of the later traces overlap the LDE and hashing of the first ones
 * (CpuProver::commit receives host matrices, prover.rs:258-292: this is that hand-over, pipelined). The host buffer
 * must stay valid and unchanged until zkm_matrix_wait returns, or until a consuming call that used the matrix returns. */
int zkm_matrix_upload_async(zkm_ctx* ctx, const uint32_t* host_row_major, size_t height, size_t width,
                            zkm_matrix** out);
int zkm_matrix_wait(zkm_ctx* ctx, const zkm_matrix* m);
/* Executor events ahead of their trace generation. Queues the copy of `bytes` bytes of event records (any of the zkm_*_event arrays,
 * page-locked for a true asynchronous copy) on the context's DMA stream and returns a device address at once. That address may be passed
 * to the core-shard trace generators whose rows are a function of one event each (zkm_tracegen_alu / _jump / _mov_cond / _branch / _mul /
 * _divrem / _memory_instrs / _misc_instrs / _syscall_instrs / _syscall / _cpu / _cpu_and_program / _program_mults / _memory_local / _global)
 * and in zkm_tracegen_shard's descriptors in place of the host pointer: they then wait for the copy on the device instead of making their
 * own, and read nothing on the host (SyscallCore's filter and Global's u16 check of message[0] run on the device; a SyscallCore trace
 * without a fixed height costs one round trip for the kept count). Called for shard i + 1 right before zkm_prove_shard of shard i, the transfer
 * runs under that proof (the reference's prove-a-record loop hands records to the prover while the previous one is proving:
 * crates/core/machine/src/utils/prove.rs:484-497). The host buffer must stay unchanged until a trace generator that used the address has
 * returned, or zkm_events_free has. */
int zkm_events_upload_async(zkm_ctx* ctx, const void* host_events, size_t bytes, void** device_events_out);
void zkm_events_free(zkm_ctx* ctx, void* device_events);
int zkm_matrix_download(zkm_ctx* ctx, const zkm_matrix* m, uint32_t* host_row_major);
size_t zkm_matrix_height(const zkm_matrix* m);
size_t zkm_matrix_width(const zkm_matrix* m);
void zkm_matrix_free(zkm_ctx* ctx, zkm_matrix* m);

/* ---- Pcs::commit (TwoAdicFriPcs + MerkleTreeMmcs), call sites prover.rs:277,403,497 ------ */
/* Commits n_mats matrices evaluated over the cosets domain_shift[i] * H_{height_i}
 * (domain_shifts == NULL: all 1, i.e. natural domains). Each is coset-LDE'd by 2^log_blowup
 * onto 3 * K, rows bit-reversed, and all are committed under one mixed-height Poseidon2
 * Merkle tree. root_out receives the 8-word commitment. */
int zkm_pcs_commit(zkm_ctx* ctx, size_t n_mats, const zkm_matrix* const* mats,
                   const uint32_t* domain_shifts, uint32_t log_blowup,
                   uint32_t root_out[ZKM_DIGEST_ELEMS], zkm_pcs_data** out);
void zkm_pcs_data_free(zkm_ctx* ctx, zkm_pcs_data* d);
/* test/inspection: copy LDE matrix idx (bit-reversed rows, row-major) to the host. */
int zkm_pcs_data_get_lde(zkm_ctx* ctx, const zkm_pcs_data* d, size_t idx, uint32_t* host_row_major);
/* Mmcs::open_batch(index): rows of every matrix at index >> (log_max_h - log_h_i), written
 * back to back into values_out, and the sibling digests bottom-up into proof_out
 * (log_max_height digests). Mirrors crates/recursion/circuit/src/fri.rs:363-405. */
int zkm_pcs_open_batch(zkm_ctx* ctx, const zkm_pcs_data* d, size_t index, uint32_t* values_out,
                       uint32_t* proof_out);

/* ---- proving key ------------------------------------------------------------------------ */
/* StarkMachine::setup's commitment to the preprocessed traces (machine.rs:406-417) +
 * pk_to_device (prover.rs:62-66). n_prep may be 0 (then the opening has no preprocessed round). */
int zkm_pk_setup(zkm_ctx* ctx, size_t n_prep, const zkm_matrix* const* prep_traces,
                 const uint32_t* prep_local_only, uint32_t pc_start,
                 const uint32_t initial_global_cumulative_sum[14], uint32_t log_blowup,
                 zkm_pk** out);
int zkm_pk_commitment(const zkm_pk* pk, uint32_t root_out[ZKM_DIGEST_ELEMS]);
/* StarkProvingKey::observe_into, machine.rs:79-86 */
int zkm_pk_observe_into(const zkm_pk* pk, zkm_challenger* challenger);
void zkm_pk_free(zkm_ctx* ctx, zkm_pk* pk);

/* ---- MachineProver::commit / open ------------------------------------------------------- */
/* commit (prover.rs:258-292): orders chips by (Reverse(height), name), commits the main traces.
 * The matrices stay owned by the caller but must outlive the returned zkm_main_data.
 * order_out[i] = caller index of the chip at sorted position i (the chip_ordering). */
int zkm_commit(zkm_ctx* ctx, size_t n_chips, const char* const* names,
               const zkm_matrix* const* main_traces, const uint32_t* public_values,
               size_t n_public_values, uint32_t log_blowup, uint32_t main_commit_out[ZKM_DIGEST_ELEMS],
               uint32_t* order_out, zkm_main_data** out);
void zkm_main_data_free(zkm_ctx* ctx, zkm_main_data* d);

/* open (prover.rs:298-653). chips[] is in the caller's order (same order as zkm_commit's
 * names/main_traces). challenger is the post-`pk.observe_into` clone (prove.rs:496) and is
 * advanced in place. The proof is written as the flat word stream documented in
 * INTEGRATION.md ("ShardProof stream", the Appendix-B order of SURVEY.md). If proof_cap is
 * too small the call fails, *proof_len holds the required length, and proof_out holds an
 * unspecified prefix of the stream (it is written in place, not copied at the end). */
int zkm_open(zkm_ctx* ctx, const zkm_pk* pk, zkm_main_data* data, const zkm_chip_desc* chips,
             const zkm_fri_config* fri, uint32_t num_pv_elts, zkm_challenger* challenger,
             uint32_t* proof_out, size_t proof_cap, size_t* proof_len);

/* commit + open for one shard with device-resident traces (prover.rs:679-690). */
int zkm_prove_shard(zkm_ctx* ctx, const zkm_pk* pk, size_t n_chips, const zkm_chip_desc* chips,
                    const zkm_matrix* const* main_traces, const uint32_t* public_values,
                    size_t n_public_values, const zkm_fri_config* fri, uint32_t num_pv_elts,
                    zkm_challenger* challenger, uint32_t* proof_out, size_t proof_cap,
                    size_t* proof_len);

/* ---- device trace generation, ALU chips ------------------- */
/* One executor event of the AddSub / Bitwise / Lt / ShiftLeft / ShiftRight / CloClz chips: byte-for-byte
 * the #[repr(C)] AluEvent of crates/core/executor/src/events/instr.rs:10-26 (opcode numbers:
 * crates/core/executor/src/opcode.rs:26-48), so the shim passes `record.add_sub_events.as_ptr()`. */
typedef struct zkm_alu_event {
  uint32_t pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t hi, a, b, c;
} zkm_alu_event;
enum zkm_alu_chip {
  ZKM_CHIP_ADD_SUB = 0,     /* crates/core/machine/src/alu/add_sub/mod.rs  (19 columns) */
  ZKM_CHIP_BITWISE = 1,     /* .../alu/bitwise/mod.rs                      (18) */
  ZKM_CHIP_LT = 2,          /* .../alu/lt/mod.rs                           (32) */
  ZKM_CHIP_SHIFT_LEFT = 3,  /* .../alu/sll/mod.rs                          (44) */
  ZKM_CHIP_SHIFT_RIGHT = 4, /* .../alu/sr/mod.rs                           (67) */
  ZKM_CHIP_CLO_CLZ = 5      /* .../alu/clo_clz/mod.rs                      (17) */
};
/* size_of::<Cols<u8>>() of the chip (NUM_*_COLS); 0 for an unknown chip. */
size_t zkm_tracegen_alu_width(int chip);
/* `record.byte_lookups` on the device (crates/core/executor/src/record.rs:62: the HashMap<ByteLookupEvent, usize> that
 * every chip's generate_dependencies fills and ByteChip::generate_trace reads): one counter per (ByteOpcode, b, c). */
typedef struct zkm_byte_lookups zkm_byte_lookups;
int zkm_byte_lookups_create(zkm_ctx* ctx, zkm_byte_lookups** out);
void zkm_byte_lookups_free(zkm_ctx* ctx, zkm_byte_lookups* blu);
/* MachineAir::generate_trace for one of the chips above, on the device: `events` (host, n_events records) are
 * copied to HBM and expanded to the padded main trace, returned as a device-resident matrix that
 * zkm_commit / zkm_prove_shard take directly. fixed_log2_rows is the shape's `fixed_log2_rows` or -1
 * for next_power_of_two(n_events) with the reference's minimum of 16 rows; rows past the events are the
 * chip's padding rows. Fails if n_events exceeds the fixed height (the reference panics).
 * If `blu` is not NULL the same pass also performs the chip's generate_dependencies: the byte lookups each
 * event's row records (the `blu` argument of the reference's event_to_row) are counted into it. */
int zkm_tracegen_alu(zkm_ctx* ctx, int chip, const zkm_alu_event* events, size_t n_events,
                     int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out);

/* The Jump chip (crates/core/machine/src/control_flow/jump/): its events are JumpEvents, byte-for-byte the
 * #[repr(C)] struct of crates/core/executor/src/events/instr.rs:200-217. 66 columns (columns.rs:11-39), zero padding
 * rows, no byte lookups. Same contract as zkm_tracegen_alu otherwise. */
typedef struct zkm_jump_event {
  uint32_t pc, next_pc, next_next_pc;
  uint8_t opcode, _pad[3];
  uint32_t a, b, c;
} zkm_jump_event;
size_t zkm_tracegen_jump_width(void);
int zkm_tracegen_jump(zkm_ctx* ctx, const zkm_jump_event* events, size_t n_events, int fixed_log2_rows,
                      zkm_matrix** out);

/* The Branch chip (crates/core/machine/src/control_flow/branch/: BEQ, BNE, BLTZ, BLEZ, BGTZ, BGEZ): BranchEvents have the
 * JumpEvent layout (crates/core/executor/src/events/instr.rs:161-178). 62 columns, zero padding rows; rows of branches
 * that are not taken record byte lookups (range checks of next_pc and next_next_pc), counted into `blu` if given. */
typedef zkm_jump_event zkm_branch_event;
size_t zkm_tracegen_branch_width(void);
int zkm_tracegen_branch(zkm_ctx* ctx, const zkm_branch_event* events, size_t n_events, int fixed_log2_rows,
                        zkm_byte_lookups* blu, zkm_matrix** out);
/* The Mul chip (crates/core/machine/src/alu/mul/mod.rs: MUL, MULT, MULTU): CompAluEvents, byte-for-byte the #[repr(C)]
 * struct of crates/core/executor/src/events/instr.rs:50-73 with its MemoryWriteRecord (events/memory.rs:69-82), 64 bytes.
 * Replaces MulChip::generate_trace (:162-190) and, with `blu`, its generate_dependencies (:192-213): 58 columns
 * including the HI register's memory-access columns, zero padding rows; byte lookups per row: two MSB, eight U16Range
 * (carries), four U8Range pairs (product bytes) and, for a real HI write, the two limbs of the timestamp difference. */
typedef struct zkm_memory_write_record {
  uint32_t value, shard, timestamp, prev_value, prev_shard, prev_timestamp;
} zkm_memory_write_record;
typedef struct zkm_comp_alu_event {
  uint32_t shard, clk, pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t hi, a, b, c;
  zkm_memory_write_record hi_record;
  uint8_t hi_record_is_real, _pad2[3];
} zkm_comp_alu_event;
size_t zkm_tracegen_mul_width(void);
int zkm_tracegen_mul(zkm_ctx* ctx, const zkm_comp_alu_event* events, size_t n_events, int fixed_log2_rows,
                     zkm_byte_lookups* blu, zkm_matrix** out);
/* The DivRem chip (crates/core/machine/src/alu/divrem/mod.rs: DIV, DIVU, MOD, MODU), CompAluEvents as well: replaces
 * DivRemChip::generate_trace (:224-381), which also records the chip's byte lookups (counted into `blu` if given). 106
 * columns: quotient and remainder (get_quotient_and_remainder, crates/core/executor/src/utils.rs:33-43: x / 0 = 2^32 - 1
 * remainder x; i32::MIN / -1 wraps), absolute values, c * quotient over 64 bits and the carries of adding the remainder,
 * three IsZeroWord / IsEqualWord blocks (field inverses), sign flags, the HI access of DIV / DIVU. Zero padding rows. */
size_t zkm_tracegen_divrem_width(void);
int zkm_tracegen_divrem(zkm_ctx* ctx, const zkm_comp_alu_event* events, size_t n_events, int fixed_log2_rows,
                        zkm_byte_lookups* blu, zkm_matrix** out);
/* The Cpu chip (crates/core/machine/src/cpu/): replaces CpuChip::generate_trace and generate_dependencies
 * (cpu/trace.rs:36-115). Events and program are byte-for-byte the reference's own FFI structs: CpuEventFfi
 * (crates/core/executor/src/events/cpu.rs:46-77, 280 bytes) and InstructionFfi (instruction.rs:22-33, 24 bytes), which it
 * already hands to its C++ row builder (`cpu_event_to_row_koalabear`, cpu/trace.rs:320-352). `shard` is
 * public_values.execution_shard; the instruction of an event is program[(pc - pc_base) / 4] (Program::fetch). 67 columns;
 * padding rows have imm_b = imm_c = is_rw_a = 1. Byte lookups (shard, clk limbs, register accesses, bytes of `a`) are
 * counted into `blu` if given. Fails if an event's pc lies outside the program. */
typedef struct zkm_memory_read_record { uint32_t value, shard, timestamp, prev_shard, prev_timestamp; } zkm_memory_read_record;
typedef struct zkm_option_memory_record {   /* OptionMemoryRecordEnum: tag Read = 0, Write = 1, None = 2 */
  uint8_t tag, _pad[3];
  zkm_memory_read_record read;
  zkm_memory_write_record write;
} zkm_option_memory_record;
typedef struct zkm_option_u32 { uint8_t tag, _pad[3]; uint32_t value; } zkm_option_u32;   /* OptionValTag: Some = 0, None = 1 */
typedef struct zkm_cpu_event {
  uint32_t clk, pc, next_pc, next_next_pc, a;
  zkm_option_memory_record a_record;
  uint32_t b;
  zkm_option_memory_record b_record;
  uint32_t c;
  zkm_option_memory_record c_record;
  zkm_option_u32 hi;
  zkm_option_memory_record hi_record, memory_record;
  uint32_t exit_code;
} zkm_cpu_event;
typedef struct zkm_instruction {
  uint8_t opcode, op_a, _pad0[2];
  uint32_t op_b, op_c;
  uint8_t imm_b, imm_c, _pad1[2];
  zkm_option_u32 raw;
} zkm_instruction;
size_t zkm_tracegen_cpu_width(void);
int zkm_tracegen_cpu(zkm_ctx* ctx, const zkm_cpu_event* events, size_t n_events, const zkm_instruction* program, size_t n_instr,
                     uint32_t pc_base, uint32_t shard, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out);
/* The same, and in the same pass over the events (which are uploaded once) the Program chip's multiplicity trace
 * (zkm_tracegen_program_mults below) into *program_mults_out, padded to program_fixed_log2_rows. */
int zkm_tracegen_cpu_and_program(zkm_ctx* ctx, const zkm_cpu_event* events, size_t n_events, const zkm_instruction* program,
                                 size_t n_instr, uint32_t pc_base, uint32_t shard, int fixed_log2_rows, int program_fixed_log2_rows,
                                 zkm_byte_lookups* blu, zkm_matrix** out, zkm_matrix** program_mults_out);
/* The Program chip (crates/core/machine/src/program/mod.rs): its preprocessed table (pc, instruction columns; 14 columns,
 * generate_preprocessed_trace :62-101) for zkm_pk_setup, and its one-column multiplicity trace (generate_trace :113-146:
 * how many CpuEvents fetched each pc). */
int zkm_tracegen_program(zkm_ctx* ctx, const zkm_instruction* program, size_t n_instr, uint32_t pc_base, int fixed_log2_rows,
                         zkm_matrix** out);
int zkm_tracegen_program_mults(zkm_ctx* ctx, const zkm_cpu_event* events, size_t n_events, size_t n_instr, uint32_t pc_base,
                               int fixed_log2_rows, zkm_matrix** out);
/* The MemoryInstructions chip (crates/core/machine/src/memory/instructions/: LB LBU LH LHU LW LWL LWR LL SB SH SW SWL SWR SC):
 * replaces generate_trace (trace.rs:44-84), which also records the byte lookups (counted into `blu` if given). Events are the
 * #[repr(C)] MemInstrEvents of crates/core/executor/src/events/instr.rs:108-136, whose `mem_access` is the #[repr(C)] enum
 * MemoryRecordEnum (events/memory.rs:88-95): a 4-byte tag (Read = 0, Write = 1) followed by the record (read: value, shard,
 * timestamp, prev_shard, prev_timestamp; write: value, shard, timestamp, prev_value, prev_shard, prev_timestamp). 79 columns,
 * zero padding rows. */
typedef struct zkm_mem_instr_event {
  uint32_t shard, clk, pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t a, b, c;
  uint32_t mem_access_tag;
  uint32_t mem_access[6];
  uint32_t prev_a_val;
} zkm_mem_instr_event;
size_t zkm_tracegen_memory_instrs_width(void);
int zkm_tracegen_memory_instrs(zkm_ctx* ctx, const zkm_mem_instr_event* events, size_t n_events, int fixed_log2_rows,
                               zkm_byte_lookups* blu, zkm_matrix** out);
/* The recursion machine's Poseidon2Wide chip, degree 3 (crates/recursion/core/src/chips/poseidon2_wide/): replaces
 * generate_trace (trace.rs:66-120). Events are Poseidon2Events — 32 Montgomery words each, input[16] then output[16]
 * (crates/recursion/core/src/lib.rs, Poseidon2Io) — and every row is one permutation with all the intermediates the AIR
 * constrains: 313 columns (columns/permutation.rs:20-36). Padding rows are the permutation of the zero state. */
int zkm_tracegen_poseidon2_wide(zkm_ctx* ctx, const uint32_t* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out);
/* The SyscallInstrs chip (crates/core/machine/src/syscall/instructions/): replaces generate_trace (trace.rs:38-86). Events are the
 * #[repr(C)] SyscallEvents of crates/core/executor/src/events/syscall.rs:7-29 (56 bytes). 77 columns, zero padding rows, no byte
 * lookups. */
typedef struct zkm_syscall_event {
  uint32_t pc, next_pc, shard, clk;
  zkm_memory_write_record a_record;
  uint8_t a_record_is_real, _pad[3];
  uint32_t syscall_id, arg1, arg2;
} zkm_syscall_event;
size_t zkm_tracegen_syscall_instrs_width(void);
int zkm_tracegen_syscall_instrs(zkm_ctx* ctx, const zkm_syscall_event* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out);
/* The syscall tables SyscallCore (precompile = 0) and SyscallPrecompile (1) (crates/core/machine/src/syscall/chip.rs): replace
 * generate_trace (:211-276; the C++ twins are syscall_core_event_to_row_koalabear / syscall_precompile_event_to_row_koalabear,
 * crates/core/machine/src/sys.rs:40-47, include/syscall.hpp). Core takes the shard's syscall events and keeps those whose code has the
 * send-to-table byte set or names a Linux syscall; Precompile takes the syscall events filed with the shard's precompile events.
 * 11 columns; the U16Range lookups of the four argument half-words (generate_dependencies :178-183) are counted into `blu` if given.
 * Linux syscalls (a code whose second byte is not zero): is_linux and the result half-words come from the event's a_record (prev_value = the
 * code, value = the value returned in $v0). Core's events hold that record already; for the Precompile table, where the reference reads both
 * off the LinuxEvent filed with the syscall event (:223-238), the caller copies syscall_code and v0 into the a_record. */
int zkm_tracegen_syscall(zkm_ctx* ctx, const zkm_syscall_event* events, size_t n_events, int precompile, int fixed_log2_rows,
                         zkm_byte_lookups* blu, zkm_matrix** out);
/* MemoryGlobalInit / MemoryGlobalFinalize (crates/core/machine/src/memory/global.rs): replaces generate_trace (:113-185; the C++ twin is
 * memory_global_event_to_row_koalabear, sys.rs:35-39, include/memory_global.hpp). Events are the #[repr(C)]
 * MemoryInitializeFinalizeEvents of crates/core/executor/src/events/memory.rs:180-209, in any order (sorted by address here as the
 * reference sorts them); previous_addr is the address in the shard's public values previous_init_addr_bits /
 * previous_finalize_addr_bits. 111 columns. Addresses that do not strictly increase are an error. */
typedef struct zkm_memory_init_finalize_event { uint32_t addr, value, shard, timestamp; } zkm_memory_init_finalize_event;
int zkm_tracegen_memory_global(zkm_ctx* ctx, const zkm_memory_init_finalize_event* events, size_t n_events, uint32_t previous_addr,
                               int fixed_log2_rows, zkm_matrix** out);
/* The Poseidon2Permute precompile (crates/core/machine/src/syscall/precompiles/poseidon2/): replaces generate_trace (trace.rs:31-66) and the
 * byte lookups of generate_dependencies (:68-101, counted into `blu` if given). The reference's Poseidon2PermuteEvent
 * (crates/core/executor/src/events/precompiles/poseidon2_permute.rs:9-27) holds Vecs; across the ABI it is flattened: shard, clk,
 * state_addr and the sixteen MemoryWriteRecords of the state words (pre_state[i] = state_records[i].prev_value, post_state[i] = .value,
 * as syscalls/precompiles/poseidon2/permute.rs:30-47 builds them). 973 columns; padding rows carry the permutation of the zero state. */
typedef struct zkm_poseidon2_permute_event { uint32_t shard, clk, state_addr; zkm_memory_write_record state_records[16]; } zkm_poseidon2_permute_event;
int zkm_tracegen_poseidon2_permute(zkm_ctx* ctx, const zkm_poseidon2_permute_event* events, size_t n_events, int fixed_log2_rows,
                                   zkm_byte_lookups* blu, zkm_matrix** out);

/* The KeccakSponge precompile (crates/core/machine/src/syscall/precompiles/keccak_sponge/): replaces generate_trace (trace.rs:59-100) and the
 * byte lookups of generate_dependencies (:33-57, counted into `blu` if given). The reference's KeccakSpongeEvent
 * (crates/core/executor/src/events/precompiles/keccak_sponge.rs:15-46) holds Vecs; across the ABI it is cut into its 36-word blocks, in
 * order, one record per block (twenty-four trace rows each): the state after the block is xored in (xored_state_list[block_index], every
 * u64 as low word, high word), the block's read records (their values are the input words), and — read on the call's first / last block
 * only — the record of the input length (the word at output_addr + 64) and the sixteen output write records. The round columns are those
 * of p3-keccak-air's generate_trace_rows (a git dependency of the reference, restated from the published crate: DESIGN.md). Fails when a
 * call's blocks do not chain through keccak-f or its output records are not the squeezed state. */
typedef struct zkm_keccak_sponge_block {
  uint32_t shard, clk, input_addr, output_addr, input_len_u32s, block_index;
  uint32_t xored_state[50];
  zkm_memory_read_record input_read_records[36];
  zkm_memory_read_record input_length_record;
  zkm_memory_write_record output_write_records[16];
} zkm_keccak_sponge_block;
#define ZKM_KECCAK_SPONGE_WIDTH 3531
int zkm_tracegen_keccak_sponge(zkm_ctx* ctx, const zkm_keccak_sponge_block* blocks, size_t n_blocks, int fixed_log2_rows,
                               zkm_byte_lookups* blu, zkm_matrix** out);

/* The SHA-256 precompiles (crates/core/machine/src/syscall/precompiles/sha256/): replace generate_trace of ShaExtendChip
 * (extend/trace.rs:31-63; 48 rows per call) and ShaCompressChip (compress/trace.rs:32-91; 80 rows per call) and the byte lookups of their
 * generate_dependencies (counted into `blu` if given). The reference's events (crates/core/executor/src/events/precompiles/
 * sha256_extend.rs:9-24, sha256_compress.rs:9-25) hold Vecs of fixed length; across the ABI they are flattened. ShaCompressEvent's `w`
 * and `h` are the values of its read records and are not repeated. Fails when a write record does not hold what its reads give. */
typedef struct zkm_sha_extend_event {
  uint32_t shard, clk, w_ptr;
  zkm_memory_read_record w_i_minus_15_reads[48], w_i_minus_2_reads[48], w_i_minus_16_reads[48], w_i_minus_7_reads[48];
  zkm_memory_write_record w_i_writes[48];
} zkm_sha_extend_event;
typedef struct zkm_sha_compress_event {
  uint32_t shard, clk, w_ptr, h_ptr;
  zkm_memory_read_record h_read_records[8], w_i_read_records[64];
  zkm_memory_write_record h_write_records[8];
} zkm_sha_compress_event;
#define ZKM_SHA_EXTEND_WIDTH 176
#define ZKM_SHA_COMPRESS_WIDTH 262
int zkm_tracegen_sha_extend(zkm_ctx* ctx, const zkm_sha_extend_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                            zkm_matrix** out);
int zkm_tracegen_sha_compress(zkm_ctx* ctx, const zkm_sha_compress_event* events, size_t n_events, int fixed_log2_rows,
                              zkm_byte_lookups* blu, zkm_matrix** out);

/* The EdAddAssign precompile (crates/core/machine/src/syscall/precompiles/edwards/ed_add.rs): replaces generate_trace (:108-153) and the byte
 * lookups of generate_dependencies (:155-186): one Ed25519 point addition per row, the eight big-field gadgets (operations/field/) computed
 * on the device. EllipticCurveAddEvent (crates/core/executor/src/events/precompiles/ec.rs:24-47) flattened: its `p` and `q` are the previous
 * values of the p write records and the values of the q read records. Fails when the words written to p are not p + q. */
typedef struct zkm_ed_add_event {
  uint32_t shard, clk, p_ptr, q_ptr;
  zkm_memory_write_record p_memory_records[16];
  zkm_memory_read_record q_memory_records[16];
} zkm_ed_add_event;
#define ZKM_ED_ADD_WIDTH 1861
int zkm_tracegen_ed_add(zkm_ctx* ctx, const zkm_ed_add_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                        zkm_matrix** out);

/* The EdDecompress precompile (crates/core/machine/src/syscall/precompiles/edwards/ed_decompress.rs): replaces generate_trace (:217-252), byte
 * lookups into `blu`. EdDecompressEvent (crates/core/executor/src/events/precompiles/edwards.rs:13-32) flattened: y_bytes and
 * decompressed_x_bytes are the values of the y read records (at ptr + 32) and of the x write records (at ptr). Fails when y is not below
 * 2^255 - 19, when it is not the y of a curve point, or when the x written is not the root the sign bit selects. */
typedef struct zkm_ed_decompress_event {
  uint32_t shard, clk, ptr, sign;
  zkm_memory_write_record x_memory_records[8];
  zkm_memory_read_record y_memory_records[8];
} zkm_ed_decompress_event;
#define ZKM_ED_DECOMPRESS_WIDTH 1566
int zkm_tracegen_ed_decompress(zkm_ctx* ctx, const zkm_ed_decompress_event* events, size_t n_events, int fixed_log2_rows,
                               zkm_byte_lookups* blu, zkm_matrix** out);

/* The short-Weierstrass precompiles (crates/core/machine/src/syscall/precompiles/weierstrass/weierstrass_add.rs, weierstrass_double.rs): the
 * eight chips Secp256k1 / Secp256r1 / Bn254 / Bls12381 x AddAssign / DoubleAssign; replace their generate_trace (weierstrass_add.rs:182-247,
 * weierstrass_double.rs:202-266), byte lookups into `blu`. `events`: the curve's EllipticCurveAddEvent / EllipticCurveDoubleEvent
 * (crates/core/executor/src/events/precompiles/ec.rs:24-72) flattened, W = 16 words per point (Bls12381: 24):
 *   add:    shard, clk, p_ptr, q_ptr, W zkm_memory_write_record of p, W zkm_memory_read_record of q      (4 + 11 W words)
 *   double: shard, clk, p_ptr, W zkm_memory_write_record of p                                              (3 + 6 W words)
 * p (and q) are the previous values of the p records (the values of the q records). Fails when a coordinate is not below the base-field
 * modulus or the words written to p are not the result. */
enum { ZKM_CURVE_SECP256K1 = 0, ZKM_CURVE_SECP256R1 = 1, ZKM_CURVE_BN254 = 2, ZKM_CURVE_BLS12381 = 3 };
int zkm_tracegen_weierstrass_add(zkm_ctx* ctx, int curve, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                 zkm_matrix** out);
int zkm_tracegen_weierstrass_double(zkm_ctx* ctx, int curve, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                    zkm_matrix** out);

/* Secp256k1Decompress / Secp256r1Decompress / Bls12381Decompress (crates/core/machine/src/syscall/precompiles/weierstrass/weierstrass_decompress.rs):
 * replaces generate_trace (:163-285), byte lookups into `blu`. `curve` is ZKM_CURVE_SECP256K1, _SECP256R1 (the sign bit is y's parity) or
 * _BLS12381 (the bit says y > p - y). `events`: EllipticCurveDecompressEvent (crates/core/executor/src/events/precompiles/ec.rs:74-94) flattened,
 * W = 8 words per field element (Bls12381: 12):
 *   shard, clk, ptr, sign_bit, W zkm_memory_read_record of x (read at ptr + 4 W), W zkm_memory_write_record of y (written at ptr)   (4 + 11 W words)
 * Fails when the bit is not 0 / 1, x is not below the modulus or not on the curve, or the words written are not the root the bit asks for. */
int zkm_tracegen_weierstrass_decompress(zkm_ctx* ctx, int curve, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                        zkm_matrix** out);

/* Uint256MulMod (crates/core/machine/src/syscall/precompiles/uint256/air.rs): replaces generate_trace (:104-203), byte lookups into `blu`. An event is
 * Uint256MulEvent (crates/core/executor/src/events/precompiles/uint256.rs:12-35) flattened: x is the previous values of the x write records (the
 * result is their values, written at clk + 1), y and the modulus the values of their read records (read at clk from y_ptr, contiguous); a
 * zero modulus stands for 2^256. Fails when the words written are not x * y mod modulus, or when x * y / modulus does not fit 256 bits. */
typedef struct {
  uint32_t shard, clk, x_ptr, y_ptr;
  zkm_memory_write_record x_memory_records[8];
  zkm_memory_read_record y_memory_records[8], modulus_memory_records[8];
} zkm_uint256_mul_event;
int zkm_tracegen_uint256_mul(zkm_ctx* ctx, const zkm_uint256_mul_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                             zkm_matrix** out);

/* U256XU2048Mul (crates/core/machine/src/syscall/precompiles/u256x2048_mul/air.rs): replaces generate_trace (:101-229), byte lookups into `blu`. An
 * event is U256xU2048MulEvent (crates/core/executor/src/events/precompiles/u256x2048_mul.rs:9-44) flattened: a and b are the values of their read
 * records (read at clk), lo_ptr / hi_ptr the values of the reads of registers $a2 / $a3, lo (the low 2048 bits of a * b) and hi (the high 256)
 * the values of their write records (written at clk + 1). Fails when the words written are not the product or the pointers do not match. */
typedef struct {
  uint32_t shard, clk, a_ptr, b_ptr, lo_ptr, hi_ptr;
  zkm_memory_read_record lo_ptr_memory, hi_ptr_memory, a_memory_records[8], b_memory_records[64];
  zkm_memory_write_record lo_memory_records[64], hi_memory_records[8];
} zkm_u256x2048_mul_event;
int zkm_tracegen_u256x2048_mul(zkm_ctx* ctx, const zkm_u256x2048_mul_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                               zkm_matrix** out);

/* BooleanCircuitGarble (crates/core/machine/src/syscall/precompiles/boolean_circuit_garble/): replaces generate_trace + generate_dependencies
 * (trace.rs:32-98), byte lookups into `blu`. A call takes 1 + num_gates rows; the BooleanCircuitGarbleEvent
 * (crates/core/executor/src/events/precompiles/boolean_circuit_garble.rs:12-45, Vecs inside) crosses the ABI cut into those rows, in order: a header
 * row (is_gate 0; reads[0] = num_gates_read_record, reads[1..5] = delta_read_records; input_address = input_addr) and a row per gate
 * (is_gate 1; reads = the gate's seventeen gates_read_records; input_address = input_addr + 20 + 68 gate_id; pre_check = 1 when every gate
 * before this one checked; `write` = output_write_record on the last gate's row). Fails when a row does not continue the row before it, a
 * call is cut short, a gate type is not 0 / 7, or the value written is not the conjunction of the checks. The table is padded like the others
 * (at least 16 rows unless fixed_log2_rows says otherwise; the reference pads this chip to the next power of two without that floor). */
typedef struct {
  uint32_t shard, clk, input_address, output_address, is_gate, gate_id, gates_num, pre_check, delta[4];
  zkm_memory_read_record reads[17];
  zkm_memory_write_record write;
} zkm_garble_row;
int zkm_tracegen_boolean_circuit_garble(zkm_ctx* ctx, const zkm_garble_row* rows, size_t n_rows, int fixed_log2_rows, zkm_byte_lookups* blu,
                                        zkm_matrix** out);

/* SysLinux (crates/core/machine/src/syscall/precompiles/sys_linux/): replaces generate_trace + generate_dependencies (trace.rs:32-102), byte lookups
 * into `blu`. An event is LinuxEvent (crates/core/executor/src/events/precompiles/linux.rs:9-28) flattened: read_record = read_records[0] (brk: the
 * BRK register; write: $a2), a3_record = write_records[0], heap_record = write_records[1] (mmap / mmap2 with a0 = 0); records an event does not
 * have are zero. Fails when v0, the value written to $a3 or the new heap are not what the syscall returns. */
typedef struct {
  uint32_t shard, clk, a0, a1, v0, syscall_code;
  zkm_memory_read_record read_record;
  zkm_memory_write_record a3_record, heap_record;
} zkm_linux_event;
int zkm_tracegen_sys_linux(zkm_ctx* ctx, const zkm_linux_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out);

/* The field-tower precompiles (crates/core/machine/src/syscall/precompiles/fptower/fp.rs, fp2_addsub.rs, fp2_mul.rs): Bn254 / Bls12381 x FpOpAssign,
 * Fp2AddSubAssign, Fp2MulAssign; replace their generate_trace, byte lookups into `blu`. `field` is ZKM_CURVE_BN254 or ZKM_CURVE_BLS12381. Events:
 * FpOpEvent / Fp2AddSubEvent / Fp2MulEvent (crates/core/executor/src/events/precompiles/fptower.rs:23-94) flattened — shard, clk, x_ptr, y_ptr,
 * [op: FieldOperation as a word, Add 0, Mul 1, Sub 2; not for Fp2Mul], W zkm_memory_write_record of x, W zkm_memory_read_record of y, W = 8 / 12
 * words for FpOp, 16 / 24 for the Fp2 chips. Fails when an operand is not below the modulus or the words written to x are not the result. */
int zkm_tracegen_fp_op(zkm_ctx* ctx, int field, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out);
int zkm_tracegen_fp2_addsub(zkm_ctx* ctx, int field, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                            zkm_matrix** out);
int zkm_tracegen_fp2_mul(zkm_ctx* ctx, int field, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out);
/* The MiscInstrs chip (crates/core/machine/src/misc/others/: SEXT EXT INS MADDU MSUBU MADD MSUB TEQ): replaces generate_trace
 * (trace.rs:42-84), which also records the byte lookups (counted into `blu` if given). Events are the #[repr(C)] MiscEvents of
 * crates/core/executor/src/events/instr.rs:239-261 (60 bytes). 72 columns, zero padding rows. */
typedef struct zkm_misc_event {
  uint32_t shard, clk, pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t a, b, c, prev_a;
  zkm_memory_write_record hi_record;
} zkm_misc_event;
size_t zkm_tracegen_misc_instrs_width(void);
int zkm_tracegen_misc_instrs(zkm_ctx* ctx, const zkm_misc_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                             zkm_matrix** out);
/* The wrap machine's hash chip, Poseidon2Skinny (crates/recursion/core/src/chips/poseidon2_skinny/): replaces generate_trace
 * (trace.rs:62-118). Same Poseidon2Events; eleven rows of 28 columns per permutation, zero padding. */
int zkm_tracegen_poseidon2_skinny(zkm_ctx* ctx, const uint32_t* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out);
/* The recursion machine's ExpReverseBitsLen chip (crates/recursion/core/src/chips/exp_reverse_bits.rs): replaces generate_trace
 * (:175-226). An ExpReverseBitsEvent is a base and a vector of exponent bits (lib.rs:208-212); here the n bases, all bits end to
 * end and n + 1 offsets into them (Montgomery words; offsets plain). One row per bit, 7 columns, zero padding. BatchFRI, PublicValues,
 * Select and the memory chips need no kernel: their traces are their events end to end (zkm_tracegen_flat). */
int zkm_tracegen_exp_reverse_bits(zkm_ctx* ctx, const uint32_t* bases, const uint32_t* bits, const uint32_t* offsets, size_t n_events,
                                  int fixed_log2_rows, zkm_matrix** out);
/* The MemoryLocal chip (crates/core/machine/src/memory/local.rs): replaces generate_trace (:147-190). Events are the
 * #[repr(C)] MemoryLocalEvents of crates/core/executor/src/events/memory.rs:226-237 (ExecutionRecord::get_local_mem_events),
 * four per row, 56 columns, zero padding. */
typedef struct zkm_memory_record { uint32_t shard, timestamp, value; } zkm_memory_record;
typedef struct zkm_memory_local_event { uint32_t addr; zkm_memory_record initial_mem_access, final_mem_access; } zkm_memory_local_event;
int zkm_tracegen_memory_local(zkm_ctx* ctx, const zkm_memory_local_event* events, size_t n_events, int fixed_log2_rows,
                              zkm_matrix** out);
/* The MovCond chip (crates/core/machine/src/misc/mov_cond/mod.rs: MEQ, MNE, WSBH): MovCondEvents, byte-for-byte the
 * #[repr(C)] struct of crates/core/executor/src/events/instr.rs:286-302. 32 columns, zero padding rows, no byte lookups. */
typedef struct zkm_mov_cond_event {
  uint32_t pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t a, b, c, prev_a;
} zkm_mov_cond_event;
size_t zkm_tracegen_mov_cond_width(void);
int zkm_tracegen_mov_cond(zkm_ctx* ctx, const zkm_mov_cond_event* events, size_t n_events, int fixed_log2_rows,
                          zkm_matrix** out);

/* GlobalChip::generate_trace + generate_dependencies (crates/core/machine/src/global/mod.rs:75-197): GlobalLookupEvents (the
 * #[repr(C)] record of crates/core/executor/src/events/global.rs:6-15) -> the 99-column trace: each message lifted to a point of
 * the septic curve (operations/global_lookup.rs:26-92), the running sum of the points from the start digest by a parallel scan
 * (operations/global_accumulation.rs:75-113); the U16Range lookup of message[0] per event is counted into `blu`. The matrix's last
 * row ends with the shard's global_cumulative_sum (commit_scope_global in zkm_chip_desc). Errors: message[0] >= 2^16; a running sum
 * that meets the point at infinity or a message's own x-coordinate (the AIR cannot express those rows either). */
typedef struct { uint32_t message[7]; uint8_t is_receive; uint8_t kind; uint8_t pad[2]; } zkm_global_lookup_event;
#define ZKM_GLOBAL_WIDTH 99
int zkm_tracegen_global(zkm_ctx* ctx, const zkm_global_lookup_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                        zkm_matrix** out);

/* ByteChip::trace() — the Byte chip's preprocessed table, 65536 x 12, row (b << 8 | c)
 * (crates/core/machine/src/bytes/mod.rs:31-104, columns bytes/columns.rs:12-46), generated on the device. */
int zkm_tracegen_byte_table(zkm_ctx* ctx, zkm_matrix** out);
/* ByteChip::generate_trace (bytes/trace.rs:46-66): the 65536 x 10 multiplicity trace, one column per ByteOpcode, from
 * the lookups counted in `blu`. extra_counts (may be NULL) is a 65536 x 10 row-major array of plain counts —
 * `record.byte_lookups` of the chips whose dependencies stay on the host — added before the conversion to field
 * elements. `blu` is left unchanged. */
int zkm_tracegen_byte_mults(zkm_ctx* ctx, const zkm_byte_lookups* blu, const uint32_t* extra_counts,
                            zkm_matrix** out);

/* ---- generate_traces of a whole core shard in one call (crates/stark/src/prover.rs:70-108) ------------------------------------------
 * One descriptor per chip of the shard; out[i] receives descriptor i's trace. Every generator is queued on the context's stream behind
 * the copy of its events — `events` is a host pointer (copied in front of the kernel) or an address zkm_events_upload_async returned (the
 * stream waits for that copy; nothing is read on the host, SyscallCore's filter and Global's range check included) — and the call
 * synchronises once, at the end. Byte lookups of all chips are counted into `blu` (NULL: a table that lives for the call);
 * ZKM_TG_BYTE_MULTS yields the Byte chip's multiplicity trace over it after every other generator has run, wherever it stands in the list;
 * ZKM_TG_PROGRAM_MULTS yields the Program chip's multiplicities, counted by the ZKM_TG_CPU descriptor's pass (its fixed_log2_rows is the
 * Program chip's height). On failure no matrix is returned (out[] all NULL). */
enum zkm_tracegen_kind {
  ZKM_TG_ALU = 0,            /* `chip` = enum zkm_alu_chip; zkm_alu_event */
  ZKM_TG_CPU = 1,            /* zkm_cpu_event; program, n_instr, pc_base, shard */
  ZKM_TG_BRANCH = 2, ZKM_TG_JUMP = 3, ZKM_TG_MOV_COND = 4, ZKM_TG_MUL = 5, ZKM_TG_DIVREM = 6, ZKM_TG_MEMORY_INSTRS = 7, ZKM_TG_MISC_INSTRS = 8,
  ZKM_TG_SYSCALL_INSTRS = 9, ZKM_TG_SYSCALL_CORE = 10, ZKM_TG_SYSCALL_PRECOMPILE = 11, ZKM_TG_MEMORY_LOCAL = 12, ZKM_TG_GLOBAL = 13,
  ZKM_TG_BYTE_MULTS = 14,    /* no events */
  ZKM_TG_PROGRAM_MULTS = 15, /* no events: needs a ZKM_TG_CPU descriptor in the same call */
  /* the recursion machine's chips (crates/recursion/core/src/chips): a compress / shrink shard's traces in one call as well */
  ZKM_TG_FLAT = 16,              /* zkm_tracegen_flat: events = the records' words, n_events = their number, `chip` = the trace width */
  ZKM_TG_POSEIDON2_WIDE = 17,    /* zkm_tracegen_poseidon2_wide: n_events permutations of 32 words */
  ZKM_TG_EXP_REVERSE_BITS = 18   /* events = one buffer [bases (n_events) | offsets (n_events + 1) | bits (n_instr)], n_instr = offsets[n_events] = the rows */
};
typedef struct zkm_tracegen_desc {
  uint32_t kind;             /* enum zkm_tracegen_kind */
  int32_t chip;              /* ZKM_TG_ALU only */
  const void* events;
  size_t n_events;
  int32_t fixed_log2_rows;   /* the shape's height, or -1: next power of two (>= 16) */
  uint32_t no_byte_lookups;  /* 1: this chip's generate_dependencies is not run (its byte lookups are not counted) */
  const zkm_instruction* program;   /* ZKM_TG_CPU only, host memory */
  size_t n_instr;
  uint32_t pc_base, shard;
} zkm_tracegen_desc;
int zkm_tracegen_shard(zkm_ctx* ctx, const zkm_tracegen_desc* descs, size_t n, zkm_byte_lookups* blu, zkm_matrix** out);

/* generate_trace / generate_preprocessed_trace of the chips whose rows are their event (or instruction) records laid
 * end to end and zero padded — the recursion machine's BaseAlu and ExtAlu chips
 * (crates/recursion/core/src/chips/alu_base.rs:99-137,204-222, alu_ext.rs): `words` are n_words field elements
 * (Montgomery words, as the records lie in memory), `width` the trace width; the height is
 * next_power_of_two(ceil(n_words / width)) with the minimum of 16 rows, or 2^fixed_log2_rows. */
int zkm_tracegen_flat(zkm_ctx* ctx, const uint32_t* words, size_t n_words, size_t width,
                      int fixed_log2_rows, zkm_matrix** out);

/* ---- fine-grained entry points (parity tests, micro-benchmarks) ------------------------- */
/* Poseidon2 width-16 permutation on n states (n x 16 words, in place), on the GPU.
 * zkm_primitives::poseidon2_init, crates/primitives/src/lib.rs:1107-1122. */
int zkm_poseidon2_permute_batch(zkm_ctx* ctx, uint32_t* states, size_t n);
/* The same permutation through the integer-pipe formulation (what the lane-parallel kernels near a tree's root
 * compute with); the entry point above runs the FP64-pipe formulation every hashing kernel uses. */
int zkm_poseidon2_permute_batch_int(zkm_ctx* ctx, uint32_t* states, size_t n);
/* Radix2Dit::coset_lde_batch + bit_reverse_rows on a host matrix:
 * out is (height << log_blowup) x width row-major. lde_shift multiplies the evaluation coset
 * (Pcs::commit passes GENERATOR / domain_shift). */
int zkm_coset_lde_batch(zkm_ctx* ctx, const uint32_t* host_row_major, size_t height, size_t width,
                        uint32_t log_blowup, uint32_t lde_shift, uint32_t* out_row_major);

/* generate_permutation_trace of one chip (crates/stark/src/permutation.rs:102-196; called from prover.rs:337-365 inside `open`): the LogUp columns —
 * per batch of lookups the sum of multiplicity / (alpha + kind + sum beta^k value_k) — and the running sum in the last extension column, from the
 * chip's main (and preprocessed, may be null when prep_width is 0) trace and the two permutation challenges (alpha, beta: 8 Montgomery words).
 * `*out`: height x 4 perm_ext_width; local_sum: the cumulative sum (4 words). A test entry point; `zkm_open` does this itself. */
int zkm_permutation_trace(zkm_ctx* ctx, const zkm_chip_desc* chip, const zkm_matrix* main, const zkm_matrix* prep, const uint32_t challenges[8],
                          zkm_matrix** out, uint32_t local_sum[4]);

/* Host-side duplex challenger (DuplexChallenger<KoalaBear,Perm,16,8>), used by zkm_open. */
void zkm_challenger_init(zkm_challenger* c);
void zkm_challenger_observe(zkm_challenger* c, const uint32_t* values, size_t n);
uint32_t zkm_challenger_sample(zkm_challenger* c);
uint32_t zkm_challenger_sample_bits(zkm_challenger* c, uint32_t bits);

/* ---- host-side arithmetic of the transcript layer (no GPU needed; CPU-only parity tests) --
 * The duplex challenger above permutes with zkm_host_poseidon2_permute (integer formulation);
 * zkm_host_poseidon2_permute_f64 is the host build of the FP64 formulation the GPU hashing kernels run
 * (same IEEE operations), so its exactness can be checked without a GPU. Words are Montgomery form. */
void zkm_host_poseidon2_permute(uint32_t state[16]);
void zkm_host_poseidon2_permute_f64(uint32_t state[16]);
/* Host mirrors of the hashing kernels' data flow, state kept in doubles between permutations exactly as on the device:
 * PaddingFreeSponge over n words (hash_leaves / absorb_row: the capacity is carried on unreduced), and one tree node
 * compress(left, right) followed, when n > 0, by compress(node, hash(row)) with both halves handed over as unreduced doubles
 * (compress_layer). */
void zkm_host_poseidon2_f64_sponge(const uint32_t* words, size_t n, uint32_t digest[8]);
void zkm_host_poseidon2_f64_compress_inject(const uint32_t left[8], const uint32_t right[8], const uint32_t* row, size_t n,
                                            uint32_t out[8]);
/* The largest magnitudes the host build of the FP64 permutation has met on this thread since the last reset, at the points
 * its exactness argument rests on: out[0] a permutation input, out[1] the sum of lane magnitudes in a partial round,
 * out[2] an input of the first sixteen (wide) S-boxes, out[3] a lane after a partial round, out[4] the sum of the magnitudes of the lanes held as dyadic rationals
 * in a partial round; out[5] = how many operations on those lanes lost a bit (checked against 64-bit significands) plus how many
 * four-instruction modular products were not exact, not congruent or out of range (checked against 128-bit integers): must be 0;
 * out[6] an input of the other (nine-instruction) S-boxes. (The device build has no probes.) */
void zkm_host_poseidon2_f64_audit(double out[7], int reset);
void zkm_host_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);
void zkm_host_ext_inv(const uint32_t a[4], uint32_t out[4]);
uint32_t zkm_host_field_mul(uint32_t a, uint32_t b);
uint32_t zkm_host_field_inv(uint32_t a);
/* Host build of the reduction the generated quotient / permutation kernels finish a linear form with (csrc/kb31.cuh reduce96_bounded):
 * (hi 2^64 + lo) / 2^32 mod p in [0, p), valid for hi 2^64 + lo < 127 * 2^63. Test hook (tests/test_host_abi.py holds it against
 * Python integers at the bound's edges). */
uint32_t zkm_host_reduce96_bounded(uint32_t hi, uint64_t lo);
uint32_t zkm_host_two_adic_generator(uint32_t bits);

#ifdef __cplusplus
}
#endif
#endif /* ZKM_HIP_H */
