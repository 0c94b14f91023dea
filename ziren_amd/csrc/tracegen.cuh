// Device trace generation for the ALU chips that consume AluEvents: one thread builds one row in VGPRs and
// stores it column by column into the column-major matrix the commit path reads, so a wavefront writes 64
// consecutive words of each column and the trace never exists in row-major form or crosses PCIe (the events,
// 28 bytes each, do).
//
// Replaces <Chip as MachineAir>::generate_trace for (crates/core/machine/src/alu/...)
//   AddSub      add_sub/mod.rs:44-68 (columns), :161-181 (event_to_row), operations/add.rs:13-57
//   Bitwise     bitwise/mod.rs:36-64, :160-195
//   Lt          lt/mod.rs:36-86, :209-274
//   ShiftLeft   sll/mod.rs:70-104, :232-287, padding rows :157-165
//   ShiftRight  sr/mod.rs:88-137, :232-339, padding rows :183-186
//   CloClz      clo_clz/mod.rs:41-63, :105-133, padding rows :147-163
// and for the Jump chip (JumpEvents): crates/core/machine/src/control_flow/jump/columns.rs:11-39, trace.rs:92-113,
// operations/koala_bear_word.rs:27-42; and the MovCond chip (MovCondEvents): misc/mov_cond/mod.rs:38-65, :141-160,
// operations/is_zero_word.rs:22-38; and the Branch chip (BranchEvents): control_flow/branch/columns.rs:11-66, trace.rs:94-141.
// Values are stored in Montgomery form, the in-memory form of the reference's KoalaBear (RowMajorMatrix<KoalaBear>).
#pragma once
#include "kb31.cuh"
#include "poseidon2.cuh"
#include "septic.cuh"
#include "bigfield.cuh"

namespace tracegen {

struct AluEvent {  // #[repr(C)] AluEvent, crates/core/executor/src/events/instr.rs:10-26
  uint32_t pc, next_pc;
  uint32_t opcode;  // u8 + three bytes of padding; only the low byte is meaningful
  uint32_t hi, a, b, c;
};

// crates/core/executor/src/opcode.rs:26-48
enum : uint32_t { ADD = 0, SUB = 1, SLL = 9, SRL = 10, SRA = 11, ROR = 12, SLT = 13, SLTU = 14, AND = 15, OR = 16, XOR = 17, NOR = 18, CLZ = 19, CLO = 20 };
enum Chip { ADD_SUB = 0, BITWISE = 1, LT = 2, SHIFT_LEFT = 3, SHIFT_RIGHT = 4, CLO_CLZ = 5, NUM_ALU_CHIPS = 6, JUMP = 6, MOV_COND = 7, BRANCH = 8, MUL = 9, DIVREM = 10, MEMORY_INSTRS = 11, SYSCALL_INSTRS = 12, MISC_INSTRS = 13, SYSCALL_CORE = 14, SYSCALL_PRECOMPILE = 15, NUM_CHIPS = 16 };

__host__ __device__ constexpr int chip_width(int chip) {
  return chip == ADD_SUB ? 19 : chip == BITWISE ? 18 : chip == LT ? 32 : chip == SHIFT_LEFT ? 44 : chip == SHIFT_RIGHT ? 67 : chip == CLO_CLZ ? 17 : chip == JUMP ? 66 : chip == MOV_COND ? 32 : chip == BRANCH ? 62 : chip == MUL ? 58 : chip == DIVREM ? 106 : chip == MEMORY_INSTRS ? 79 : chip == SYSCALL_INSTRS ? 77 : chip == MISC_INSTRS ? 72 : chip == SYSCALL_CORE || chip == SYSCALL_PRECOMPILE ? 11 : 0;
}
// words per event record: the seven-word AluEvent / JumpEvent / BranchEvent / MovCondEvent, the sixteen-word CompAluEvent
__host__ __device__ constexpr int event_words(int chip) { return chip == MUL || chip == DIVREM || chip == MEMORY_INSTRS ? 16 : chip == SYSCALL_INSTRS || chip == SYSCALL_CORE || chip == SYSCALL_PRECOMPILE ? 14 : chip == MISC_INSTRS ? 15 : 7; }

constexpr int THREADS = 256;

// The IsZero / IsEqual blocks of several chips need the field inverse of a byte, of a difference of two bytes, or of a sum of three
// bytes: a table of 1/1 .. 1/767 (canonical) in constant memory instead of a 31-bit exponentiation per cell.
constexpr uint32_t SMALL_INV = 768;
__constant__ uint32_t d_small_inv[SMALL_INV];
__device__ __forceinline__ uint32_t small_inverse(uint32_t x) {   // canonical in, canonical out; 0 for 0
  if (x < SMALL_INV) return d_small_inv[x];
  if (x > kb::P - SMALL_INV) return kb::P - d_small_inv[kb::P - x];   // 1 / (-d) = -(1 / d)
  return kb::from_monty(kb::inv(kb::to_monty(x)));
}
inline hipError_t upload_tables() {
  static uint32_t inv[SMALL_INV];
  inv[0] = 0;
  for (uint32_t i = 1; i < SMALL_INV; i++) inv[i] = kb::from_monty(kb::inv(kb::to_monty(i)));
  return hipMemcpyToSymbol(HIP_SYMBOL(d_small_inv), inv, sizeof inv);
}

// Rows are built as canonical integers (bytes, flags, pcs); alu_rows converts each cell to Montgomery form as it
// stores it (from_canonical_u32 of any u32), byte_mults reads the same cells to form the byte lookups.
__device__ __forceinline__ uint32_t fbool(bool b) { return b ? 1u : 0u; }
__device__ __forceinline__ void word(uint32_t* dst, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) dst[i] = (v >> (8 * i)) & 0xff;
}

template <int CHIP> __device__ __forceinline__ void event_row(const AluEvent& e, uint32_t* r);
template <int CHIP> __device__ __forceinline__ void padding_row(uint32_t* r) {}

template <> __device__ __forceinline__ void event_row<ADD_SUB>(const AluEvent& e, uint32_t* r) {
  const bool is_add = e.opcode == ADD;
  const uint32_t op1 = is_add ? e.b : e.a, op2 = e.c;
  r[0] = e.pc;
  r[1] = e.next_pc;
  word(r + 2, op1 + op2);
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    carry = (((op1 >> (8 * i)) & 0xff) + ((op2 >> (8 * i)) & 0xff) + carry) > 255;
    r[6 + i] = fbool(carry);
  }
  word(r + 9, op1);
  word(r + 13, op2);
  r[17] = fbool(is_add);
  r[18] = fbool(e.opcode == SUB);
}

template <> __device__ __forceinline__ void event_row<BITWISE>(const AluEvent& e, uint32_t* r) {
  r[0] = e.pc;
  r[1] = e.next_pc;
  word(r + 2, e.a);
  word(r + 6, e.b);
  word(r + 10, e.c);
  r[14] = fbool(e.opcode == NOR);
  r[15] = fbool(e.opcode == XOR);
  r[16] = fbool(e.opcode == OR);
  r[17] = fbool(e.opcode == AND);
}

template <> __device__ __forceinline__ void event_row<LT>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, IS_SLT = 2, IS_SLTU = 3, A = 4, B = 8, C = 12, BYTE_FLAGS = 16, B_MASKED = 20, C_MASKED = 21,
         NOT_EQ_INV = 22, MSB_B = 23, MSB_C = 24, BIT_B = 25, BIT_C = 26, SLTU_ = 27, IS_COMP_EQ = 28, IS_SIGN_EQ = 29, CMP_BYTES = 30 };
  const bool slt = e.opcode == SLT;
  r[PC] = e.pc;
  r[NEXT_PC] = e.next_pc;
  r[IS_SLT] = fbool(slt);
  r[IS_SLTU] = fbool(e.opcode == SLTU);
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + C, e.c);
  r[B_MASKED] = (e.b >> 24) & 0x7f;
  r[C_MASKED] = (e.c >> 24) & 0x7f;
  // SLT compares with the sign bits masked off
  const uint32_t bc = slt ? e.b & 0x7fffffffu : e.b, cc = slt ? e.c & 0x7fffffffu : e.c;
  const uint32_t diff = bc ^ cc;
  // index of the most significant differing byte (or none)
  const int top = diff ? (31 - __clz(diff)) >> 3 : -1;
  const uint32_t b_byte = top >= 0 ? (bc >> (8 * top)) & 0xff : 0, c_byte = top >= 0 ? (cc >> (8 * top)) & 0xff : 0;
#pragma unroll
  for (int i = 0; i < 4; i++) r[BYTE_FLAGS + i] = fbool(i == top);
  const bool sltu = top >= 0 && b_byte < c_byte;
  r[SLTU_] = fbool(sltu);
  r[IS_COMP_EQ] = fbool(diff == 0);
  r[NOT_EQ_INV] = top >= 0 ? kb::from_monty(kb::inv(kb::sub(kb::to_monty(b_byte), kb::to_monty(c_byte)))) : 0u;
  r[CMP_BYTES] = b_byte;
  r[CMP_BYTES + 1] = c_byte;
  const uint32_t msb_b = e.b >> 31, msb_c = e.c >> 31;
  r[MSB_B] = fbool(msb_b);
  r[MSB_C] = fbool(msb_c);
  r[BIT_B] = fbool(msb_b && slt);
  r[BIT_C] = fbool(msb_c && slt);
  r[IS_SIGN_EQ] = fbool(!slt || msb_b == msb_c);
}

template <> __device__ __forceinline__ void event_row<SHIFT_LEFT>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, A = 2, B = 6, C = 10, C_LSB = 14, BY_BITS = 22, MULT = 30, RESULT = 31, CARRY = 35, BY_BYTES = 39, IS_REAL = 43 };
  r[PC] = e.pc;
  r[NEXT_PC] = e.next_pc;
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + C, e.c);
  const uint32_t nbits = e.c & 7, nbytes = (e.c & 31) >> 3;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[C_LSB + i] = fbool((e.c >> i) & 1);
    r[BY_BITS + i] = fbool(nbits == (uint32_t)i);
  }
  r[MULT] = (1u << nbits);
  // b * 2^nbits byte by byte: limb i keeps 8 bits, the rest carries into limb i+1
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t v = (((e.b >> (8 * i)) & 0xff) << nbits) + carry;
    carry = v >> 8;
    r[RESULT + i] = (v & 0xff);
    r[CARRY + i] = carry;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) r[BY_BYTES + i] = fbool(nbytes == (uint32_t)i);
  r[IS_REAL] = 1;
}
template <> __device__ __forceinline__ void padding_row<SHIFT_LEFT>(uint32_t* r) {
  r[22] = 1;  // shift_by_n_bits[0]
  r[30] = 1;  // bit_shift_multiplier
  r[39] = 1;  // shift_by_n_bytes[0]
}

template <> __device__ __forceinline__ void event_row<SHIFT_RIGHT>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, B = 2, C = 6, BY_BITS = 10, BY_BYTES = 18, BYTE_RES = 22, BIT_RES = 30, SHR_CARRY = 38, SHR_SHIFTED = 46,
         B_MSB = 54, C_LSB = 55, IS_SRL = 63, IS_ROR = 64, IS_SRA = 65, IS_REAL = 66 };
  r[PC] = e.pc;
  r[NEXT_PC] = e.next_pc;
  word(r + B, e.b);
  word(r + C, e.c);
  const uint32_t nbits = e.c & 7, nbytes = (e.c & 31) >> 3;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[BY_BITS + i] = fbool(nbits == (uint32_t)i);
    r[C_LSB + i] = fbool((e.c >> i) & 1);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) r[BY_BYTES + i] = fbool(nbytes == (uint32_t)i);
  // 64-bit extension of b: sign (SRA), b repeated (ROR) or zero (SRL), then whole bytes shifted out
  const uint32_t hi = e.opcode == SRA ? (uint32_t)((int32_t)e.b >> 31) : e.opcode == ROR ? e.b : 0u;
  const uint64_t bytes = (((uint64_t)hi << 32) | e.b) >> (8 * nbytes);
  uint32_t last_carry = 0;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    const uint32_t byte = (uint32_t)(bytes >> (8 * i)) & 0xff;
    const uint32_t shifted = byte >> nbits, carry = byte & ((1u << nbits) - 1);  // bytes/utils.rs:2-11 (shr_carry)
    r[BYTE_RES + i] = byte;
    r[SHR_CARRY + i] = carry;
    r[SHR_SHIFTED + i] = shifted;
    r[BIT_RES + i] = (shifted + (last_carry << (8 - nbits))) & 0xff;
    last_carry = carry;
  }
  r[B_MSB] = fbool(e.b >> 31);
  r[IS_SRL] = fbool(e.opcode == SRL);
  r[IS_ROR] = fbool(e.opcode == ROR);
  r[IS_SRA] = fbool(e.opcode == SRA);
  r[IS_REAL] = 1;
}
template <> __device__ __forceinline__ void padding_row<SHIFT_RIGHT>(uint32_t* r) {
  r[10] = 1;  // shift_by_n_bits[0]
  r[18] = 1;  // shift_by_n_bytes[0]
}

template <> __device__ __forceinline__ void event_row<CLO_CLZ>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, A = 2, B = 6, BB = 10, IS_BB_ZERO = 14, IS_CLZ = 15, IS_REAL = 16 };
  const uint32_t bb = e.opcode == CLZ ? e.b : ~e.b;
  r[PC] = e.pc;
  r[NEXT_PC] = e.next_pc;
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + BB, bb);
  r[IS_BB_ZERO] = fbool(bb == 0);
  r[IS_CLZ] = fbool(e.opcode == CLZ);
  r[IS_REAL] = 1;
}
template <> __device__ __forceinline__ void padding_row<CLO_CLZ>(uint32_t* r) {
  r[2] = 32;  // a = Word::from(32)
  r[14] = 1;  // is_bb_zero
}

// Jump chip. JumpEvent (crates/core/executor/src/events/instr.rs:200-217) is seven words like AluEvent, laid out
// pc, next_pc, next_next_pc, opcode, a, b, c; alu_rows hands it over in the AluEvent slots (opcode slot = word 2 etc.),
// so the fields are re-read here by position.
__device__ __forceinline__ void range_checker(uint32_t* r, uint32_t value) {  // KoalaBearWordRangeChecker::populate
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = (value >> (24 + i)) & 1;
  r[8] = r[0] & r[1];
#pragma unroll
  for (int i = 0; i < 5; i++) r[9 + i] = r[8 + i] & r[2 + i];
}
template <> __device__ __forceinline__ void event_row<JUMP>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, NEXT_PC_RC = 5, NEXT_NEXT_PC = 19, NEXT_NEXT_PC_RC = 23, OP_A = 37, OP_B = 41, OP_C = 45, IS_JUMP = 49,
         IS_JUMPI = 50, IS_JUMPDIRECT = 51, OP_A_RC = 52 };
  const uint32_t pc = e.pc, next_pc = e.next_pc, next_next_pc = e.opcode, opcode = e.hi & 0xff, a = e.a, b = e.b, c = e.c;
  r[PC] = pc;
  r[IS_JUMP] = fbool(opcode == 27);
  r[IS_JUMPI] = fbool(opcode == 28);
  r[IS_JUMPDIRECT] = fbool(opcode == 29);
  word(r + OP_A, a);
  word(r + OP_B, b);
  word(r + OP_C, c);
  range_checker(r + OP_A_RC, a);
  word(r + NEXT_PC, next_pc);
  range_checker(r + NEXT_PC_RC, next_pc);
  word(r + NEXT_NEXT_PC, next_next_pc);
  range_checker(r + NEXT_NEXT_PC_RC, next_next_pc);
}

// Branch chip. BranchEvent (events/instr.rs:161-178) has JumpEvent's layout: pc, next_pc, next_next_pc, opcode, a, b, c.
template <> __device__ __forceinline__ void event_row<BRANCH>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, NEXT_PC_RC = 5, TARGET_PC = 19, NEXT_NEXT_PC = 23, NEXT_NEXT_PC_RC = 27, OP_A = 41, OP_B = 45, OP_C = 49,
         IS_BEQ = 53, IS_BNE = 54, IS_BLTZ = 55, IS_BLEZ = 56, IS_BGTZ = 57, IS_BGEZ = 58, IS_BRANCHING = 59, A_GT_B = 60, A_LT_B = 61 };
  const uint32_t pc = e.pc, next_pc = e.next_pc, next_next_pc = e.opcode, opcode = e.hi & 0xff, a = e.a, b = e.b, c = e.c;
  const bool eq = a == b, lt = (int32_t)a < (int32_t)b, gt = (int32_t)a > (int32_t)b;
  const bool branching = opcode == 21 ? eq : opcode == 26 ? !eq : opcode == 25 ? lt : opcode == 24 ? (lt || eq) : opcode == 23 ? gt : (eq || gt);
  r[PC] = pc;
  r[IS_BEQ] = fbool(opcode == 21);
  r[IS_BNE] = fbool(opcode == 26);
  r[IS_BLTZ] = fbool(opcode == 25);
  r[IS_BLEZ] = fbool(opcode == 24);
  r[IS_BGTZ] = fbool(opcode == 23);
  r[IS_BGEZ] = fbool(opcode == 22);
  word(r + OP_A, a);
  word(r + OP_B, b);
  word(r + OP_C, c);
  r[A_LT_B] = fbool(lt);
  r[A_GT_B] = fbool(gt);
  word(r + NEXT_PC, next_pc);
  word(r + TARGET_PC, next_pc + c);
  word(r + NEXT_NEXT_PC, next_next_pc);
  range_checker(r + NEXT_PC_RC, next_pc);
  range_checker(r + NEXT_NEXT_PC_RC, next_next_pc);
  r[IS_BRANCHING] = fbool(branching);
}

// MovCond chip. MovCondEvent (events/instr.rs:286-302): pc, next_pc, opcode, a, b, c, prev_a — the AluEvent slots
// (pc, next_pc, opcode, hi, a, b, c) therefore hold a in `hi`, b in `a`, c in `b` and prev_a in `c`.
template <> __device__ __forceinline__ void event_row<MOV_COND>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, OP_A = 2, PREV_A = 6, OP_B = 10, OP_C = 14, C_EQ_0 = 18, IS_MNE = 29, IS_MEQ = 30, IS_WSBH = 31 };
  const uint32_t a = e.hi, b = e.a, c = e.b, prev_a = e.c;
  r[PC] = e.pc;
  r[NEXT_PC] = e.next_pc;
  word(r + OP_A, a);
  word(r + PREV_A, prev_a);
  word(r + OP_B, b);
  word(r + OP_C, c);
  uint32_t res[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {  // IsZeroOperation per byte: (inverse, result)
    const uint32_t byte = (c >> (8 * i)) & 0xff;
    r[C_EQ_0 + 2 * i] = small_inverse(byte);
    res[i] = r[C_EQ_0 + 2 * i + 1] = fbool(byte == 0);
  }
  r[C_EQ_0 + 8] = res[0] & res[1];
  r[C_EQ_0 + 9] = res[2] & res[3];
  r[C_EQ_0 + 10] = fbool(c == 0);
  r[IS_MNE] = fbool(e.opcode == 51);
  r[IS_MEQ] = fbool(e.opcode == 50);
  r[IS_WSBH] = fbool(e.opcode == 52);
}

// Mul chip: CompAluEvent (crates/core/executor/src/events/instr.rs:50-73) as sixteen words: shard, clk, pc, next_pc, opcode,
// hi, a, b, c, hi_record {value, shard, timestamp, prev_value, prev_shard, prev_timestamp}, hi_record_is_real.
// Columns alu/mul/mod.rs:79-141; row alu/mul/mod.rs:221-337; the HI access columns (prev_value, value, prev_shard,
// prev_clk, compare_clk, diff_16bit_limb, diff_8bit_limb) memory/consistency/trace.rs:43-100.
namespace mulcols {
enum { PC = 0, NEXT_PC = 1, HI = 2, A = 6, B = 10, C = 14, CARRY = 18, PRODUCT = 26, MSB_B = 34, MSB_C = 35, B_SIGN_EXTEND = 36,
       C_SIGN_EXTEND = 37, IS_MUL = 38, IS_MULT = 39, IS_MULTU = 40, IS_REAL = 41, OP_HI_ACCESS = 42, HI_RECORD_IS_REAL = 55,
       SHARD = 56, CLK = 57 };
}
__device__ __forceinline__ void mul_row(const uint32_t* p, uint32_t* r) {
  using namespace mulcols;
  const uint32_t shard = p[0], clk = p[1], opcode = p[4] & 0xff, hi = p[5], a = p[6], b = p[7], c = p[8];
  const bool real_hi = (p[15] & 0xff) != 0;
  r[PC] = p[2];
  r[NEXT_PC] = p[3];
  r[HI_RECORD_IS_REAL] = fbool(real_hi);
  if (real_hi) {
    const uint32_t value = p[9], rshard = p[10], ts = p[11], prev_value = p[12], prev_shard = p[13], prev_ts = p[14];
    uint32_t* m = r + OP_HI_ACCESS;
    word(m, prev_value);
    word(m + 4, value);
    m[8] = prev_shard;
    m[9] = prev_ts;
    const bool use_clk = prev_shard == rshard;
    m[10] = fbool(use_clk);
    const uint32_t diff_minus_one = (use_clk ? ts : rshard) - (use_clk ? prev_ts : prev_shard) - 1u;
    m[11] = diff_minus_one & 0xffff;
    m[12] = (diff_minus_one >> 16) & 0xff;
    r[SHARD] = shard;
    r[CLK] = clk;
  }
  const uint32_t b_msb = b >> 31, c_msb = c >> 31;
  const bool bse = opcode == 3 && b_msb, cse = opcode == 3 && c_msb;   // MULT sign-extends negative operands
  r[MSB_B] = b_msb;
  r[MSB_C] = c_msb;
  r[B_SIGN_EXTEND] = fbool(bse);
  r[C_SIGN_EXTEND] = fbool(cse);
  // the low 64 bits of the (sign-extended) product, byte by byte, with the schoolbook carries of the reference:
  // column k of the uncarried product is sum_{i+j=k} b_i c_j, at most 8 * 255^2 < 2^19
  uint32_t bb[8], cc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    bb[i] = i < 4 ? (b >> (8 * i)) & 0xff : (bse ? 0xffu : 0u);
    cc[i] = i < 4 ? (c >> (8 * i)) & 0xff : (cse ? 0xffu : 0u);
    // The bytes go through an empty asm so that the compiler sees eight unrelated values: left to itself (ROCm 7.2
    // clang, gfx950) it fuses the column sums into v_perm_b32 + v_dot4_u32_u8 and gets column 1 wrong
    // (tests/test_tracegen.py::test_gpu_mul_tracegen_matches_oracle caught it).
    asm volatile("" : "+v"(bb[i]));
    asm volatile("" : "+v"(cc[i]));
  }
  uint32_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t m = carry;
#pragma unroll
    for (int i = 0; i <= k; i++) m += bb[i] * cc[k - i];
    carry = m >> 8;
    r[CARRY + k] = carry;
    r[PRODUCT + k] = m & 0xff;
  }
  word(r + HI, hi);
  word(r + A, a);
  word(r + B, b);
  word(r + C, c);
  r[IS_REAL] = 1;
  r[IS_MUL] = fbool(opcode == 2);
  r[IS_MULT] = fbool(opcode == 3);
  r[IS_MULTU] = fbool(opcode == 4);
}

// DivRem chip: CompAluEvents; columns alu/divrem/mod.rs:106-202, row :224-381. IsZeroWordOperation = per byte (inverse,
// result), lower / upper half, result (operations/is_zero_word.rs:25-38); IsEqualWordOperation runs it on the field
// differences of the bytes (is_equal_word.rs:15-28). Canonical values throughout; alu_rows converts on the store.
namespace divcols {
enum { PC = 0, NEXT_PC = 1, B = 2, C = 6, QUOTIENT = 10, REMAINDER = 14, ABS_REMAINDER = 18, ABS_C = 22, MAX_ABS_C_OR_1 = 26,
       C_TIMES_QUOTIENT = 30, CARRY = 38, IS_C_0 = 46, IS_DIV = 57, IS_DIVU = 58, IS_MOD = 59, IS_MODU = 60, IS_OVERFLOW = 61,
       IS_OVERFLOW_B = 62, IS_OVERFLOW_C = 73, MSB_B = 84, MSB_REM = 85, MSB_C = 86, B_NEG = 87, REM_NEG = 88, C_NEG = 89,
       REMAINDER_CHECK_MULTIPLICITY = 90, OP_HI_ACCESS = 91, SHARD = 104, CLK = 105 };
}
// bytes of `a` minus bytes of `b` as field elements -> the eleven IsZeroWordOperation columns
__device__ __forceinline__ void is_equal_word_cols(uint32_t a, uint32_t b, uint32_t* r) {
  uint32_t z[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t x = (a >> (8 * i)) & 0xff, y = (b >> (8 * i)) & 0xff;
    const uint32_t diff = x >= y ? x - y : x + kb::P - y;   // canonical field difference
    z[i] = fbool(diff == 0);
    r[2 * i] = small_inverse(diff);
    r[2 * i + 1] = z[i];
  }
  r[8] = z[0] & z[1];
  r[9] = z[2] & z[3];
  r[10] = z[0] & z[1] & z[2] & z[3];
}
__device__ __forceinline__ void memory_write_cols(const uint32_t* rec, uint32_t* m) {   // rec: the six words of a MemoryWriteRecord
  const uint32_t value = rec[0], rshard = rec[1], ts = rec[2], prev_value = rec[3], prev_shard = rec[4], prev_ts = rec[5];
  word(m, prev_value);
  word(m + 4, value);
  m[8] = prev_shard;
  m[9] = prev_ts;
  const bool use_clk = prev_shard == rshard;
  m[10] = fbool(use_clk);
  const uint32_t diff_minus_one = (use_clk ? ts : rshard) - (use_clk ? prev_ts : prev_shard) - 1u;
  m[11] = diff_minus_one & 0xffff;
  m[12] = (diff_minus_one >> 16) & 0xff;
}
__device__ __forceinline__ void divrem_row(const uint32_t* p, uint32_t* r) {
  using namespace divcols;
  const uint32_t shard = p[0], clk = p[1], opcode = p[4] & 0xff, b = p[7], c = p[8];
  const bool is_signed = opcode == 5 || opcode == 7, is_div = opcode == 5 || opcode == 6;
  word(r + B, b);
  word(r + C, c);
  r[PC] = p[2];
  r[NEXT_PC] = p[3];
  r[IS_DIV] = fbool(opcode == 5);
  r[IS_DIVU] = fbool(opcode == 6);
  r[IS_MOD] = fbool(opcode == 7);
  r[IS_MODU] = fbool(opcode == 8);
  is_equal_word_cols(c, 0, r + IS_C_0);
  if (is_div) {   // DIV / DIVU always write HI (alu/divrem/mod.rs:246-255)
    memory_write_cols(p + 9, r + OP_HI_ACCESS);
    r[SHARD] = shard;
    r[CLK] = clk;
  }
  // get_quotient_and_remainder (crates/core/executor/src/utils.rs:33-43)
  uint32_t quotient, remainder;
  if (c == 0) {
    quotient = 0xffffffffu;
    remainder = b;
  } else if (is_signed) {
    if (b == 0x80000000u && c == 0xffffffffu) {   // wrapping_div / wrapping_rem
      quotient = 0x80000000u;
      remainder = 0;
    } else {
      quotient = (uint32_t)((int32_t)b / (int32_t)c);
      remainder = (uint32_t)((int32_t)b % (int32_t)c);
    }
  } else {
    quotient = b / c;
    remainder = b % c;
  }
  word(r + QUOTIENT, quotient);
  word(r + REMAINDER, remainder);
  r[MSB_REM] = remainder >> 31;
  r[MSB_B] = b >> 31;
  r[MSB_C] = c >> 31;
  is_equal_word_cols(b, 0x80000000u, r + IS_OVERFLOW_B);
  is_equal_word_cols(c, 0xffffffffu, r + IS_OVERFLOW_C);
  uint32_t abs_rem = remainder, abs_c = c;
  if (is_signed) {
    r[REM_NEG] = r[MSB_REM];
    r[B_NEG] = r[MSB_B];
    r[C_NEG] = r[MSB_C];
    r[IS_OVERFLOW] = fbool(b == 0x80000000u && c == 0xffffffffu);
    abs_rem = (remainder >> 31) ? 0u - remainder : remainder;
    abs_c = (c >> 31) ? 0u - c : c;
  }
  word(r + ABS_REMAINDER, abs_rem);
  word(r + ABS_C, abs_c);
  word(r + MAX_ABS_C_OR_1, abs_c > 1 ? abs_c : 1u);
  r[REMAINDER_CHECK_MULTIPLICITY] = fbool(c != 0);
  const uint64_t ctq = is_signed ? (uint64_t)((int64_t)(int32_t)quotient * (int64_t)(int32_t)c) : (uint64_t)quotient * c;
  const uint64_t rem64 = is_signed ? (uint64_t)(int64_t)(int32_t)remainder : (uint64_t)remainder;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t q = (uint32_t)(ctq >> (8 * i)) & 0xff;
    r[C_TIMES_QUOTIENT + i] = q;
    carry = (q + ((uint32_t)(rem64 >> (8 * i)) & 0xff) + carry) >> 8;
    r[CARRY + i] = carry;
  }
}

// MemoryInstructions chip: MemInstrEvents of sixteen words (crates/core/executor/src/events/instr.rs:108-136): shard, clk, pc,
// next_pc, opcode, a, b, c, mem_access tag (Read 0 / Write 1), six record words, prev_a_val. Columns
// memory/instructions/columns.rs:12-117; row trace.rs:100-262.
namespace memcols {
enum { PC = 0, NEXT_PC = 1, SHARD = 2, CLK = 3, OP_A = 4, OP_B = 8, OP_C = 12, IS_LB = 16, ADDR_WORD = 30, ADDR_ALIGNED = 34, ADDR_LS_TWO_BITS = 35,
       LS_IS_ONE = 36, LS_IS_TWO = 37, LS_IS_THREE = 38, ADDR_RC = 39, MEMORY_ACCESS = 53, PREV_A_VAL = 66, UNSIGNED_MEM_VAL = 70,
       MOST_SIG_BIT = 74, MOST_SIG_BYTE = 75, MEM_VALUE_IS_NEG = 76, MOST_SIG_BYTES_ZERO = 77 };
}
__device__ __forceinline__ void memory_instr_row(const uint32_t* p, uint32_t* r) {
  using namespace memcols;
  const uint32_t o = p[4] & 0xff, b = p[6], c = p[7], prev_a = p[15];
  r[SHARD] = p[0];
  r[CLK] = p[1];
  r[PC] = p[2];
  r[NEXT_PC] = p[3];
  word(r + OP_A, p[5]);
  word(r + OP_B, b);
  word(r + OP_C, c);
  const uint32_t* rec = p + 9;
  const uint32_t mem_value = rec[0];
  uint32_t* m = r + MEMORY_ACCESS;   // prev_value(4), then MemoryAccessCols
  word(m + 4, mem_value);
  const bool is_write = p[8] == 1;
  word(m, is_write ? rec[3] : mem_value);
  const uint32_t rshard = rec[1], ts = rec[2], prev_shard = is_write ? rec[4] : rec[3], prev_ts = is_write ? rec[5] : rec[4];
  m[8] = prev_shard;
  m[9] = prev_ts;
  const bool use_clk = prev_shard == rshard;
  m[10] = fbool(use_clk);
  const uint32_t diff_minus_one = (use_clk ? ts : rshard) - (use_clk ? prev_ts : prev_shard) - 1u;
  m[11] = diff_minus_one & 0xffff;
  m[12] = (diff_minus_one >> 16) & 0xff;
  word(r + PREV_A_VAL, prev_a);
  const uint32_t addr = b + c, ls = addr & 3;
  word(r + ADDR_WORD, addr);
  range_checker(r + ADDR_RC, addr);
  r[ADDR_ALIGNED] = addr & ~3u;
  r[ADDR_LS_TWO_BITS] = ls;
  r[LS_IS_ONE] = fbool(ls == 1);
  r[LS_IS_TWO] = fbool(ls == 2);
  r[LS_IS_THREE] = fbool(ls == 3);
  if (o <= 38) {   // the eight loads
    uint32_t u = mem_value;                                              // LW, LL
    if (o == 31 || o == 32) u = (mem_value >> (8 * ls)) & 0xff;          // LB, LBU
    else if (o == 33 || o == 34) u = (ls >> 1) ? mem_value >> 16 : mem_value & 0xffff;   // LH, LHU
    else if (o == 36) { const uint32_t sh = 24 - 8 * ls; u = (prev_a & ~(0xffffffffu << sh)) | (mem_value << sh); }   // LWL
    else if (o == 37) { const uint32_t sh = 8 * ls; u = (prev_a & ~(0xffffffffu >> sh)) | (mem_value >> sh); }         // LWR
    word(r + UNSIGNED_MEM_VAL, u);
    if (o == 31 || o == 33) {
      const uint32_t byte = o == 31 ? u & 0xff : (u >> 8) & 0xff;
      r[MEM_VALUE_IS_NEG] = byte >> 7;
      r[MOST_SIG_BYTE] = byte;
      r[MOST_SIG_BIT] = byte >> 7;
    }
  }
#pragma unroll
  for (int i = 0; i < 14; i++) r[IS_LB + i] = fbool(o == 31u + i);
  const uint32_t upper = ((addr >> 8) & 0xff) + ((addr >> 16) & 0xff) + (addr >> 24);
  r[MOST_SIG_BYTES_ZERO] = small_inverse(upper);
  r[MOST_SIG_BYTES_ZERO + 1] = fbool(upper == 0);
}

// SyscallInstrs chip: SyscallEvents of fourteen words (crates/core/executor/src/events/syscall.rs:7-29): pc, next_pc, shard, clk,
// a_record (value, shard, timestamp, prev_value, prev_shard, prev_timestamp), a_record_is_real, syscall_id, arg1, arg2. Columns
// syscall/instructions/columns.rs:9-58; row trace.rs:88-176 (six IsZeroOperations of the syscall id's differences: field inverses).
namespace syscols {
enum { PC = 0, NEXT_PC = 1, SHARD = 2, CLK = 3, NUM_EXTRA_CYCLES = 4, IS_HALT = 5, IS_SYS_LINUX = 6, IS_PREV_A1_ZERO = 7, SYSCALL_ID = 9, OP_A = 10,
       OP_B = 14, OP_C = 18, PREV_A = 22, IS_ENTER_UNCONSTRAINED = 26, IS_HINT_LEN = 28, IS_HALT_CHECK = 30, IS_EXIT_GROUP_CHECK = 32, IS_COMMIT = 34,
       IS_COMMIT_DEFERRED = 36, INDEX_BITMAP = 38, OP_B_RC = 46, OP_C_RC = 60, OP_B_CHECK = 74, OP_C_CHECK = 75, IS_REAL = 76 };
}
__device__ __forceinline__ void is_zero_cols(uint32_t id, uint32_t code, uint32_t* r) {   // IsZeroOperation of (id - code) in the field
  const uint32_t diff = id >= code ? id - code : id + kb::P - code;
  r[0] = small_inverse(diff);
  r[1] = fbool(diff == 0);
}
__device__ __forceinline__ void syscall_instr_row(const uint32_t* p, uint32_t* r) {
  using namespace syscols;
  const uint32_t value = p[4], prev = p[7], arg1 = p[12], arg2 = p[13];
  r[IS_REAL] = 1;
  r[PC] = p[0];
  r[NEXT_PC] = p[1];
  r[SHARD] = p[2];
  r[CLK] = p[3];
  word(r + OP_A, value);
  word(r + OP_B, arg1);
  word(r + OP_C, arg2);
  word(r + PREV_A, prev);
  r[SYSCALL_ID] = p[11];
  const uint32_t id = prev & 0xffff;
  r[NUM_EXTRA_CYCLES] = prev >> 24;
  const bool is_halt = id == 0 || id == 4246;
  r[IS_HALT] = fbool(is_halt);
  r[IS_SYS_LINUX] = fbool((prev & 0xff00) != 0);
  const bool send_to_table = ((prev >> 8) & 0xff) != 0 || ((prev >> 16) & 0xff) == 1;
  is_zero_cols((prev >> 8) & 0xff, 0, r + IS_PREV_A1_ZERO);
  is_zero_cols(id, 3, r + IS_ENTER_UNCONSTRAINED);
  is_zero_cols(id, 0xf0, r + IS_HINT_LEN);
  is_zero_cols(id, 0, r + IS_HALT_CHECK);
  is_zero_cols(id, 4246, r + IS_EXIT_GROUP_CHECK);
  is_zero_cols(id, 0x10, r + IS_COMMIT);
  is_zero_cols(id, 0x1a, r + IS_COMMIT_DEFERRED);
  if (id == 0x10 || id == 0x1a) {
#pragma unroll
    for (int i = 0; i < 8; i++) r[INDEX_BITMAP + i] = fbool(arg1 == (uint32_t)i);
  }
  if (send_to_table || is_halt) { r[OP_B_CHECK] = 1; range_checker(r + OP_B_RC, arg1); }
  if (send_to_table || id == 0x1a) { r[OP_C_CHECK] = 1; range_checker(r + OP_C_RC, arg2); }
}

// SyscallCore / SyscallPrecompile tables (syscall/chip.rs:71-107 columns, :211-276 rows; include/syscall.hpp:9-60): shard, clk, syscall_id,
// the half-words of arg1 and arg2, the half-words of the result and is_linux (Linux syscalls only: Core reads them off the V0 write
// record; Linux precompile events are not built, so Precompile rows carry zeros), is_real. Core's events arrive already filtered.
__device__ __forceinline__ void syscall_table_row(const uint32_t* p, uint32_t* r, bool precompile) {
  const uint32_t value = p[4], prev = p[7], arg1 = p[12], arg2 = p[13];
  r[0] = p[2]; r[1] = p[3]; r[2] = p[11];
  r[3] = arg1 & 0xffff; r[4] = arg1 >> 16; r[5] = arg2 & 0xffff; r[6] = arg2 >> 16;
  const bool is_linux = ((prev >> 8) & 0xff) != 0;      // Precompile: a Linux event's syscall event carries its code and v0 in the a_record
  r[7] = is_linux ? value & 0xffff : 0; r[8] = is_linux ? value >> 16 : 0;
  r[9] = fbool(is_linux);
  r[10] = 1;
}

// MiscInstrs chip: MiscEvents of fifteen words (crates/core/executor/src/events/instr.rs:239-261): shard, clk, pc, next_pc, opcode, a, b,
// c, prev_a, hi_record (six words). Columns misc/others/columns/*.rs — cells 20..63 are a union of SextCols / ExtCols / InsCols /
// MaddsubCols; row misc/others/trace.rs:90-273, AddDoubleOperation operations/adddouble.rs:19-78.
namespace misccols {
enum { SHARD = 0, CLK = 1, PC = 2, NEXT_PC = 3, OP_A = 4, PREV_A = 8, OP_B = 12, OP_C = 16, SPECIFIC = 20, IS_SEXT = 64, IS_INS = 65, IS_EXT = 66,
       IS_MADDU = 67, IS_MSUBU = 68, IS_MADD = 69, IS_MSUB = 70, IS_TEQ = 71 };
}
__device__ __forceinline__ void misc_instr_row(const uint32_t* p, uint32_t* r) {
  using namespace misccols;
  const uint32_t o = p[4] & 0xff, a = p[5], b = p[6], c = p[7], prev_a = p[8];
  r[SHARD] = p[0];
  r[CLK] = p[1];
  r[PC] = p[2];
  r[NEXT_PC] = p[3];
  word(r + OP_A, a);
  word(r + OP_B, b);
  word(r + OP_C, c);
  word(r + PREV_A, prev_a);
  r[IS_SEXT] = fbool(o == 55);
  r[IS_EXT] = fbool(o == 53);
  r[IS_INS] = fbool(o == 45);
  r[IS_MADDU] = fbool(o == 46);
  r[IS_MSUBU] = fbool(o == 47);
  r[IS_MADD] = fbool(o == 48);
  r[IS_MSUB] = fbool(o == 49);
  r[IS_TEQ] = fbool(o == 54);
  uint32_t* sp = r + SPECIFIC;
  if (o == 55 || o == 54) {          // SextCols: most_sig_bit, sig_byte, a_eq_b (11), is_seb, is_seh
    const bool half = c > 0;
    sp[half ? 14 : 13] = 1;
    sp[0] = half ? (b & 0xffff) >> 15 : (b & 0xff) >> 7;
    sp[1] = half ? (b >> 8) & 0xff : b & 0xff;
    is_equal_word_cols(a, b, sp + 2);
  } else if (o >= 46 && o <= 49) {   // MaddsubCols: mul_lo, mul_hi, value, value_hi, carry[7], src2_hi, src2_lo, op_hi_access (13)
    const bool is_sign = o >= 48, is_add = o == 46 || o == 48;
    const uint64_t multiply = is_sign ? (uint64_t)((int64_t)(int32_t)b * (int64_t)(int32_t)c) : (uint64_t)b * c;
    const uint32_t src2_lo = is_add ? prev_a : a, src2_hi = is_add ? p[12] : p[9];   // hi_record.prev_value / .value
    const uint64_t addend = ((uint64_t)src2_hi << 32) + src2_lo, expected = multiply + addend;
    word(sp + 0, (uint32_t)multiply);
    word(sp + 4, (uint32_t)(multiply >> 32));
    word(sp + 8, (uint32_t)expected);
    word(sp + 12, (uint32_t)(expected >> 32));
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      carry = ((uint32_t)((multiply >> (8 * i)) & 0xff) + (uint32_t)((addend >> (8 * i)) & 0xff) + carry) > 255 ? 1u : 0u;
      sp[16 + i] = carry;
    }
    word(sp + 23, src2_hi);
    word(sp + 27, src2_lo);
    memory_write_cols(p + 9, sp + 31);
  } else if (o == 53) {              // ExtCols: lsb, msbd, sll_val
    const uint32_t lsb = c & 0x1f, msbd = c >> 5;
    sp[0] = lsb;
    sp[1] = msbd;
    word(sp + 2, b << ((31 - lsb - msbd) & 31));
  } else {                           // INS — InsCols: lsb, msb, ror_val, srl1_val, srl_val, sll_val, add_val
    const uint32_t lsb = c & 0x1f, msb = c >> 5;
    const uint32_t ror = (prev_a >> lsb) | (prev_a << ((32 - lsb) & 31)), srl1 = ror >> 1, srl = srl1 >> ((msb - lsb) & 31);
    const uint32_t sll = b << ((31 - msb + lsb) & 31);
    sp[0] = lsb;
    sp[1] = msb;
    word(sp + 2, ror);
    word(sp + 6, srl1);
    word(sp + 10, srl);
    word(sp + 14, sll);
    word(sp + 18, srl + sll);
  }
}

// ---- byte lookups: the ALU chips' generate_dependencies, ByteChip::generate_trace and ByteChip::trace -----------------
// ByteOpcode, crates/core/executor/src/opcode.rs:195-216
enum : uint32_t { B_AND = 0, B_OR = 1, B_XOR = 2, B_SLL = 3, B_U8RANGE = 4, B_SHRCARRY = 5, B_LTU = 6, B_MSB = 7, B_U16RANGE = 8, B_NOR = 9 };
constexpr int NUM_BYTE_OPS = 10, BYTE_ROWS = 1 << 16, BYTE_PREP_COLS = 12;

// counts[op][b << 8 | c] += 1 (bytes/trace.rs:54-66; U16Range is indexed by its value and is not used by these chips).
// Real lookup streams are heavily skewed (range checks of zero upper bytes, carries of 0 / 1, sign-extension bytes):
// plain global atomics serialise on the hot counters (164 ms for 2^21 ShiftRight events). Each block therefore first
// combines its lookups in an LDS hash table (open addressing, key = counter index, claimed with atomicCAS) and then
// issues one global atomicAdd per distinct counter it holds — a hot counter costs one global atomic per block (2048
// rows); when the table is crowded (many distinct counters, so no hot spot) a lookup goes to its counter directly.
constexpr uint32_t HASH_EMPTY = 0xffffffffu;
struct LookupSink {
  uint32_t* keys;    // LDS, HASH_SLOTS entries
  uint32_t* vals;
  uint32_t mask;     // HASH_SLOTS - 1
  uint32_t* counts;  // global [op][row]
};
__device__ __forceinline__ void lookup(const LookupSink& k, uint32_t op, uint32_t b, uint32_t c) {
  const uint32_t idx = op * BYTE_ROWS + ((b & 0xff) << 8 | (c & 0xff));
  uint32_t slot = (idx * 2654435761u >> 12) & k.mask;
  for (int probe = 0; probe < 8; probe++) {
    const uint32_t prev = atomicCAS(k.keys + slot, HASH_EMPTY, idx);
    if (prev == HASH_EMPTY || prev == idx) {
      atomicAdd(k.vals + slot, 1u);
      return;
    }
    slot = (slot + 1) & k.mask;
  }
  atomicAdd(k.counts + idx, 1u);  // table crowded (many distinct counters, i.e. no hot spot): count directly
}
// The Program chip's fetch counters go through the same table under keys with bit 31 set (a loop's handful of program counters are as
// hot as any byte counter: 2^21 cycles of a six-instruction loop were six addresses taking 350 000 global atomics each — 6 ms of an
// otherwise 1.5 ms kernel).
constexpr uint32_t HASH_PROGRAM_KEY = 0x80000000u;
__device__ __forceinline__ void count_fetch(const LookupSink& k, uint32_t instr_index, uint32_t* program_counts) {
  const uint32_t key = HASH_PROGRAM_KEY | instr_index;
  uint32_t slot = (instr_index * 2654435761u >> 12) & k.mask;
  for (int probe = 0; probe < 8; probe++) {
    const uint32_t prev = atomicCAS(k.keys + slot, HASH_EMPTY, key);
    if (prev == HASH_EMPTY || prev == key) {
      atomicAdd(k.vals + slot, 1u);
      return;
    }
    slot = (slot + 1) & k.mask;
  }
  atomicAdd(program_counts + instr_index, 1u);
}
constexpr int HASH_SLOTS = 8192;   // 64 KiB of LDS: two blocks per CU
constexpr int TILES_PER_BLOCK = 8; // a block walks 8 x THREADS rows with one table: hot counters cost one global atomic per 2048 rows
// ByteRecord::add_u8_range_checks (crates/core/executor/src/events/byte.rs:72-82): bytes in pairs
__device__ __forceinline__ void range_checks(const LookupSink& counts, const uint32_t* bytes, int n) {
  for (int i = 0; i + 1 < n; i += 2) lookup(counts, B_U8RANGE, bytes[i], bytes[i + 1]);
  if (n & 1) lookup(counts, B_U8RANGE, bytes[n - 1], 0);
}

// The byte lookups each chip's event_to_row records (= the `send_byte`s of its AIR with multiplicity 1), read off
// the row: add_sub/mod.rs:176 -> operations/add.rs:48-53; bitwise/mod.rs:183-193; lt/mod.rs:227-241,268-274;
// sll/mod.rs:276-279; sr/mod.rs:258-265,309-316,334-337; clo_clz/mod.rs:123-130.
template <int CHIP> __device__ __forceinline__ void row_lookups(const uint32_t* r, uint32_t opcode, const LookupSink& counts);
template <> __device__ __forceinline__ void row_lookups<ADD_SUB>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  range_checks(counts, r + 9, 4);   // operand_1
  range_checks(counts, r + 13, 4);  // operand_2
  range_checks(counts, r + 2, 4);   // value
}
template <> __device__ __forceinline__ void row_lookups<BITWISE>(const uint32_t* r, uint32_t opcode, const LookupSink& counts) {
  const uint32_t op = opcode == AND ? B_AND : opcode == OR ? B_OR : opcode == XOR ? B_XOR : B_NOR;
#pragma unroll
  for (int i = 0; i < 4; i++) lookup(counts, op, r[6 + i], r[10 + i]);
}
template <> __device__ __forceinline__ void row_lookups<LT>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  lookup(counts, B_AND, r[8 + 3], 0x7f);
  lookup(counts, B_AND, r[12 + 3], 0x7f);
  lookup(counts, B_LTU, r[30], r[31]);
}
template <> __device__ __forceinline__ void row_lookups<SHIFT_LEFT>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  range_checks(counts, r + 31, 4);  // bit_shift_result
  range_checks(counts, r + 35, 4);  // bit_shift_result_carry
}
template <> __device__ __forceinline__ void row_lookups<SHIFT_RIGHT>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  lookup(counts, B_MSB, r[2 + 3], 0);
  uint32_t nbits = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) nbits += r[10 + i] * i;   // shift_by_n_bits is one-hot
#pragma unroll
  for (int i = 7; i >= 0; i--) lookup(counts, B_SHRCARRY, r[22 + i], nbits);
  range_checks(counts, r + 22, 8);
  range_checks(counts, r + 30, 8);
  range_checks(counts, r + 38, 8);
  range_checks(counts, r + 46, 8);
}

template <> __device__ __forceinline__ void row_lookups<CLO_CLZ>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  range_checks(counts, r + 10, 4);       // bb
  lookup(counts, B_LTU, r[2], 33);       // a < 33
}

template <> __device__ __forceinline__ void row_lookups<JUMP>(const uint32_t*, uint32_t, const LookupSink&) {}      // none
template <> __device__ __forceinline__ void row_lookups<MOV_COND>(const uint32_t*, uint32_t, const LookupSink&) {}  // none
template <> __device__ __forceinline__ void row_lookups<BRANCH>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  if (r[59]) return;               // taken branches record nothing (control_flow/branch/trace.rs:137-140)
  range_checks(counts, r + 1, 4);   // next_pc
  range_checks(counts, r + 23, 4);  // next_next_pc
}

// Mul: two MSB lookups, the u16 range checks of the carries (table row = the value), the u8 range checks of the product
// bytes, and for a real HI write the two limbs of the timestamp difference (alu/mul/mod.rs:268-283,331-334, trace.rs:95-99)
template <> __device__ __forceinline__ void row_lookups<MUL>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  using namespace mulcols;
  if (r[HI_RECORD_IS_REAL]) {
    lookup(counts, B_U16RANGE, r[OP_HI_ACCESS + 11] >> 8, r[OP_HI_ACCESS + 11]);
    lookup(counts, B_U8RANGE, 0, r[OP_HI_ACCESS + 12]);
  }
  lookup(counts, B_MSB, r[B + 3], 0);
  lookup(counts, B_MSB, r[C + 3], 0);
#pragma unroll
  for (int i = 0; i < 8; i++) lookup(counts, B_U16RANGE, r[CARRY + i] >> 8, r[CARRY + i]);
  range_checks(counts, r + PRODUCT, 8);
}

// DivRem: the HI write's two limbs (DIV / DIVU), three MSB lookups, U8Range pairs of quotient, remainder and c * quotient
// (alu/divrem/mod.rs:250-252,299-313,352-357)
template <> __device__ __forceinline__ void row_lookups<DIVREM>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  using namespace divcols;
  if (r[IS_DIV] | r[IS_DIVU]) {
    lookup(counts, B_U16RANGE, r[OP_HI_ACCESS + 11] >> 8, r[OP_HI_ACCESS + 11]);
    lookup(counts, B_U8RANGE, 0, r[OP_HI_ACCESS + 12]);
  }
  lookup(counts, B_MSB, r[B + 3], 0);
  lookup(counts, B_MSB, r[C + 3], 0);
  lookup(counts, B_MSB, r[REMAINDER + 3], 0);
  range_checks(counts, r + QUOTIENT, 4);
  range_checks(counts, r + REMAINDER, 4);
  range_checks(counts, r + C_TIMES_QUOTIENT, 8);
}

// MemoryInstructions: the access's two limbs, AND(addr byte 0, 3), MSB of a signed load, U8Range(addr bytes 1, 2), and
// LTU(35, addr byte 0) when the address fits one byte (memory/instructions/trace.rs:117,139-146,221-227,246-261)
template <> __device__ __forceinline__ void row_lookups<MEMORY_INSTRS>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  using namespace memcols;
  lookup(counts, B_U16RANGE, r[MEMORY_ACCESS + 11] >> 8, r[MEMORY_ACCESS + 11]);
  lookup(counts, B_U8RANGE, 0, r[MEMORY_ACCESS + 12]);
  lookup(counts, B_AND, r[ADDR_WORD], 3);
  if (r[IS_LB] | r[IS_LB + 2]) lookup(counts, B_MSB, r[MOST_SIG_BYTE], 0);
  lookup(counts, B_U8RANGE, r[ADDR_WORD + 1], r[ADDR_WORD + 2]);
  if (r[MOST_SIG_BYTES_ZERO + 1]) lookup(counts, B_LTU, 35, r[ADDR_WORD]);
}

// MiscInstrs (misc/others/trace.rs:144-152,192-207,226-273): SEXT: MSB; MADD*: U8Range pairs of the three 64-bit operands of the add,
// the HI access's two limbs; EXT: U8Range(lsb, msbd), LTU(lsb + msbd, 32); INS: U8Range(lsb, msb), LTU(lsb, msb + 1), LTU(msb, 32)
template <> __device__ __forceinline__ void row_lookups<MISC_INSTRS>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  using namespace misccols;
  const uint32_t* sp = r + SPECIFIC;
  if (r[IS_SEXT]) {
    lookup(counts, B_MSB, sp[1], 0);
  } else if (r[IS_MADDU] | r[IS_MSUBU] | r[IS_MADD] | r[IS_MSUB]) {
    range_checks(counts, sp + 0, 8);                    // the product: mul_lo, mul_hi
    range_checks(counts, sp + 27, 4);                   // the addend: src2_lo ...
    range_checks(counts, sp + 23, 4);                   // ... src2_hi
    range_checks(counts, sp + 8, 8);                    // the sum: value, value_hi
    lookup(counts, B_U16RANGE, sp[31 + 11] >> 8, sp[31 + 11]);
    lookup(counts, B_U8RANGE, 0, sp[31 + 12]);
  } else if (r[IS_EXT]) {
    lookup(counts, B_U8RANGE, sp[0], sp[1]);
    lookup(counts, B_LTU, sp[0] + sp[1], 32);
  } else if (r[IS_INS]) {
    lookup(counts, B_U8RANGE, sp[0], sp[1]);
    lookup(counts, B_LTU, sp[0], sp[1] + 1);
    lookup(counts, B_LTU, sp[1], 32);
  }
}
template <> __device__ __forceinline__ void row_lookups<SYSCALL_INSTRS>(const uint32_t*, uint32_t, const LookupSink&) {}   // none (trace.rs:88-176)
template <> __device__ __forceinline__ void row_lookups<SYSCALL_CORE>(const uint32_t* r, uint32_t, const LookupSink& counts) {   // chip.rs:178-183
  for (int k = 3; k < 7; k++) lookup(counts, B_U16RANGE, r[k] >> 8, r[k]);
}
template <> __device__ __forceinline__ void row_lookups<SYSCALL_PRECOMPILE>(const uint32_t* r, uint32_t, const LookupSink& counts) {
  for (int k = 3; k < 7; k++) lookup(counts, B_U16RANGE, r[k] >> 8, r[k]);
}

// events: n_events records of event_words(CHIP) words; out: column-major, `height` rows; grid = height / (tiles * THREADS), with
// tiles = 1 for the plain row writer (most blocks in flight) and TILES_PER_BLOCK when counting.
// counts (may be null): NUM_BYTE_OPS columns of BYTE_ROWS plain u32 counters; the byte lookups of every event row are
// added to them — the chip's generate_dependencies in the same pass that builds its trace.
template <int CHIP>
__device__ __forceinline__ void alu_rows_body(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                              uint32_t* __restrict__ out, uint32_t* counts, int tiles) {
  constexpr int W = chip_width(CHIP);
  extern __shared__ uint32_t hash_lds[];  // 2 * HASH_SLOTS words when counting, nothing otherwise (keeps the pure row writer at full occupancy)
  uint32_t* hkeys = hash_lds;
  uint32_t* hvals = hash_lds + HASH_SLOTS;
  const size_t row0 = (size_t)blockIdx.x * tiles * THREADS;
  const bool count = counts != nullptr && row0 < n_events;  // block-uniform
  if (count) {
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
    __syncthreads();
  }
  for (int t = 0; t < tiles; t++) {
    const size_t row = row0 + (size_t)t * THREADS + threadIdx.x;
    if (row >= height) break;
    uint32_t r[W];
#pragma unroll
    for (int c = 0; c < W; c++) r[c] = 0;
    if (row < n_events) {
      const uint32_t* p = events + row * event_words(CHIP);
      if constexpr (CHIP == MUL || CHIP == DIVREM || CHIP == MEMORY_INSTRS || CHIP == SYSCALL_INSTRS || CHIP == MISC_INSTRS || CHIP == SYSCALL_CORE ||
                    CHIP == SYSCALL_PRECOMPILE) {
        if constexpr (CHIP == MUL) mul_row(p, r); else if constexpr (CHIP == DIVREM) divrem_row(p, r);
        else if constexpr (CHIP == MEMORY_INSTRS) memory_instr_row(p, r); else if constexpr (CHIP == SYSCALL_INSTRS) syscall_instr_row(p, r);
        else if constexpr (CHIP == SYSCALL_CORE) syscall_table_row(p, r, false); else if constexpr (CHIP == SYSCALL_PRECOMPILE) syscall_table_row(p, r, true);
        else misc_instr_row(p, r);
        if (count) row_lookups<CHIP>(r, 0, LookupSink{hkeys, hvals, HASH_SLOTS - 1, counts});
      } else {
        AluEvent e{p[0], p[1], (CHIP == JUMP || CHIP == BRANCH) ? p[2] : (p[2] & 0xff), p[3], p[4], p[5], p[6]};
        event_row<CHIP>(e, r);
        if (count) row_lookups<CHIP>(r, e.opcode, LookupSink{hkeys, hvals, HASH_SLOTS - 1, counts});
      }
    } else {
      padding_row<CHIP>(r);
    }
#pragma unroll
    for (int c = 0; c < W; c++) out[(size_t)c * height + row] = kb::to_monty(r[c]);
  }
  if (count) {
    __syncthreads();
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x)
      if (hkeys[i] != HASH_EMPTY) atomicAdd(counts + hkeys[i], hvals[i]);
  }
}

template <int CHIP>
__global__ __launch_bounds__(THREADS) void alu_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                    uint32_t* __restrict__ out, uint32_t* counts, int tiles) {
  alu_rows_body<CHIP>(events, n_events, height, out, counts, tiles);
}

// The same with the event count in device memory: the rows of events a device-side pass selected (syscall_core_compact).
template <int CHIP>
__global__ __launch_bounds__(THREADS) void alu_rows_counted(const uint32_t* __restrict__ events, const uint32_t* __restrict__ n_events, size_t height,
                                                            uint32_t* __restrict__ out, uint32_t* counts, int tiles) {
  alu_rows_body<CHIP>(events, (size_t)*n_events, height, out, counts, tiles);
}

// SyscallChip::generate_trace's filter for the Core table (crates/core/machine/src/syscall/chip.rs:252-259) on the device, for events that
// are already in HBM: keeps, in order, the syscall events whose code (a_record.prev_value, word 7) has the send-to-table byte set or
// names a Linux syscall. One block; per pass of blockDim.x events an ordered prefix count (wave ballots + one LDS pass over the waves).
// *n_kept = how many were kept; *flags |= 1 when more were kept than `cap` rows hold (the caller's fixed height).
__global__ __launch_bounds__(1024) void syscall_core_compact(const uint32_t* __restrict__ events, size_t n_events, uint32_t* __restrict__ kept,
                                                             uint32_t* __restrict__ n_kept, size_t cap, uint32_t* __restrict__ flags) {
  constexpr int EW = event_words(SYSCALL_CORE);
  __shared__ uint32_t wave_count[16];
  __shared__ uint32_t base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (size_t i0 = 0; i0 < n_events; i0 += blockDim.x) {
    const size_t i = i0 + threadIdx.x;
    bool keep = false;
    if (i < n_events) {
      const uint32_t code = events[i * EW + 7];
      keep = ((code >> 16) & 0xff) == 1 || ((code >> 8) & 0xff) != 0;
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = base;
    for (int w = 0; w < wave; w++) before += wave_count[w];
    const size_t dst = (size_t)before + __popcll(m & ((1ull << lane) - 1));
    if (keep && dst < cap)
      for (int k = 0; k < EW; k++) kept[dst * EW + k] = events[i * EW + k];
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = base; for (int w = 0; w < waves; w++) t += wave_count[w]; base = t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (base > cap) { atomicOr(flags, 1u); *n_kept = (uint32_t)cap; }
    else *n_kept = base;
  }
}

// ---- Cpu chip (crates/core/machine/src/cpu/): CpuEventFfi records of 70 words (events/cpu.rs:46-106: clk, pc, next_pc,
// next_next_pc, a, a_record[12], b, b_record[12], c, c_record[12], hi[2], hi_record[12], memory_record[12], exit_code; an
// OptionMemoryRecordEnum is tag (Read 0, Write 1, None 2), a five-word read record, a six-word write record) and the
// program as InstructionFfi records of 6 words (instruction.rs:22-33: opcode u8 | op_a u8, op_b, op_c, imm_b u8 | imm_c u8,
// raw[2]). Columns cpu/columns/mod.rs:16-77; row cpu/trace.rs:117-257; padding rows :57-60.
constexpr int CPU_EVENT_WORDS = 70, INSTRUCTION_WORDS = 6, CPU_WIDTH = 67, PROGRAM_PREP_WIDTH = 14;
namespace cpucols {
enum { SHARD = 0, CLK_16 = 1, CLK_8 = 2, SHARD_TO_SEND = 3, CLK_TO_SEND = 4, PC = 5, NEXT_PC = 6, NEXT_NEXT_PC = 7, INSTRUCTION = 8,
       NUM_EXTRA_CYCLES = 21, IS_RW_A = 22, IS_CHECK_MEMORY = 23, IS_HALT = 24, IS_SEQUENTIAL = 25, OP_A_VALUE = 26, HI_OR_PREV_A = 30,
       OP_A_ACCESS = 34, OP_B_ACCESS = 47, OP_C_ACCESS = 56, IS_REAL = 65, OP_A_IMMUTABLE = 66 };
}
// the opcode predicates of crates/core/executor/src/instruction.rs:70-310 (opcode numbers: opcode.rs:26-90)
__device__ __forceinline__ bool op_is_branch(uint32_t o) { return o >= 21 && o <= 26; }
__device__ __forceinline__ bool op_is_jump(uint32_t o) { return o >= 27 && o <= 29; }
__device__ __forceinline__ bool op_is_memory(uint32_t o) { return o >= 31 && o <= 44; }
__device__ __forceinline__ bool op_is_maddsub(uint32_t o) { return o >= 46 && o <= 49; }
__device__ __forceinline__ void instruction_cols(const uint32_t* in, uint32_t* r) {   // opcode, op_a, op_b[4], op_c[4], op_a_0, imm_b, imm_c
  r[0] = in[0] & 0xff;
  r[1] = (in[0] >> 8) & 0xff;
  word(r + 2, in[1]);
  word(r + 6, in[2]);
  r[10] = fbool(((in[0] >> 8) & 0xff) == 0);
  r[11] = fbool((in[3] & 0xff) != 0);
  r[12] = fbool(((in[3] >> 8) & 0xff) != 0);
}
// MemoryAccessCols (value, prev_shard, prev_clk, compare_clk, diff limbs) at `m`, from (value, shard, timestamp, prev_shard, prev_timestamp)
__device__ __forceinline__ void memory_access_cols(uint32_t value, uint32_t shard, uint32_t ts, uint32_t prev_shard, uint32_t prev_ts, uint32_t* m) {
  word(m, value);
  m[4] = prev_shard;
  m[5] = prev_ts;
  const bool use_clk = prev_shard == shard;
  m[6] = fbool(use_clk);
  const uint32_t diff_minus_one = (use_clk ? ts : shard) - (use_clk ? prev_ts : prev_shard) - 1u;
  m[7] = diff_minus_one & 0xffff;
  m[8] = (diff_minus_one >> 16) & 0xff;
}
__device__ __forceinline__ void access_lookups(const uint32_t* m, const LookupSink& counts) {
  lookup(counts, B_U16RANGE, m[7] >> 8, m[7]);
  lookup(counts, B_U8RANGE, 0, m[8]);
}
// out: column-major, `height` rows; rows past n_events are the chip's padding rows. counts as in alu_rows.
__global__ __launch_bounds__(THREADS) void cpu_rows(const uint32_t* __restrict__ events, size_t n_events, const uint32_t* __restrict__ program,
                                                    size_t n_instr, uint32_t pc_base, uint32_t shard, size_t height,
                                                    uint32_t* __restrict__ out, uint32_t* counts, int tiles, int* bad_pc,
                                                    uint32_t* program_counts /* nullable: plain fetch counters, one per instruction */) {
  using namespace cpucols;
  extern __shared__ uint32_t hash_lds[];
  uint32_t* hkeys = hash_lds;
  uint32_t* hvals = hash_lds + HASH_SLOTS;
  const size_t row0 = (size_t)blockIdx.x * tiles * THREADS;
  const bool count = counts != nullptr && row0 < n_events;
  if (count) {
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
    __syncthreads();
  }
  const LookupSink sink{hkeys, hvals, HASH_SLOTS - 1, counts};
  for (int t = 0; t < tiles; t++) {
    const size_t row = row0 + (size_t)t * THREADS + threadIdx.x;
    if (row >= height) break;
    uint32_t r[CPU_WIDTH];
#pragma unroll
    for (int c = 0; c < CPU_WIDTH; c++) r[c] = 0;
    if (row < n_events) {
      const uint32_t* e = events + row * CPU_EVENT_WORDS;
      const uint32_t clk = e[0], pc = e[1];
      const size_t idx = (size_t)(pc - pc_base) >> 2;   // Program::fetch
      const bool in_program = pc >= pc_base && idx < n_instr;
      if (!in_program) atomicOr(bad_pc, 1);
      // the shard clock is range-checked as a 16-bit and an 8-bit limb (cpu/air/mod.rs:39,131-138): a clock of 2^24 or more would be
      // truncated into a trace that satisfies no verifier (the executor closes a shard long before: executor.rs:325,2423)
      if (clk >> 24) atomicOr(bad_pc, 2);
      if (in_program && program_counts) {   // ProgramChip::generate_trace in the same pass
        if (count && idx < 0x7fffffffu) count_fetch(LookupSink{hkeys, hvals, HASH_SLOTS - 1, counts}, (uint32_t)idx, program_counts);
        else atomicAdd(program_counts + idx, 1u);
      }
      const uint32_t* in = program + (in_program ? idx : 0) * INSTRUCTION_WORDS;
      const uint32_t o = in[0] & 0xff;
      r[SHARD] = shard;
      r[CLK_16] = clk & 0xffff;
      r[CLK_8] = (clk >> 16) & 0xff;
      r[PC] = pc;
      r[NEXT_PC] = e[2];
      r[NEXT_NEXT_PC] = e[3];
      instruction_cols(in, r + INSTRUCTION);
      const bool is_syscall = o == 30;
      const bool check_memory = o == 3 || o == 4 || o == 5 || o == 6 || is_syscall || op_is_maddsub(o) || op_is_memory(o);
      r[OP_A_IMMUTABLE] = fbool((o >= 39 && o <= 43) || op_is_branch(o) || o == 54);
      r[IS_RW_A] = fbool(is_syscall || o == 45 || op_is_maddsub(o) || o == 50 || o == 51 || op_is_memory(o));
      r[IS_CHECK_MEMORY] = fbool(check_memory);
      const uint32_t* a_rec = e + 5;
      const uint32_t* b_rec = e + 18;
      const uint32_t* c_rec = e + 31;
      word(r + OP_A_VALUE, e[4]);
      if ((e[43] & 0xff) == 0) word(r + HI_OR_PREV_A, e[44]);   // hi: Some
      word(r + OP_A_ACCESS + 4, e[4]);
      word(r + OP_B_ACCESS, e[17]);
      word(r + OP_C_ACCESS, e[30]);
      r[SHARD_TO_SEND] = check_memory ? shard : 0u;
      r[CLK_TO_SEND] = check_memory ? clk : 0u;
      const uint32_t a_tag = a_rec[0] & 0xff, b_tag = b_rec[0] & 0xff, c_tag = c_rec[0] & 0xff;
      if (a_tag == 1) {          // write: value, shard, timestamp, prev_value, prev_shard, prev_timestamp at words 6..11
        word(r + OP_A_ACCESS, a_rec[9]);
        memory_access_cols(a_rec[6], a_rec[7], a_rec[8], a_rec[10], a_rec[11], r + OP_A_ACCESS + 4);
      } else if (a_tag == 0) {   // read: value, shard, timestamp, prev_shard, prev_timestamp at words 1..5
        word(r + OP_A_ACCESS, a_rec[1]);
        memory_access_cols(a_rec[1], a_rec[2], a_rec[3], a_rec[4], a_rec[5], r + OP_A_ACCESS + 4);
      }
      if (b_tag == 0) memory_access_cols(b_rec[1], b_rec[2], b_rec[3], b_rec[4], b_rec[5], r + OP_B_ACCESS);
      if (c_tag == 0) memory_access_cols(c_rec[1], c_rec[2], c_rec[3], c_rec[4], c_rec[5], r + OP_C_ACCESS);
      bool is_halt = false;
      if (is_syscall) {   // HALT = 0, SYS_EXT_GROUP = 4246 in the low two bytes of the previous value of `a`
        const uint32_t id0 = r[OP_A_ACCESS], id1 = r[OP_A_ACCESS + 1];
        is_halt = (id0 == 0 && id1 == 0) || (id0 == (4246 & 0xff) && id1 == (4246 >> 8));
        r[IS_HALT] = fbool(is_halt);
        r[NUM_EXTRA_CYCLES] = r[OP_A_ACCESS + 3];
      }
      r[IS_SEQUENTIAL] = fbool(!is_halt && !op_is_branch(o) && !op_is_jump(o));
      r[IS_REAL] = 1;
      if (count) {   // cpu/trace.rs:203-221,240-255 and the three accesses
        lookup(sink, B_U16RANGE, (shard >> 8) & 0xff, shard);
        lookup(sink, B_U16RANGE, r[CLK_16] >> 8, r[CLK_16]);
        lookup(sink, B_U8RANGE, 0, r[CLK_8]);
        if (a_tag != 2) access_lookups(r + OP_A_ACCESS + 4, sink);
        if (b_tag == 0) access_lookups(r + OP_B_ACCESS, sink);
        if (c_tag == 0) access_lookups(r + OP_C_ACCESS, sink);
        lookup(sink, B_U8RANGE, r[OP_A_ACCESS + 4], r[OP_A_ACCESS + 5]);
        lookup(sink, B_U8RANGE, r[OP_A_ACCESS + 6], r[OP_A_ACCESS + 7]);
      }
    } else {
      r[INSTRUCTION + 11] = 1;   // imm_b
      r[INSTRUCTION + 12] = 1;   // imm_c
      r[IS_RW_A] = 1;
    }
#pragma unroll
    for (int c = 0; c < CPU_WIDTH; c++) out[(size_t)c * height + row] = kb::to_monty(r[c]);
  }
  if (count) {
    __syncthreads();
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) {
      const uint32_t key = hkeys[i];
      if (key == HASH_EMPTY) continue;
      if (key & HASH_PROGRAM_KEY) atomicAdd(program_counts + (key & ~HASH_PROGRAM_KEY), hvals[i]);
      else atomicAdd(counts + key, hvals[i]);
    }
  }
}

// ProgramChip::generate_preprocessed_trace (program/mod.rs:62-101): row i = (pc_base + 4 i, instruction columns), zero padding
__global__ void program_rows(const uint32_t* __restrict__ program, size_t n_instr, uint32_t pc_base, size_t height, uint32_t* __restrict__ out) {
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= height) return;
  uint32_t r[PROGRAM_PREP_WIDTH];
#pragma unroll
  for (int c = 0; c < PROGRAM_PREP_WIDTH; c++) r[c] = 0;
  if (row < n_instr) {
    r[0] = pc_base + 4u * (uint32_t)row;
    instruction_cols(program + row * INSTRUCTION_WORDS, r + 1);
  }
#pragma unroll
  for (int c = 0; c < PROGRAM_PREP_WIDTH; c++) out[(size_t)c * height + row] = kb::to_monty(r[c]);
}
// ProgramChip::generate_trace (program/mod.rs:113-146): how often each pc was fetched; out starts as plain zero counters
__global__ void program_count(const uint32_t* __restrict__ events, size_t n_events, size_t n_instr, uint32_t pc_base, uint32_t* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_events) return;
  const uint32_t pc = events[i * CPU_EVENT_WORDS + 1];
  const size_t idx = (size_t)(pc - pc_base) >> 2;
  if (pc >= pc_base && idx < n_instr) atomicAdd(out + idx, 1u);
}
__global__ void counts_to_field(uint32_t* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = kb::to_monty(out[i]);
}

// MemoryLocalChip::generate_trace (memory/local.rs:147-190): MemoryLocalEvents of seven words (addr, initial {shard,
// timestamp, value}, final {...}), four per row; one thread per entry writes its fourteen columns (addr, initial_shard,
// final_shard, initial_clk, final_clk, initial_value[4], final_value[4], is_real); entries past n_events are zero.
constexpr int MEMORY_LOCAL_ENTRIES = 4, MEMORY_LOCAL_ENTRY_COLS = 14, MEMORY_LOCAL_WIDTH = 56;
__global__ void memory_local_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // entry index; consecutive threads -> consecutive rows of one entry slot
  const size_t row = i % height, k = i / height;
  if (k >= MEMORY_LOCAL_ENTRIES) return;
  const size_t ev = row * MEMORY_LOCAL_ENTRIES + k;
  uint32_t r[MEMORY_LOCAL_ENTRY_COLS];
#pragma unroll
  for (int c = 0; c < MEMORY_LOCAL_ENTRY_COLS; c++) r[c] = 0;
  if (ev < n_events) {
    const uint32_t* e = events + ev * 7;
    r[0] = e[0]; r[1] = e[1]; r[2] = e[4]; r[3] = e[2]; r[4] = e[5];
    word(r + 5, e[3]);
    word(r + 9, e[6]);
    r[13] = 1;
  }
#pragma unroll
  for (int c = 0; c < MEMORY_LOCAL_ENTRY_COLS; c++) out[(k * MEMORY_LOCAL_ENTRY_COLS + c) * height + row] = kb::to_monty(r[c]);
}

// ---- recursion Poseidon2Wide chip, degree 3 (crates/recursion/core/src/chips/poseidon2_wide/): one thread runs one
// permutation and writes every intermediate the AIR constrains as it goes — 313 columns: external_rounds_state[8][16],
// internal_rounds_state[16], internal_rounds_s0[12], output_state[16], external_rounds_sbox_state[8][16],
// internal_rounds_sbox_state[13] (columns/permutation.rs:20-36, rows trace.rs:277-420). Events are 32 Montgomery words
// (Poseidon2Event: input[16], output[16]); rows past n_events are the permutation of the zero state (:99-105). All values
// stay in Montgomery form, which is what the matrix stores. rc_ext / rc_int / diag: the constant-memory tables of the hashing
// kernels (poseidon2.cuh; the round constants are kept there minus p, for the folded S-box).
constexpr int POSEIDON2_WIDE_WIDTH = 313;
__device__ __forceinline__ void wide_external_layer(uint32_t s[16]) {   // mds_light_permutation, chips/poseidon2_wide/mod.rs:45-71
#pragma unroll
  for (int j = 0; j < 16; j += 4) {
    const uint32_t t01 = kb::add(s[j], s[j + 1]), t23 = kb::add(s[j + 2], s[j + 3]), t0123 = kb::add(t01, t23);
    const uint32_t t01123 = kb::add(t0123, s[j + 1]), t01233 = kb::add(t0123, s[j + 3]);
    const uint32_t x0 = s[j], x2 = s[j + 2];
    s[j + 3] = kb::add(t01233, kb::add(x0, x0));
    s[j + 1] = kb::add(t01123, kb::add(x2, x2));
    s[j] = kb::add(t01123, t01);
    s[j + 2] = kb::add(t01233, t23);
  }
  uint32_t sums[4];
#pragma unroll
  for (int k = 0; k < 4; k++) sums[k] = kb::add(kb::add(s[k], s[4 + k]), kb::add(s[8 + k], s[12 + k]));
#pragma unroll
  for (int j = 0; j < 16; j++) s[j] = kb::add(s[j], sums[j & 3]);
}
__device__ __forceinline__ uint32_t wide_sbox(uint32_t x) { return kb::mul(kb::mul(x, x), x); }
__global__ __launch_bounds__(THREADS) void poseidon2_wide_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                               uint32_t* __restrict__ out) {
  enum { EXT_STATE = 0, INT_STATE = 128, INT_S0 = 144, OUTPUT = 156, EXT_SBOX = 172, INT_SBOX = 300 };
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= height) return;
  uint32_t s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = row < n_events ? events[row * 32 + i] : 0u;
  auto put = [&](int col, uint32_t v) { out[(size_t)col * height + row] = v; };
#pragma unroll
  for (int i = 0; i < 16; i++) put(EXT_STATE + i, s[i]);
  wide_external_layer(s);
  for (int rd = 0; rd < 8; rd++) {
    if (rd == 4) {
      for (int r = 0; r < 13; r++) {
        s[0] = wide_sbox(kb::add(s[0], p2::d_rc_int[r] + kb::P));
        put(INT_SBOX + r, s[0]);
        uint32_t sum = s[0];
#pragma unroll
        for (int i = 1; i < 16; i++) sum = kb::add(sum, s[i]);
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = kb::add(kb::mul(s[i], p2::d_diag[i]), sum);
        if (r < 12) put(INT_S0 + r, s[0]);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) put(EXT_STATE + 64 + i, s[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      s[i] = wide_sbox(kb::add(s[i], p2::d_rc_ext[rd][i] + kb::P));
      put(EXT_SBOX + 16 * rd + i, s[i]);
    }
    wide_external_layer(s);
    const int next = rd == 3 ? INT_STATE : rd == 7 ? OUTPUT : EXT_STATE + 16 * (rd + 1);
#pragma unroll
    for (int i = 0; i < 16; i++) put(next + i, s[i]);
  }
}

// ---- MemoryGlobalInit / MemoryGlobalFinalize (memory/global.rs:113-185, include/memory_global.hpp:9-43): events (addr, value, shard, timestamp)
// sorted by address; one thread per row. Row i compares its address with row i - 1's (row 0: with the previous shard's last address,
// `previous_addr`, when that is not zero): the flag of the most significant differing bit. 111 columns.
constexpr int MEMORY_GLOBAL_WIDTH = 111;
__global__ __launch_bounds__(THREADS) void memory_global_rows(const uint32_t* __restrict__ events, size_t n_events, uint32_t previous_addr, size_t height,
                                                              uint32_t* __restrict__ out, int* __restrict__ bad) {
  enum { SHARD = 0, TIMESTAMP = 1, ADDR = 2, LT = 3, ADDR_BITS = 35, AND_DECOMP = 67, VALUE = 73, IS_REAL = 105, IS_NEXT_COMP = 106, IS_PREV_ADDR_ZERO = 107,
         IS_FIRST_COMP = 109, IS_LAST_ADDR = 110 };
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= height) return;
  auto put = [&](int col, uint32_t canonical) { out[(size_t)col * height + row] = kb::to_monty(canonical); };
  if (row >= n_events) {
    for (int c = 0; c < MEMORY_GLOBAL_WIDTH; c++) out[(size_t)c * height + row] = 0;
    return;
  }
  const uint32_t addr = events[4 * row], value = events[4 * row + 1];
  put(SHARD, events[4 * row + 2]);
  put(TIMESTAMP, events[4 * row + 3]);
  put(ADDR, addr);   // an address is below p (the chip range-checks it; to_monty reduces a larger one like from_canonical would)
  const bool compare = row > 0 || previous_addr != 0;
  const uint32_t before = row > 0 ? events[4 * (row - 1)] : previous_addr;
  if (compare && !(before < addr)) *bad = 1;
  const int first_lt = compare && before < addr ? 31 - __clz(before ^ addr) : -1;   // the most significant bit where they differ: there before has 0, addr 1
  for (int k = 0; k < 32; k++) {
    out[(size_t)(LT + k) * height + row] = k == first_lt ? kb::ONE : 0u;
    out[(size_t)(ADDR_BITS + k) * height + row] = (addr >> k) & 1 ? kb::ONE : 0u;
    out[(size_t)(VALUE + k) * height + row] = (value >> k) & 1 ? kb::ONE : 0u;
  }
  uint32_t prod = (addr >> 24) & (addr >> 25) & 1;
  put(AND_DECOMP, prod);
  for (int k = 0; k < 5; k++) { prod &= (addr >> (26 + k)) & 1; put(AND_DECOMP + 1 + k, prod); }
  put(IS_REAL, 1);
  put(IS_NEXT_COMP, row > 0);
  if (row == 0) {
    out[(size_t)IS_PREV_ADDR_ZERO * height] = previous_addr ? kb::inv(kb::to_monty(previous_addr)) : 0u;
    put(IS_PREV_ADDR_ZERO + 1, previous_addr == 0);
  } else {
    put(IS_PREV_ADDR_ZERO, 0); put(IS_PREV_ADDR_ZERO + 1, 0);
  }
  put(IS_FIRST_COMP, row == 0 && previous_addr != 0);
  put(IS_LAST_ADDR, row == n_events - 1);
}

// ---- Poseidon2Permute precompile (syscall/precompiles/poseidon2/: columns.rs:9-27, trace.rs:31-128): flattened events of 99 words (shard, clk,
// state_addr, sixteen MemoryWriteRecords); one thread per row runs the permutation on the records' previous values and writes every
// intermediate (the 313 columns of poseidon2_wide_rows above), then the memory columns and the range checkers of the pre- and post-state
// words; the two byte lookups of each of the sixteen accesses go to the block's LDS table. Padding rows: the permutation of the zero state.
constexpr int POSEIDON2_PERMUTE_WIDTH = 973, POSEIDON2_PERMUTE_EVENT_WORDS = 99;
__global__ __launch_bounds__(THREADS) void poseidon2_permute_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                                  uint32_t* __restrict__ out, uint32_t* counts, int* __restrict__ bad) {
  enum { EXT_STATE = 0, INT_STATE = 128, INT_S0 = 144, OUTPUT = 156, EXT_SBOX = 172, INT_SBOX = 300, SHARD = 313, CLK = 314, STATE_ADDR = 315,
         STATE_MEM = 316, PRE_RC = 524, POST_RC = 748, IS_REAL = 972 };
  extern __shared__ uint32_t hash_lds[];
  uint32_t* hkeys = hash_lds;
  uint32_t* hvals = hash_lds + HASH_SLOTS;
  const bool count = counts != nullptr;
  if (count) {
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
    __syncthreads();
  }
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * POSEIDON2_PERMUTE_EVENT_WORDS;
    auto put = [&](int col, uint32_t monty) { out[(size_t)col * height + row] = monty; };
    uint32_t s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = real ? kb::to_monty(e[3 + 6 * i + 3]) : 0u;   // the record's prev_value
#pragma unroll
    for (int i = 0; i < 16; i++) put(EXT_STATE + i, s[i]);
    wide_external_layer(s);
    for (int rd = 0; rd < 8; rd++) {
      if (rd == 4) {
        for (int r = 0; r < 13; r++) {
          s[0] = wide_sbox(kb::add(s[0], p2::d_rc_int[r] + kb::P));
          put(INT_SBOX + r, s[0]);
          uint32_t sum = s[0];
#pragma unroll
          for (int i = 1; i < 16; i++) sum = kb::add(sum, s[i]);
#pragma unroll
          for (int i = 0; i < 16; i++) s[i] = kb::add(kb::mul(s[i], p2::d_diag[i]), sum);
          if (r < 12) put(INT_S0 + r, s[0]);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) put(EXT_STATE + 64 + i, s[i]);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        s[i] = wide_sbox(kb::add(s[i], p2::d_rc_ext[rd][i] + kb::P));
        put(EXT_SBOX + 16 * rd + i, s[i]);
      }
      wide_external_layer(s);
      const int next = rd == 3 ? INT_STATE : rd == 7 ? OUTPUT : EXT_STATE + 16 * (rd + 1);
#pragma unroll
      for (int i = 0; i < 16; i++) put(next + i, s[i]);
    }
    if (real) {
      put(SHARD, kb::to_monty(e[0])); put(CLK, kb::to_monty(e[1])); put(STATE_ADDR, kb::to_monty(e[2])); put(IS_REAL, kb::ONE);
      const LookupSink sink{hkeys, hvals, HASH_SLOTS - 1, counts};
      for (int i = 0; i < 16; i++) {
        const uint32_t* rec = e + 3 + 6 * i;
        if (rec[0] >= kb::P || rec[3] >= kb::P || kb::to_monty(rec[0]) != s[i]) *bad = 1;   // post-state = the permutation of the pre-state
        uint32_t m[13], rc[14];
        memory_write_cols(rec, m);
        for (int c = 0; c < 13; c++) put(STATE_MEM + 13 * i + c, kb::to_monty(m[c]));
        range_checker(rc, rec[3]);
        for (int c = 0; c < 14; c++) put(PRE_RC + 14 * i + c, kb::to_monty(rc[c]));
        range_checker(rc, rec[0]);
        for (int c = 0; c < 14; c++) put(POST_RC + 14 * i + c, kb::to_monty(rc[c]));
        if (count) { lookup(sink, B_U16RANGE, m[11] >> 8, m[11]); lookup(sink, B_U8RANGE, 0, m[12]); }
      }
    } else {
      for (int c = SHARD; c < POSEIDON2_PERMUTE_WIDTH; c++) put(c, 0u);
    }
  }
  if (count) {
    __syncthreads();
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x)
      if (hkeys[i] != HASH_EMPTY) atomicAdd(counts + hkeys[i], hvals[i]);
  }
}

// ---- KeccakSponge precompile (syscall/precompiles/keccak_sponge/: columns.rs:17-37, trace.rs:102-195): KeccakSpongeEvents cut into their
// 36-word blocks (337 words each, include/zkm_hip.h zkm_keccak_sponge_block), twenty-four rows per block. One thread per row: it runs
// keccak-f from the block's xored state up to its round (twelve rounds on average, a few hundred 64-bit operations each — small next to the
// 3531 cells it then stores), and writes the round's KeccakCols (the layout of p3-keccak-air: step flags, export, preimage and A in 16-bit
// limbs, the bits of C, C' and A', A'' in limbs, the bits of A''[0][0], A'''[0][0] in limbs) and the sponge's own columns, which are zero
// except on a block's first row (the block is read and xored in, the call is received) and last row (the state goes on or is written out).
// Rows past the last block are rounds of the permutation of the zero state, row i carrying round i mod 24 (trace.rs:79-93).
constexpr int KECCAK_SPONGE_WIDTH = 3531, KECCAK_SPONGE_BLOCK_WORDS = 337, NUM_KECCAK_COLS = 2633;
__constant__ uint64_t d_keccak_rc[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
template <int R> __device__ __forceinline__ uint64_t rotl64(uint64_t v) {
  if constexpr (R == 0) return v;
  else return (v << R) | (v >> (64 - R));
}
// rotation offsets r[x][y] of rho, as template arguments so that every rotation is by an immediate
template <int X, int Y> struct KeccakRot {
  static constexpr int T[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
  static constexpr int value = T[X][Y];
};
// B[x, y] = rot(A'[(x + 3y) mod 5, x], r[(x + 3y) mod 5][x]) with A' stored [y][x]
template <int X, int Y> __device__ __forceinline__ uint64_t keccak_b(const uint64_t* ap) {
  constexpr int XA = (X + 3 * Y) % 5;
  return rotl64<KeccakRot<XA, X>::value>(ap[5 * X + XA]);
}
template <int X, int Y> __device__ __forceinline__ void keccak_chi_lane(const uint64_t* ap, uint64_t* app) {
  app[5 * Y + X] = keccak_b<X, Y>(ap) ^ (~keccak_b<(X + 1) % 5, Y>(ap) & keccak_b<(X + 2) % 5, Y>(ap));
}
template <int Y> __device__ __forceinline__ void keccak_chi_row(const uint64_t* ap, uint64_t* app) {
  keccak_chi_lane<0, Y>(ap, app); keccak_chi_lane<1, Y>(ap, app); keccak_chi_lane<2, Y>(ap, app); keccak_chi_lane<3, Y>(ap, app); keccak_chi_lane<4, Y>(ap, app);
}
// one round on a[5 * y + x]: c = column parities, cp = C', ap = A' (after theta), app = A'' (after chi, before iota)
__device__ __forceinline__ void keccak_round_parts(const uint64_t* a, uint64_t* c, uint64_t* cp, uint64_t* ap, uint64_t* app) {
#pragma unroll
  for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[5 + x] ^ a[10 + x] ^ a[15 + x] ^ a[20 + x];
#pragma unroll
  for (int x = 0; x < 5; x++) {
    const uint64_t d = c[(x + 4) % 5] ^ rotl64<1>(c[(x + 1) % 5]);
    cp[x] = c[x] ^ d;
#pragma unroll
    for (int y = 0; y < 5; y++) ap[5 * y + x] = a[5 * y + x] ^ d;
  }
  keccak_chi_row<0>(ap, app); keccak_chi_row<1>(ap, app); keccak_chi_row<2>(ap, app); keccak_chi_row<3>(ap, app); keccak_chi_row<4>(ap, app);
}
__global__ __launch_bounds__(THREADS) void keccak_sponge_rows(const uint32_t* __restrict__ blocks, size_t n_blocks, size_t height,
                                                              uint32_t* __restrict__ out, uint32_t* counts, int* __restrict__ bad) {
  enum { STEP = 0, EXPORT = 24, PREIMAGE = 25, A = 125, C = 225, C_PRIME = 545, A_PRIME = 865, A_PP = 2465, A_PP_00_BITS = 2565, A_PPP_00 = 2629,
         BLOCK_MEM = 2633, SHARD = 2957, CLK = 2958, IS_REAL = 2959, READ_BLOCK = 2960, INPUT_ADDRESS = 2961, OUTPUT_ADDRESS = 2962, INPUT_LEN = 2963,
         ALREADY_ABSORBED = 2964, IS_ABSORBED = 2965, RECEIVE_SYSCALL = 2966, WRITE_OUTPUT = 2967, IS_FIRST = 2968, IS_FINAL = 2969, ORIGINAL_STATE = 2970,
         XORED_RATE = 3170, INPUT_LENGTH_MEM = 3314, OUTPUT_MEM = 3323 };
  enum { E_SHARD = 0, E_CLK = 1, E_INPUT_ADDR = 2, E_OUTPUT_ADDR = 3, E_INPUT_LEN = 4, E_BLOCK_INDEX = 5, E_XORED = 6, E_READS = 56, E_LEN_RECORD = 236,
         E_WRITES = 241 };
  extern __shared__ uint32_t hash_lds[];
  uint32_t* hkeys = hash_lds;
  uint32_t* hvals = hash_lds + HASH_SLOTS;
  const bool count = counts != nullptr;
  if (count) {
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
    __syncthreads();
  }
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const size_t blk = row / 24;
    const int round = (int)(row % 24);
    const bool real = blk < n_blocks;
    const uint32_t* e = blocks + blk * KECCAK_SPONGE_BLOCK_WORDS;
    auto put = [&](int col, uint32_t monty) { out[(size_t)col * height + row] = monty; };
    auto fail = [&](int why) { atomicMax(bad, 16 - why); };     // the lowest code is reported: 16 - *bad on the host
    auto put_bits = [&](int col, uint64_t v) {
      for (int z = 0; z < 64; z++) put(col + z, (v >> z) & 1 ? kb::ONE : 0u);
    };
    auto put_limbs = [&](int col, uint64_t v) {
#pragma unroll
      for (int l = 0; l < 4; l++) put(col + l, kb::to_monty((uint32_t)(v >> (16 * l)) & 0xffffu));
    };
    uint64_t a[25], c[5], cp[5], ap[25], app[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = real ? (uint64_t)e[E_XORED + 2 * i] | ((uint64_t)e[E_XORED + 2 * i + 1] << 32) : 0ull;
#pragma unroll
    for (int i = 0; i < 25; i++) put_limbs(PREIMAGE + 4 * i, a[i]);
    for (int r = 0; r < round; r++) {
      keccak_round_parts(a, c, cp, ap, app);
#pragma unroll
      for (int i = 0; i < 25; i++) a[i] = app[i];
      a[0] ^= d_keccak_rc[r];
    }
    keccak_round_parts(a, c, cp, ap, app);
    for (int i = 0; i < 24; i++) put(STEP + i, i == round ? kb::ONE : 0u);
    put(EXPORT, 0u);
#pragma unroll
    for (int i = 0; i < 25; i++) put_limbs(A + 4 * i, a[i]);
#pragma unroll
    for (int x = 0; x < 5; x++) { put_bits(C + 64 * x, c[x]); put_bits(C_PRIME + 64 * x, cp[x]); }
#pragma unroll
    for (int i = 0; i < 25; i++) put_bits(A_PRIME + 64 * i, ap[i]);
#pragma unroll
    for (int i = 0; i < 25; i++) put_limbs(A_PP + 4 * i, app[i]);
    put_bits(A_PP_00_BITS, app[0]);
    const uint64_t out00 = app[0] ^ d_keccak_rc[round];
    put_limbs(A_PPP_00, out00);
    // ---- the sponge's own columns
    if (!real) {
      for (int col = BLOCK_MEM; col < KECCAK_SPONGE_WIDTH; col++) put(col, 0u);
    } else {
      const uint32_t len = e[E_INPUT_LEN], index = e[E_BLOCK_INDEX];
      const uint32_t n_ev_blocks = len / 36;
      if (len == 0 || len % 36 != 0 || index >= n_ev_blocks) fail(1);
      const bool first = index == 0, final = index + 1 == n_ev_blocks;
      const LookupSink sink{hkeys, hvals, HASH_SLOTS - 1, counts};
      put(SHARD, kb::to_monty(e[E_SHARD])); put(CLK, kb::to_monty(e[E_CLK])); put(IS_REAL, kb::ONE);
      put(READ_BLOCK, round == 0 ? kb::ONE : 0u);
      put(INPUT_ADDRESS, kb::to_monty(e[E_INPUT_ADDR] + index * 144u));
      put(OUTPUT_ADDRESS, kb::to_monty(e[E_OUTPUT_ADDR]));
      put(INPUT_LEN, kb::to_monty(len));
      put(ALREADY_ABSORBED, kb::to_monty(36u * index));
      put(IS_ABSORBED, round == 23 && !final ? kb::ONE : 0u);
      put(RECEIVE_SYSCALL, first && round == 0 ? kb::ONE : 0u);
      put(WRITE_OUTPUT, final && round == 23 ? kb::ONE : 0u);
      put(IS_FIRST, first ? kb::ONE : 0u);
      put(IS_FINAL, final ? kb::ONE : 0u);
      for (int j = 0; j < 50; j++) {           // the state the block is absorbed into: xored ^ block on the rate part
        const uint32_t before = e[E_XORED + j] ^ (j < 36 ? e[E_READS + 5 * j] : 0u);
        if (first && before) fail(2);
        for (int k = 0; k < 4; k++) put(ORIGINAL_STATE + 4 * j + k, kb::to_monty((before >> (8 * k)) & 0xff));
      }
      if (round == 0) {
        for (int j = 0; j < 36; j++) {
          const uint32_t* rec = e + E_READS + 5 * j;
          uint32_t m[9];
          memory_access_cols(rec[0], rec[1], rec[2], rec[3], rec[4], m);
          for (int k = 0; k < 9; k++) put(BLOCK_MEM + 9 * j + k, kb::to_monty(m[k]));
          const uint32_t xored = e[E_XORED + j], before = xored ^ rec[0];
          for (int k = 0; k < 4; k++) put(XORED_RATE + 4 * j + k, kb::to_monty((xored >> (8 * k)) & 0xff));
          if (count) {
            access_lookups(m, sink);
            for (int k = 0; k < 4; k++) lookup(sink, B_XOR, before >> (8 * k), rec[0] >> (8 * k));
          }
        }
      } else {
        for (int col = BLOCK_MEM; col < SHARD; col++) put(col, 0u);
        for (int col = XORED_RATE; col < INPUT_LENGTH_MEM; col++) put(col, 0u);
      }
      if (first && round == 0) {
        const uint32_t* rec = e + E_LEN_RECORD;
        if (rec[0] != len) fail(3);
        uint32_t m[9];
        memory_access_cols(rec[0], rec[1], rec[2], rec[3], rec[4], m);
        for (int k = 0; k < 9; k++) put(INPUT_LENGTH_MEM + k, kb::to_monty(m[k]));
        if (count) access_lookups(m, sink);
      } else {
        for (int k = 0; k < 9; k++) put(INPUT_LENGTH_MEM + k, 0u);
      }
      if (final && round == 23) {
        uint64_t squeezed[8] = {out00, app[1], app[2], app[3], app[4], app[5], app[6], app[7]};
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const uint32_t* rec = e + E_WRITES + 6 * j;
          if (rec[0] != (uint32_t)(squeezed[j / 2] >> (32 * (j & 1)))) fail(4);
          uint32_t m[13];
          memory_write_cols(rec, m);
          for (int k = 0; k < 13; k++) put(OUTPUT_MEM + 13 * j + k, kb::to_monty(m[k]));
          if (count) { lookup(sink, B_U16RANGE, m[11] >> 8, m[11]); lookup(sink, B_U8RANGE, 0, m[12]); }
        }
      } else {
        for (int col = OUTPUT_MEM; col < KECCAK_SPONGE_WIDTH; col++) put(col, 0u);
      }
      if (round == 23 && !final) {     // the call's next block follows and is absorbed into this block's permuted state
        if (blk + 1 >= n_blocks) {
          fail(5);
        } else {
          const uint32_t* nx = e + KECCAK_SPONGE_BLOCK_WORDS;
          bool ok = nx[E_BLOCK_INDEX] == index + 1 && nx[E_INPUT_LEN] == len && nx[E_SHARD] == e[E_SHARD] && nx[E_CLK] == e[E_CLK];
#pragma unroll
          for (int j = 0; j < 50; j++) {
            const uint64_t lane = j / 2 == 0 ? out00 : app[j / 2];
            ok = ok && (nx[E_XORED + j] ^ (j < 36 ? nx[E_READS + 5 * j] : 0u)) == (uint32_t)(lane >> (32 * (j & 1)));
          }
          if (!ok) fail(5);
        }
      }
    }
  }
  if (count) {
    __syncthreads();
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x)
      if (hkeys[i] != HASH_EMPTY) atomicAdd(counts + hkeys[i], hvals[i]);
  }
}

// ---- SHA-256 precompiles (syscall/precompiles/sha256/): ShaExtend — 48 rows per call, row j holds w[16 + j] = w[j] + s0(w[j + 1]) +
// w[j + 9] + s1(w[j + 14]) with every rotate / shift / xor / add spelled out in byte columns (extend/columns.rs:17-73, trace.rs:103-150) —
// and ShaCompress — 80 rows per call: eight that read the state, the 64 rounds, eight that add the result to the words read and write
// them back (compress/columns.rs:17-108, trace.rs:127-302). One thread per row builds the row as canonical integers in registers /
// scratch and stores it column by column; a ShaCompress thread first re-runs the rounds before its own (at most 63, a few dozen integer
// operations each). The operation gadgets (operations/fixed_rotate_right.rs, fixed_shift_right.rs, xor.rs, and.rs, not.rs, add.rs,
// add4.rs, add5.rs) are the ShaOps methods: each fills its columns and records the byte lookups its populate() records.
constexpr int SHA_EXTEND_WIDTH = 176, SHA_EXTEND_EVENT_WORDS = 1251, SHA_COMPRESS_WIDTH = 262, SHA_COMPRESS_EVENT_WORDS = 412;
// the order-16 subgroup g^k (g = two_adic_generator(4)) and the inverses of g^k - g and g^k - 1 (0 where that is 0): ShaExtendCols::populate_flags
__constant__ uint32_t d_sha_cycle16[16] = {1u, 148625052u, 1748172362u, 665723362u, 2113994754u, 982097957u, 391001680u, 668978722u, 2130706432u,
                                           1982081381u, 382534071u, 1464983071u, 16711679u, 1148608476u, 1739704753u, 1461727711u};
__constant__ uint32_t d_sha_inv_start[16] = {1228590512u, 0u, 1571094643u, 501619392u, 497113253u, 1627680u, 962112900u, 1167342754u, 1236834581u,
                                             334489361u, 1562850574u, 1632342401u, 1837572255u, 667351042u, 171865469u, 167359330u};
__constant__ uint32_t d_sha_inv_end[16] = {0u, 902115921u, 1069475251u, 1703008016u, 8355839u, 1728187304u, 1052763572u, 893871851u, 1065353216u,
                                           1236834581u, 1077942860u, 402519128u, 2122350593u, 427698416u, 1061231181u, 1228590511u};
__constant__ uint32_t d_sha_k[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
struct ShaOps {
  const LookupSink& sink;
  bool count;
  __device__ __forceinline__ void range(uint32_t v) const {
    if (count) { lookup(sink, B_U8RANGE, v, v >> 8); lookup(sink, B_U8RANGE, v >> 16, v >> 24); }
  }
  // value(4), shift(4), carry(4): bytes moved by rotation / 8, then shifted by rotation % 8 with the bits that fall out carried down
  __device__ __forceinline__ uint32_t shift_or_rotate(uint32_t* r, uint32_t x, int rotation, bool rotate) const {
    const int nbytes = rotation / 8, nbits = rotation % 8;
    uint32_t first_shift = 0, last_carry = 0;
    for (int i = 3; i >= 0; i--) {
      const uint32_t in = rotate ? (x >> (8 * ((i + nbytes) % 4))) & 0xff : (i + nbytes < 4 ? (x >> (8 * (i + nbytes))) & 0xff : 0u);
      const uint32_t shift = in >> nbits, carry = in & ((1u << nbits) - 1);
      if (count) lookup(sink, B_SHRCARRY, in, nbits);
      r[4 + i] = shift; r[8 + i] = carry;
      if (i == 3) first_shift = shift; else r[i] = shift + (last_carry << (8 - nbits));
      last_carry = carry;
    }
    r[3] = rotate ? first_shift + (last_carry << (8 - nbits)) : first_shift;
    return rotate ? (x >> rotation) | (x << (32 - rotation)) : x >> rotation;
  }
  __device__ __forceinline__ uint32_t bitwise(uint32_t* r, uint32_t op, uint32_t x, uint32_t y) const {
    const uint32_t out = op == B_XOR ? x ^ y : x & y;
    word(r, out);
    if (count)
      for (int i = 0; i < 4; i++) lookup(sink, op, x >> (8 * i), y >> (8 * i));
    return out;
  }
  __device__ __forceinline__ uint32_t not_(uint32_t* r, uint32_t x) const { word(r, ~x); range(x); return ~x; }
  // value(4), is_carry_0..n-1 (4 each, one-hot per byte), carry(4)
  __device__ __forceinline__ uint32_t add_many(uint32_t* r, const uint32_t* v, int n) const {
    uint32_t sum = 0;
    for (int k = 0; k < n; k++) sum += v[k];
    word(r, sum);
    uint32_t carry = 0;
    for (int i = 0; i < 4; i++) {
      uint32_t res = carry;
      for (int k = 0; k < n; k++) res += (v[k] >> (8 * i)) & 0xff;
      carry = res >> 8;
      for (int c = 0; c < n; c++) r[4 + 4 * c + i] = fbool(carry == (uint32_t)c);
      r[4 + 4 * n + i] = carry;
    }
    for (int k = 0; k < n; k++) range(v[k]);
    range(sum);
    return sum;
  }
  __device__ __forceinline__ uint32_t add(uint32_t* r, uint32_t a, uint32_t b) const {     // value(4), carry(3)
    word(r, a + b);
    uint32_t carry = 0;
    for (int i = 0; i < 3; i++) {
      carry = fbool((((a >> (8 * i)) & 0xff) + ((b >> (8 * i)) & 0xff) + carry) > 255);
      r[4 + i] = carry;
    }
    range(a); range(b); range(a + b);
    return a + b;
  }
  __device__ __forceinline__ void read(uint32_t* r, const uint32_t* rec) const {            // MemoryReadCols from a five-word record
    memory_access_cols(rec[0], rec[1], rec[2], rec[3], rec[4], r);
    if (count) access_lookups(r, sink);
  }
  __device__ __forceinline__ void write(uint32_t* r, const uint32_t* rec) const {           // MemoryWriteCols from a six-word record
    memory_write_cols(rec, r);
    if (count) { lookup(sink, B_U16RANGE, r[11] >> 8, r[11]); lookup(sink, B_U8RANGE, 0, r[12]); }
  }
};

__global__ __launch_bounds__(THREADS) void sha_extend_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                                           uint32_t* counts, int* __restrict__ bad) {
  enum { SHARD = 0, CLK = 1, W_PTR = 2, I = 3, CYCLE_16 = 4, CYCLE_16_START = 5, CYCLE_16_END = 7, CYCLE_48 = 9, CYCLE_48_START = 12, CYCLE_48_END = 13,
         W_I_MINUS_15 = 14, RR_7 = 23, RR_18 = 35, RS_3 = 47, S0_INTERMEDIATE = 59, S0 = 63, W_I_MINUS_2 = 67, RR_17 = 76, RR_19 = 88, RS_10 = 100,
         S1_INTERMEDIATE = 112, S1 = 116, W_I_MINUS_16 = 120, W_I_MINUS_7 = 129, S2 = 138, W_I = 162, IS_REAL = 175 };
  enum { E_READS_15 = 3, E_READS_2 = 3 + 240, E_READS_16 = 3 + 480, E_READS_7 = 3 + 720, E_WRITES = 3 + 960 };
  extern __shared__ uint32_t hash_lds[];
  uint32_t* hkeys = hash_lds;
  uint32_t* hvals = hash_lds + HASH_SLOTS;
  const bool count = counts != nullptr;
  if (count) {
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
    __syncthreads();
  }
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    uint32_t r[SHA_EXTEND_WIDTH];
    for (int c = 0; c < SHA_EXTEND_WIDTH; c++) r[c] = 0;
    const bool real = row < 48 * n_events;
    const size_t j = row % 48;               // a call's row; padding rows count on from the real ones (a multiple of 48)
    const int k16 = (int)((j + 1) % 16);
    r[CYCLE_16] = d_sha_cycle16[k16];
    r[CYCLE_16_START] = d_sha_inv_start[k16]; r[CYCLE_16_START + 1] = fbool(k16 == 1);
    r[CYCLE_16_END] = d_sha_inv_end[k16]; r[CYCLE_16_END + 1] = fbool(k16 == 0);
    r[I] = 16 + (uint32_t)j;
    r[CYCLE_48] = fbool(j < 16); r[CYCLE_48 + 1] = fbool(j >= 16 && j < 32); r[CYCLE_48 + 2] = fbool(j >= 32);
    if (real) {
      const uint32_t* e = events + (row / 48) * SHA_EXTEND_EVENT_WORDS;
      const LookupSink sink{hkeys, hvals, HASH_SLOTS - 1, counts};
      const ShaOps ops{sink, count};
      r[IS_REAL] = 1;
      r[CYCLE_48_START] = fbool(j == 0); r[CYCLE_48_END] = fbool(j == 47);
      r[SHARD] = e[0]; r[CLK] = e[1]; r[W_PTR] = e[2];
      const uint32_t *m15 = e + E_READS_15 + 5 * j, *m2 = e + E_READS_2 + 5 * j, *m16 = e + E_READS_16 + 5 * j, *m7 = e + E_READS_7 + 5 * j;
      ops.read(r + W_I_MINUS_15, m15); ops.read(r + W_I_MINUS_2, m2); ops.read(r + W_I_MINUS_16, m16); ops.read(r + W_I_MINUS_7, m7);
      const uint32_t w15 = m15[0], w2 = m2[0];
      const uint32_t s0 = ops.bitwise(r + S0, B_XOR, ops.bitwise(r + S0_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + RR_7, w15, 7, true),
                                                                 ops.shift_or_rotate(r + RR_18, w15, 18, true)),
                                      ops.shift_or_rotate(r + RS_3, w15, 3, false));
      const uint32_t s1 = ops.bitwise(r + S1, B_XOR, ops.bitwise(r + S1_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + RR_17, w2, 17, true),
                                                                 ops.shift_or_rotate(r + RR_19, w2, 19, true)),
                                      ops.shift_or_rotate(r + RS_10, w2, 10, false));
      const uint32_t four[4] = {m16[0], s0, m7[0], s1};
      const uint32_t w_i = ops.add_many(r + S2, four, 4);
      const uint32_t* wr = e + E_WRITES + 6 * j;
      if (wr[0] != w_i) *bad = 1;
      ops.write(r + W_I, wr);
    }
    for (int c = 0; c < SHA_EXTEND_WIDTH; c++) out[(size_t)c * height + row] = kb::to_monty(r[c]);
  }
  if (count) {
    __syncthreads();
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x)
      if (hkeys[i] != HASH_EMPTY) atomicAdd(counts + hkeys[i], hvals[i]);
  }
}

__global__ __launch_bounds__(THREADS) void sha_compress_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                                             uint32_t* counts, int* __restrict__ bad) {
  enum { SHARD = 0, CLK = 1, W_PTR = 2, H_PTR = 3, START = 4, OCTET = 5, OCTET_NUM = 13, MEM = 23, MEM_ADDR = 36, A = 37, K = 69, E_RR_6 = 73, E_RR_11 = 85,
         E_RR_25 = 97, S1_INTERMEDIATE = 109, S1 = 113, E_AND_F = 117, E_NOT = 121, E_NOT_AND_G = 125, CH = 129, TEMP1 = 133, A_RR_2 = 161, A_RR_13 = 173,
         A_RR_22 = 185, S0_INTERMEDIATE = 197, S0 = 201, A_AND_B = 205, A_AND_C = 209, B_AND_C = 213, MAJ_INTERMEDIATE = 217, MAJ = 221, TEMP2 = 225,
         D_ADD_TEMP1 = 232, TEMP1_ADD_TEMP2 = 239, FINALIZED_OPERAND = 246, FINALIZE_ADD = 250, IS_INITIALIZE = 257, IS_COMPRESSION = 258,
         IS_FINALIZE = 259, IS_LAST_ROW = 260, IS_REAL = 261 };
  enum { E_H_READS = 4, E_W_READS = 44, E_H_WRITES = 364 };
  extern __shared__ uint32_t hash_lds[];
  uint32_t* hkeys = hash_lds;
  uint32_t* hvals = hash_lds + HASH_SLOTS;
  const bool count = counts != nullptr;
  if (count) {
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
    __syncthreads();
  }
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    uint32_t r[SHA_COMPRESS_WIDTH];
    for (int c = 0; c < SHA_COMPRESS_WIDTH; c++) r[c] = 0;
    const bool real = row < 80 * n_events;
    const int step = (int)(row % 80), octet = step % 8, octet_num = step / 8;
    r[OCTET + octet] = 1; r[OCTET_NUM + octet_num] = 1;
    r[IS_LAST_ROW] = fbool(step == 79);
    if (!real) {
      if (octet_num != 0 && octet_num != 9) word(r + K, d_sha_k[step - 8]);
    } else {
      const uint32_t* e = events + (row / 80) * SHA_COMPRESS_EVENT_WORDS;
      const LookupSink sink{hkeys, hvals, HASH_SLOTS - 1, counts};
      const ShaOps ops{sink, count};
      r[IS_REAL] = 1;
      r[SHARD] = e[0]; r[CLK] = e[1]; r[W_PTR] = e[2]; r[H_PTR] = e[3];
      r[START] = fbool(step == 0);
      uint32_t v[8], og[8];
      for (int i = 0; i < 8; i++) v[i] = og[i] = e[E_H_READS + 5 * i];
      const int rounds_before = octet_num == 0 ? 0 : octet_num == 9 ? 64 : step - 8;
      for (int j = 0; j < rounds_before; j++) {     // Sha256CompressSyscall::execute's round (syscalls/precompiles/sha256/compress.rs:63-85)
        const uint32_t a = v[0], b = v[1], c = v[2], d = v[3], ee = v[4], f = v[5], g = v[6], hh = v[7];
        const uint32_t s1 = ((ee >> 6) | (ee << 26)) ^ ((ee >> 11) | (ee << 21)) ^ ((ee >> 25) | (ee << 7));
        const uint32_t temp1 = hh + s1 + ((ee & f) ^ (~ee & g)) + d_sha_k[j] + e[E_W_READS + 5 * j];
        const uint32_t s0 = ((a >> 2) | (a << 30)) ^ ((a >> 13) | (a << 19)) ^ ((a >> 22) | (a << 10));
        const uint32_t temp2 = s0 + ((a & b) ^ (a & c) ^ (b & c));
        v[7] = g; v[6] = f; v[5] = ee; v[4] = d + temp1; v[3] = c; v[2] = b; v[1] = a; v[0] = temp1 + temp2;
      }
      for (int i = 0; i < 8; i++) word(r + A + 4 * i, v[i]);
      if (octet_num == 0) {
        r[IS_INITIALIZE] = 1;
        const uint32_t* m = e + E_H_READS + 5 * octet;
        word(r + MEM, m[0]);
        ops.read(r + MEM + 4, m);
        r[MEM_ADDR] = e[3] + 4u * octet;
      } else if (octet_num < 9) {
        const int j = step - 8;
        r[IS_COMPRESSION] = 1;
        word(r + K, d_sha_k[j]);
        const uint32_t* m = e + E_W_READS + 5 * j;
        word(r + MEM, m[0]);
        ops.read(r + MEM + 4, m);
        r[MEM_ADDR] = e[2] + 4u * j;
        const uint32_t a = v[0], b = v[1], c = v[2], d = v[3], ee = v[4], f = v[5], g = v[6], hh = v[7];
        const uint32_t s1 = ops.bitwise(r + S1, B_XOR, ops.bitwise(r + S1_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + E_RR_6, ee, 6, true),
                                                                   ops.shift_or_rotate(r + E_RR_11, ee, 11, true)),
                                        ops.shift_or_rotate(r + E_RR_25, ee, 25, true));
        const uint32_t e_and_f = ops.bitwise(r + E_AND_F, B_AND, ee, f);
        const uint32_t e_not = ops.not_(r + E_NOT, ee);
        const uint32_t ch = ops.bitwise(r + CH, B_XOR, e_and_f, ops.bitwise(r + E_NOT_AND_G, B_AND, e_not, g));
        const uint32_t five[5] = {hh, s1, ch, m[0], d_sha_k[j]};
        const uint32_t temp1 = ops.add_many(r + TEMP1, five, 5);
        const uint32_t s0 = ops.bitwise(r + S0, B_XOR, ops.bitwise(r + S0_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + A_RR_2, a, 2, true),
                                                                   ops.shift_or_rotate(r + A_RR_13, a, 13, true)),
                                        ops.shift_or_rotate(r + A_RR_22, a, 22, true));
        const uint32_t a_and_b = ops.bitwise(r + A_AND_B, B_AND, a, b), a_and_c = ops.bitwise(r + A_AND_C, B_AND, a, c);
        const uint32_t b_and_c = ops.bitwise(r + B_AND_C, B_AND, b, c);
        const uint32_t maj = ops.bitwise(r + MAJ, B_XOR, ops.bitwise(r + MAJ_INTERMEDIATE, B_XOR, a_and_b, a_and_c), b_and_c);
        const uint32_t temp2 = ops.add(r + TEMP2, s0, maj);
        ops.add(r + D_ADD_TEMP1, d, temp1);
        ops.add(r + TEMP1_ADD_TEMP2, temp1, temp2);
      } else {
        r[IS_FINALIZE] = 1;
        const uint32_t sum = ops.add(r + FINALIZE_ADD, og[octet], v[octet]);
        const uint32_t* m = e + E_H_WRITES + 6 * octet;
        if (m[0] != sum || m[3] != og[octet]) *bad = 1;
        ops.write(r + MEM, m);
        r[MEM_ADDR] = e[3] + 4u * octet;
        word(r + FINALIZED_OPERAND, v[octet]);
      }
    }
    for (int c = 0; c < SHA_COMPRESS_WIDTH; c++) out[(size_t)c * height + row] = kb::to_monty(r[c]);
  }
  if (count) {
    __syncthreads();
    for (int i = threadIdx.x; i < HASH_SLOTS; i += blockDim.x)
      if (hkeys[i] != HASH_EMPTY) atomicAdd(counts + hkeys[i], hvals[i]);
  }
}

// ==== Big-field precompiles: the Ed25519 chips, the short-Weierstrass chips, the field-tower chips ====================================================
// (syscall/precompiles/edwards/, weierstrass/, fptower/ over the gadgets of operations/field/). A row holds a handful of *field gadgets*: result,
// carry, witness_low, witness_high of an identity op = result + carry p over the integers, checked as a polynomial identity in x = 2^8 on byte
// limbs: op(x) - result(x) - carry(x) p(x) vanishes at 256, its quotient by (x - 256), shifted by the field's witness offset and split into two
// bytes, is the witness (operations/field/util.rs:21-66). One thread per row. The integers are bigfield.cuh's (32-bit limbs in registers,
// Barrett, Fermat inverse); the byte-limb polynomials live in LDS — one accumulator of 2 N - 1 words and two staged operands per thread, laid
// out [element][thread] so that a wavefront's accesses never conflict. (In scratch memory, where the compiler puts a dynamically indexed local
// array, the same loops made EdAddAssign twenty milliseconds per 2^16 rows.)
// Byte lookups. A row range-checks ≈ 1500 limb bytes in pairs, and witness bytes are uniformly random: 2^16 rows make 5 * 10^7 increments
// spread evenly over the 65536 U8Range counters. No per-block table combines those (a block of 256 rows meets every key about three times),
// and as device-scope atomics they cost 20 ms per 2^16 rows — ten times the rows themselves. So the row kernels count only their skewed
// lookups (memory timestamps, comparisons) in the block's LDS table, and u8_pair_histogram below counts the range checks from the columns.
constexpr int BF_HASH_SLOTS = 2048;
__host__ __device__ constexpr int bf_threads(int nl) { return nl == 8 ? 256 : 128; }                    // rows per block
__host__ __device__ constexpr int bf_words_per_thread(int nl) { return (8 * nl - 1) + 2 * nl; }         // accumulator + two operands
__host__ __device__ constexpr size_t bf_lds_bytes(int nl, bool count) {
  return (count ? 2 * (size_t)BF_HASH_SLOTS * 4 : 0) + (size_t)bf_threads(nl) * bf_words_per_thread(nl) * 4;
}
enum { FOP_ADD = 0, FOP_SUB = 1, FOP_MUL = 2, FOP_DIV = 3 };
template <int NL> struct CurveField { bigfield::Modulus<NL> m; uint32_t a[NL]; int32_t witness_offset; };
// a row's column writers: canonical value -> Montgomery word of the column-major matrix; the memory columns of a precompile's records
struct RowCols {
  uint32_t* out; size_t height, row; const LookupSink& sink; bool count;
  __device__ __forceinline__ void put(int col, uint32_t canonical) const { out[(size_t)col * height + row] = kb::to_monty(canonical); }
  __device__ void zeros(int base, int n) const { for (int i = 0; i < n; i++) put(base + i, 0u); }
  // the columns of a 6-word write record (MemoryWriteCols, 13) and of a 5-word read record (MemoryReadCols, 9); zero for a padding row
  __device__ void write_cols(int base, const uint32_t* rec) const {
    uint32_t mw[13];
    for (int c = 0; c < 13; c++) mw[c] = 0;
    if (rec) {
      memory_write_cols(rec, mw);
      if (count) { lookup(sink, B_U16RANGE, mw[11] >> 8, mw[11]); lookup(sink, B_U8RANGE, 0, mw[12]); }
    }
    for (int c = 0; c < 13; c++) put(base + c, mw[c]);
  }
  __device__ void read_cols(int base, const uint32_t* rec) const {
    uint32_t mr[9];
    for (int c = 0; c < 9; c++) mr[c] = 0;
    if (rec) {
      memory_access_cols(rec[0], rec[1], rec[2], rec[3], rec[4], mr);
      if (count) access_lookups(mr, sink);
    }
    for (int c = 0; c < 9; c++) put(base + c, mr[c]);
  }
};
template <int NL> struct FieldRow : RowCols {
  static constexpr int N = 4 * NL, NW = 2 * N - 2, G = 2 * N + 2 * NW, T = bf_threads(NL);
  const CurveField<NL>& f;
  int32_t* van;          // LDS: 2 N - 1 coefficients, stride T
  uint32_t *opa, *opb;   // LDS: NL limbs each, stride T
  // the block's LDS after the lookup table (if any): accumulators, then the operands
  __device__ static FieldRow make(uint32_t* out, size_t height, size_t row, const LookupSink& sink, bool count, const CurveField<NL>& f, uint32_t* lds) {
    int32_t* van = (int32_t*)lds + threadIdx.x;
    uint32_t* opa = lds + (size_t)T * (2 * N - 1) + threadIdx.x;
    return FieldRow{{out, height, row, sink, count}, f, van, opa, opa + (size_t)T * NL};
  }
  __device__ __forceinline__ int32_t& V(int k) const { return van[(size_t)k * T]; }
  __device__ __forceinline__ int32_t A(int i) const { return (int32_t)((opa[(size_t)(i >> 2) * T] >> (8 * (i & 3))) & 0xff); }
  __device__ __forceinline__ int32_t B(int i) const { return (int32_t)((opb[(size_t)(i >> 2) * T] >> (8 * (i & 3))) & 0xff); }
  __device__ __forceinline__ void stage_a(const uint32_t* x) const {
#pragma unroll
    for (int l = 0; l < NL; l++) opa[(size_t)l * T] = x[l];
  }
  __device__ __forceinline__ void stage_b(const uint32_t* x) const {
#pragma unroll
    for (int l = 0; l < NL; l++) opb[(size_t)l * T] = x[l];
  }
  __device__ void clear() const { for (int k = 0; k < 2 * N - 1; k++) V(k) = 0; }
  // accumulator += sign * (staged a)(x) * (staged b)(x): one coefficient at a time, summed in a register
  __device__ void mac_staged(int sign) const {
    for (int k = 0; k < 2 * N - 1; k++) {
      const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
      int32_t acc = 0;
      for (int i = lo; i <= hi; i++) acc += A(i) * B(k - i);
      V(k) += sign * acc;
    }
  }
  __device__ void mac(const uint32_t* x, const uint32_t* y) const { stage_a(x); stage_b(y); mac_staged(1); }
  __device__ void add(const uint32_t* x, int sign) const {
    stage_a(x);
    for (int i = 0; i < N; i++) V(i) += sign * A(i);
  }
  // the gadget's columns at `base` from the accumulator (the identity's polynomial before the carry term) and the two integers. Its byte
  // lookups — the range checks of all four limb vectors in pairs (i, i + 1), even i — are not counted here: u8_pair_histogram reads them
  // off the finished columns
  __device__ void gadget(int base, const uint32_t* result, const uint32_t* carry) const {
    stage_a(carry); stage_b(f.m.p); mac_staged(-1);
    int32_t above = 0;
    for (int k = NW; k >= 1; k--) {          // w[k - 1] = van[k] + 256 w[k]
      above = V(k) + 256 * above;
      const uint32_t shifted = (uint32_t)(above + f.witness_offset);
      put(base + 2 * N + k - 1, shifted & 0xff);
      put(base + 2 * N + NW + k - 1, shifted >> 8);
    }
#pragma unroll
    for (int l = 0; l < NL; l++)
#pragma unroll
      for (int j = 0; j < 4; j++) { put(base + 4 * l + j, (result[l] >> (8 * j)) & 0xff); put(base + N + 4 * l + j, (carry[l] >> (8 * j)) & 0xff); }
  }
  // FieldOpCols::populate_with_modulus (operations/field/field_op.rs:154-224) for a, b below p; Sub and Div through the reversed identities
  __device__ void op(int base, const uint32_t* a, const uint32_t* b, int kind, uint32_t* res) const {
    const bigfield::Modulus<NL>& m = f.m;
    uint32_t t[2 * NL], t2[2 * NL], q[NL + 1];
    for (int i = 0; i <= NL; i++) q[i] = 0;
    clear();
    if (kind == FOP_ADD) {
      for (int i = 0; i < NL; i++) res[i] = a[i];
      const uint32_t carry_out = bigfield::add<NL>(res, b);
      const uint32_t wraps = carry_out | (bigfield::cmp<NL>(res, m.p) >= 0 ? 1u : 0u);
      if (wraps) bigfield::sub<NL>(res, m.p);
      q[0] = wraps;
      add(a, 1); add(b, 1); add(res, -1);
    } else if (kind == FOP_SUB) {                 // result + b = a + carry p
      for (int i = 0; i < NL; i++) res[i] = a[i];
      const uint32_t borrows = bigfield::sub<NL>(res, b);
      if (borrows) bigfield::add<NL>(res, m.p);
      q[0] = borrows;
      add(res, 1); add(b, 1); add(a, -1);
    } else if (kind == FOP_MUL) {
      bigfield::mul<NL, NL>(a, b, t);
      bigfield::divmod<NL>(t, m, q, res);
      mac(a, b); add(res, -1);
    } else {                                      // result * b = a + carry p
      bool zero = true;
      for (int i = 0; i < NL; i++) zero = zero && a[i] == 0;
      if (zero) { for (int i = 0; i < NL; i++) res[i] = 0; } else {
        uint32_t inv[NL];
        if constexpr (NL > 8) bigfield::inverse_call<NL>(b, &m, inv); else bigfield::inverse<NL>(b, m, inv);
        bigfield::mulmod<NL>(a, inv, m, res);
      }
      bigfield::mul<NL, NL>(res, b, t);
      for (int i = 0; i < 2 * NL; i++) t2[i] = i < NL ? a[i] : 0u;
      bigfield::sub<2 * NL>(t, t2);
      uint32_t rem[NL];
      bigfield::divmod<NL>(t, m, q, rem);
      mac(res, b); add(a, -1);
    }
    gadget(base, res, q);
  }
  // FieldLtCols::populate (operations/field/range.rs:27-60): the flag of the most significant byte where lhs < rhs, the two compared bytes
  __device__ void lt(int base, const uint32_t* lhs, const uint32_t* rhs) const {
    stage_a(lhs); stage_b(rhs);
    int at = -1;
    for (int i = N - 1; i >= 0 && at < 0; i--) {
      const int32_t x = A(i), y = B(i);
      if (x < y) at = i;
      else if (x > y) break;        // not below: no flag (callers report the error)
    }
    for (int i = 0; i < N; i++) put(base + i, i == at ? 1u : 0u);
    const uint32_t a = at >= 0 ? (uint32_t)A(at) : 0u, b = at >= 0 ? (uint32_t)B(at) : 0u;
    put(base + N, a); put(base + N + 1, b);
    if (count && at >= 0) lookup(sink, B_LTU, a, b);
  }
  __device__ void lt(int base, const uint32_t* lhs) const { lt(base, lhs, f.m.p); }      // against the modulus
  // FieldInnerProductCols::populate (operations/field/field_inner_product.rs:27-79) of (a0, a1) . (b0, 1): a0 b0 + a1 = result + carry p
  __device__ void inner_with_one(int base, const uint32_t* a0, const uint32_t* b0, const uint32_t* a1, uint32_t* res) const {
    uint32_t t[2 * NL], t2[2 * NL], q[NL + 1];
    bigfield::mul<NL, NL>(a0, b0, t);
    for (int i = 0; i < 2 * NL; i++) t2[i] = i < NL ? a1[i] : 0u;
    bigfield::add<2 * NL>(t, t2);
    bigfield::divmod<NL>(t, f.m, q, res);
    clear(); mac(a0, b0); add(a1, 1); add(res, -1);
    gadget(base, res, q);
  }
};
// every big-field kernel starts and ends the same way: the block's lookup table in LDS, flushed into the global counters at the end
struct BfBlock {
  uint32_t *hkeys, *hvals, *scratch;
  bool count;
  __device__ BfBlock(uint32_t* lds, uint32_t* counts) : hkeys(lds), hvals(lds + BF_HASH_SLOTS), scratch(counts ? lds + 2 * BF_HASH_SLOTS : lds), count(counts != nullptr) {
    if (count) {
      for (int i = threadIdx.x; i < BF_HASH_SLOTS; i += blockDim.x) { hkeys[i] = HASH_EMPTY; hvals[i] = 0; }
      __syncthreads();
    }
  }
  __device__ void flush(uint32_t* counts) const {
    if (count) {
      __syncthreads();
      for (int i = threadIdx.x; i < BF_HASH_SLOTS; i += blockDim.x)
        if (hkeys[i] != HASH_EMPTY) atomicAdd(counts + hkeys[i], hvals[i]);
    }
  }
};

// The U8Range lookups of adjacent byte columns, counted from the trace: for every real row and every pair of columns (start + 2 j,
// start + 2 j + 1) of the given segments (a segment's odd last column pairs with zero), one lookup (U8Range, b, c). A block takes a slab of rows and half of the key space (the
// top bit of b), reads the slab's byte columns (coalesced along the rows; the other half's block reads them again, out of
// the memory-side cache) and counts in a dense LDS histogram of 32768 counters — no probing, no overflow. It leaves its counters in its
// own part of `partial` [slab][key]; u8_pair_reduce adds the slabs up into the lookup counters. Rows per slab: a power of two >= 256.
struct U8Segments { int n; int start[4]; int cols[4]; int reps[4]; int stride[4]; };      // reps = 0 or 1: once; else `reps` copies `stride` columns apart
constexpr int U8H_RANGES = 2, U8H_KEYS = 65536 / U8H_RANGES, U8H_THREADS = 1024, U8H_MAX_SLABS = 256;      // 128 KiB of counters: one block of sixteen waves per CU
__global__ __launch_bounds__(U8H_THREADS) void u8_pair_histogram(const uint32_t* __restrict__ trace, size_t height, size_t n_real, const U8Segments seg,
                                                                 int log_slab_rows, uint32_t* __restrict__ partial) {
  extern __shared__ uint32_t hist[];      // U8H_KEYS counters
  const uint32_t slab = blockIdx.x / U8H_RANGES, range = blockIdx.x % U8H_RANGES;
  for (int i = threadIdx.x; i < U8H_KEYS; i += U8H_THREADS) hist[i] = 0;
  __syncthreads();
  const size_t row0 = (size_t)slab << log_slab_rows;
  const uint32_t slab_rows = 1u << log_slab_rows;
  for (int sg = 0; sg < seg.n; sg++) {
    const uint32_t* base = trace + (size_t)seg.start[sg] * height + row0;
    const uint32_t pairs = (uint32_t)(seg.cols[sg] + 1) / 2, reps = seg.reps[sg] > 1 ? (uint32_t)seg.reps[sg] : 1u;
    const uint32_t items = (pairs * reps) << log_slab_rows;      // (pair, row) with the row fastest: a wavefront shares the pair
    // four (pair, row) items per thread and step, their eight loads issued before any is used: an out-of-range item reads the slab's first
    // word instead (always there) and is dropped afterwards, so no load sits behind a branch
    for (uint32_t first = threadIdx.x; first < items; first += 4 * U8H_THREADS) {
      uint32_t bw[4], cw[4];
      bool keep[4], alone[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t idx = first + u * U8H_THREADS;
        const uint32_t p = idx >> log_slab_rows, r = idx & (slab_rows - 1);
        const uint32_t rep = p / pairs, pair = p - rep * pairs;      // the same for a whole wavefront
        keep[u] = idx < items && row0 + r < n_real;
        alone[u] = 2 * pair + 1 >= (uint32_t)seg.cols[sg];      // the last column of an odd segment is checked with a zero (ByteRecord::add_u8_range_checks)
        const uint32_t* at = keep[u] ? base + ((size_t)rep * seg.stride[sg] + 2 * pair) * height + r : base;
        bw[u] = at[0];
        cw[u] = at[keep[u] && !alone[u] ? height : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t b = kb::from_monty(bw[u]), c = alone[u] ? 0u : kb::from_monty(cw[u]);
        if (keep[u] && (b >> 7) == range) atomicAdd(&hist[(b & 0x7f) << 8 | (c & 0xff)], 1u);
      }
    }
  }
  __syncthreads();
  uint32_t* mine = partial + (size_t)slab * 65536 + (size_t)range * U8H_KEYS;
  for (int i = threadIdx.x; i < U8H_KEYS; i += U8H_THREADS) mine[i] = hist[i];
}
__global__ __launch_bounds__(256) void u8_pair_reduce(const uint32_t* __restrict__ partial, int slabs, uint32_t* __restrict__ u8range_counts) {
  const uint32_t key = blockIdx.x * 256 + threadIdx.x;
  uint32_t sum = 0;
  for (int s = 0; s < slabs; s++) sum += partial[(size_t)s * 65536 + key];
  if (sum) atomicAdd(u8range_counts + key, sum);
}

// ---- EdAddAssign (syscall/precompiles/edwards/ed_add.rs:41-57, :68-95, :210-243): one Ed25519 point addition per row, 1861 columns: the memory
// columns of p (written at clk + 1) and q, and eight gadgets: x3_numerator, y3_numerator (FieldInnerProductCols), x1_mul_y1, x2_mul_y2, f, d_mul_f
// (FieldOpCols), x3_ins, y3_ins (FieldDenCols). Padding rows hold the gadgets of the zero inputs: zero except witness_high = 2^14 >> 8.
constexpr int ED_ADD_WIDTH = 1861, ED_ADD_EVENT_WORDS = 180, ED_LIMBS = 32, ED_GADGET = 188;
__constant__ CurveField<8> d_ed25519 = {
    {{0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu},
     {76u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 2u}},      // floor(2^512 / (2^255 - 19)) = 2^257 + 76 (the next term, 19^2 * 4 / 2^255, is below one)
    {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u},              // not a Weierstrass curve: `a` is unused
    1 << 14};
__constant__ uint32_t d_ed25519_d[8] = {0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu};
__global__ __launch_bounds__(bf_threads(8)) void ed_add_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                                             uint32_t* counts, int* __restrict__ bad) {
  enum { IS_REAL = 0, SHARD = 1, CLK = 2, P_PTR = 3, Q_PTR = 4, P_ACCESS = 5, Q_ACCESS = 5 + 16 * 13, GADGETS = 5 + 16 * 13 + 16 * 9, G = ED_GADGET };
  enum { E_P_RECORDS = 4, E_Q_RECORDS = 4 + 96 };
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * ED_ADD_EVENT_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const FieldRow<8> R = FieldRow<8>::make(out, height, row, sink, blk.count && real, d_ed25519, blk.scratch);
    const bigfield::Modulus<8>& m = d_ed25519.m;
    uint32_t x1[8], y1[8], x2[8], y2[8];
    for (int k = 0; k < 8; k++) {
      x1[k] = real ? e[E_P_RECORDS + 6 * k + 3] : 0u;            // p = the previous values of the p write records
      y1[k] = real ? e[E_P_RECORDS + 6 * (8 + k) + 3] : 0u;
      x2[k] = real ? e[E_Q_RECORDS + 5 * k] : 0u;
      y2[k] = real ? e[E_Q_RECORDS + 5 * (8 + k)] : 0u;
    }
    uint32_t t[16], t2[16], q[9], x3n[8], y3n[8], a[8], b[8], f[8], df[8], res[8];
    // x3_numerator = x1 y2 + x2 y1, y3_numerator = y1 y2 + x1 x2 (FieldInnerProductCols::populate, field_inner_product.rs:27-79)
    auto inner = [&](int base, const uint32_t* a0, const uint32_t* b0, const uint32_t* a1, const uint32_t* b1, uint32_t* result) {
      bigfield::mul<8, 8>(a0, b0, t); bigfield::mul<8, 8>(a1, b1, t2);
      bigfield::add<16>(t, t2);                        // below 2 p^2 < 2^511
      bigfield::divmod<8>(t, m, q, result);
      R.clear(); R.mac(a0, b0); R.mac(a1, b1); R.add(result, -1);
      R.gadget(base, result, q);
    };
    // FieldDenCols::populate (field_den.rs:27-81): result (1 +- b) = a. The two denominators 1 + d f and 1 - d f are inverted together: one
    // Fermat inverse of their product (Montgomery's trick), then one product each
    uint32_t inv_plus[8], inv_minus[8];
    auto denominators = [&](const uint32_t* bb) {
      uint32_t dp[8], dm[8], prod[8], inv[8];
      const uint32_t one[8] = {1};
      for (int i = 0; i < 8; i++) { dp[i] = bb[i]; dm[i] = m.p[i]; }
      bigfield::sub<8>(dm, bb);
      bigfield::add<8>(dp, one); bigfield::add<8>(dm, one);
      if (bigfield::cmp<8>(dp, m.p) >= 0) bigfield::sub<8>(dp, m.p);
      if (bigfield::cmp<8>(dm, m.p) >= 0) bigfield::sub<8>(dm, m.p);
      bigfield::mulmod<8>(dp, dm, m, prod);
      bigfield::inverse<8>(prod, m, inv);            // 1 - (d f)^2 is not zero for points of the curve (d is not a square)
      bigfield::mulmod<8>(inv, dm, m, inv_plus);
      bigfield::mulmod<8>(inv, dp, m, inv_minus);
    };
    auto den = [&](int base, const uint32_t* num, const uint32_t* bb, bool sign, uint32_t* result) {
      bigfield::mulmod<8>(num, sign ? inv_plus : inv_minus, m, result);
      // carry = (b result + (sign ? result : a) - (sign ? a : result)) / p
      bigfield::mul<8, 8>(bb, result, t);
      for (int i = 0; i < 16; i++) t2[i] = i < 8 ? (sign ? result[i] : num[i]) : 0u;
      bigfield::add<16>(t, t2);
      for (int i = 0; i < 16; i++) t2[i] = i < 8 ? (sign ? num[i] : result[i]) : 0u;
      bigfield::sub<16>(t, t2);
      uint32_t rem[8];
      bigfield::divmod<8>(t, m, q, rem);
      R.clear(); R.mac(bb, result); R.add(sign ? result : num, 1); R.add(sign ? num : result, -1);
      R.gadget(base, result, q);
    };
    inner(GADGETS, x1, y2, x2, y1, x3n);
    inner(GADGETS + G, y1, y2, x1, x2, y3n);
    R.op(GADGETS + 2 * G, x1, y1, FOP_MUL, a);
    R.op(GADGETS + 3 * G, x2, y2, FOP_MUL, b);
    R.op(GADGETS + 4 * G, a, b, FOP_MUL, f);
    R.op(GADGETS + 5 * G, f, d_ed25519_d, FOP_MUL, df);
    denominators(df);
    den(GADGETS + 6 * G, x3n, df, true, res);
    bool ok = true;
    if (real)
      for (int k = 0; k < 8; k++) ok = ok && e[E_P_RECORDS + 6 * k] == res[k];
    den(GADGETS + 7 * G, y3n, df, false, res);
    if (real)
      for (int k = 0; k < 8; k++) ok = ok && e[E_P_RECORDS + 6 * (8 + k)] == res[k];
    if (!ok) *bad = 1;
    R.put(IS_REAL, real ? 1u : 0u);
    R.put(SHARD, real ? e[0] : 0u); R.put(CLK, real ? e[1] : 0u); R.put(P_PTR, real ? e[2] : 0u); R.put(Q_PTR, real ? e[3] : 0u);
    for (int k = 0; k < 16; k++) {
      R.write_cols(P_ACCESS + 13 * k, real ? e + E_P_RECORDS + 6 * k : nullptr);
      R.read_cols(Q_ACCESS + 9 * k, real ? e + E_Q_RECORDS + 5 * k : nullptr);
    }
  }
  blk.flush(counts);
}

// ---- EdDecompress (syscall/precompiles/edwards/ed_decompress.rs:39-57, :85-101): x = sqrt((y^2 - 1) / (d y^2 + 1)) from y and a sign bit, one row
// per call, 1566 columns: memory columns of x (written) and y (read), a FieldLtCols of y against p, five FieldOpCols (yy, u, dyy, v, u_div_v), a
// FieldSqrtCols (a FieldOpCols whose result columns hold the even root, the root's FieldLtCols, its low bit), and neg_x. The root is
// a^((p + 3) / 8), times sqrt(-1) when that squares to -a (curves/src/edwards/ed25519.rs:75-113). Padding rows: y = 0.
constexpr int ED_DECOMPRESS_WIDTH = 1566, ED_DECOMPRESS_EVENT_WORDS = 92;
__constant__ uint32_t d_ed25519_sqrt_exp[8] = {0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0fffffffu};
__constant__ uint32_t d_ed25519_sqrt_m1[8] = {0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u};
__global__ __launch_bounds__(bf_threads(8)) void ed_decompress_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                                    uint32_t* __restrict__ out, uint32_t* counts, int* __restrict__ bad) {
  enum { IS_REAL = 0, SHARD = 1, CLK = 2, PTR = 3, SIGN = 4, X_ACCESS = 5, Y_ACCESS = 109, Y_RANGE = 181, YY = 215, U = 403, DYY = 591, V = 779, U_DIV_V = 967,
         X_MULT = 1155, X_RANGE = 1343, X_LSB = 1377, NEG_X = 1378 };
  enum { E_X_RECORDS = 4, E_Y_RECORDS = 52 };
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * ED_DECOMPRESS_EVENT_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const FieldRow<8> R = FieldRow<8>::make(out, height, row, sink, blk.count && real, d_ed25519, blk.scratch);
    const bigfield::Modulus<8>& m = d_ed25519.m;
    int why = 0;
    uint32_t y[8];
    for (int k = 0; k < 8; k++) y[k] = real ? e[E_Y_RECORDS + 5 * k] : 0u;
    if (bigfield::cmp<8>(y, m.p) >= 0) { why = 1; for (int k = 0; k < 8; k++) y[k] = 0; }
    if (real && e[3] > 1) why = 1;
    uint32_t yy[8], u[8], dyy[8], v[8], udv[8], x[8], sq[8], neg[8];
    const uint32_t one[8] = {1}, zero[8] = {0};
    R.lt(Y_RANGE, y);
    R.op(YY, y, y, FOP_MUL, yy);
    R.op(U, yy, one, FOP_SUB, u);
    R.op(DYY, d_ed25519_d, yy, FOP_MUL, dyy);
    R.op(V, one, dyy, FOP_ADD, v);
    R.op(U_DIV_V, u, v, FOP_DIV, udv);
    {      // the even square root of u / v
      uint32_t exponent[8];
#pragma unroll
      for (int i = 0; i < 8; i++) exponent[i] = d_ed25519_sqrt_exp[i];
      bool zero = true;
      for (int i = 0; i < 8; i++) zero = zero && udv[i] == 0;
      if (zero) { for (int i = 0; i < 8; i++) x[i] = 0; } else bigfield::pow<8>(udv, exponent, m, x);
      bigfield::mulmod<8>(x, x, m, sq);
      if (bigfield::cmp<8>(sq, udv) != 0) {
        uint32_t neg_a[8];
        for (int i = 0; i < 8; i++) neg_a[i] = m.p[i];
        bigfield::sub<8>(neg_a, udv);
        if (bigfield::cmp<8>(neg_a, m.p) >= 0) bigfield::sub<8>(neg_a, m.p);
        if (bigfield::cmp<8>(sq, neg_a) == 0) bigfield::mulmod<8>(x, d_ed25519_sqrt_m1, m, x); else why = 2;
      }
      if (x[0] & 1) { uint32_t tmp[8]; for (int i = 0; i < 8; i++) tmp[i] = m.p[i]; bigfield::sub<8>(tmp, x); for (int i = 0; i < 8; i++) x[i] = tmp[i]; }
    }
    // FieldSqrtCols (field_sqrt.rs:34-85): x * x = u_div_v in the multiplication's carry / witness columns, the root itself in its result columns
    R.op(X_MULT, x, x, FOP_MUL, sq);
    R.stage_a(x);
    for (int i = 0; i < ED_LIMBS; i++) R.put(X_MULT + i, (uint32_t)R.A(i));
    if (R.count) lookup(sink, B_AND, x[0], 1);      // the range checks of the product's and the root's bytes: u8_pair_histogram
    R.lt(X_RANGE, x);
    R.put(X_LSB, x[0] & 1);
    R.op(NEG_X, zero, x, FOP_SUB, neg);
    R.put(IS_REAL, real ? 1u : 0u);
    R.put(SHARD, real ? e[0] : 0u); R.put(CLK, real ? e[1] : 0u); R.put(PTR, real ? e[2] : 0u); R.put(SIGN, real ? e[3] : 0u);
    for (int k = 0; k < 8; k++) {
      R.write_cols(X_ACCESS + 13 * k, real ? e + E_X_RECORDS + 6 * k : nullptr);
      R.read_cols(Y_ACCESS + 9 * k, real ? e + E_Y_RECORDS + 5 * k : nullptr);
      if (real && e[E_X_RECORDS + 6 * k] != (e[3] ? neg[k] : x[k])) why = why ? why : 3;
    }
    if (real && why) atomicMax(bad, 16 - why);
  }
  blk.flush(counts);
}

// ---- Short-Weierstrass AddAssign / DoubleAssign (syscall/precompiles/weierstrass/weierstrass_add.rs, weierstrass_double.rs) for Secp256k1,
// Secp256r1, Bn254 (NL = 8) and Bls12381 (NL = 12): nine / eleven FieldOpCols per row. The curve's base field arrives as a kernel argument
// (modulus, its Barrett constant, `a`, the witness offset). Padding rows: an addition's are the operations of the zero inputs; a doubling's
// are those of the point (0, 1) with the dummy write record of weierstrass_double.rs:225-239 on the first word of y.
template <int NL, bool DOUBLE>
__global__ __launch_bounds__(bf_threads(NL)) void weierstrass_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                                   uint32_t* __restrict__ out, uint32_t* counts, int* __restrict__ bad,
                                                                   const CurveField<NL> field) {
  constexpr int W = 2 * NL, G = FieldRow<NL>::G;
  constexpr int P_ACCESS = DOUBLE ? 4 : 5, Q_ACCESS = P_ACCESS + 13 * W, GADGETS = P_ACCESS + 13 * W + (DOUBLE ? 0 : 9 * W);
  constexpr int EV_WORDS = DOUBLE ? 3 + 6 * W : 4 + 11 * W, E_P = DOUBLE ? 3 : 4, E_Q = 4 + 6 * W;
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * EV_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const FieldRow<NL> R = FieldRow<NL>::make(out, height, row, sink, blk.count && real, field, blk.scratch);
    uint32_t px[NL], py[NL], qx[NL], qy[NL];
    bool ok = true;
    for (int k = 0; k < NL; k++) {
      px[k] = real ? e[E_P + 6 * k + 3] : 0u;
      py[k] = real ? e[E_P + 6 * (NL + k) + 3] : (DOUBLE && k == 0 ? 1u : 0u);      // a doubling's padding point is (0, 1)
      qx[k] = real && !DOUBLE ? e[E_Q + 5 * k] : 0u;
      qy[k] = real && !DOUBLE ? e[E_Q + 5 * (NL + k)] : 0u;
    }
    if (bigfield::cmp<NL>(px, field.m.p) >= 0 || bigfield::cmp<NL>(py, field.m.p) >= 0 || bigfield::cmp<NL>(qx, field.m.p) >= 0 ||
        bigfield::cmp<NL>(qy, field.m.p) >= 0) {
      ok = false;
      for (int k = 0; k < NL; k++) px[k] = py[k] = qx[k] = qy[k] = 0;
    }
    uint32_t num[NL], den[NL], slope[NL], sq[NL], sum[NL], x3[NL], dx[NL], prod[NL], y3[NL];
    auto col = [&](int k) { return GADGETS + G * k; };
    if (!DOUBLE) {      // slope_denominator, slope_numerator, slope, slope_squared, p_x_plus_q_x, x3_ins, p_x_minus_x, y3_ins, slope_times_p_x_minus_x
      R.op(col(1), qy, py, FOP_SUB, num);
      R.op(col(0), qx, px, FOP_SUB, den);
      R.op(col(2), num, den, FOP_DIV, slope);
      R.op(col(3), slope, slope, FOP_MUL, sq);
      R.op(col(4), px, qx, FOP_ADD, sum);
      R.op(col(5), sq, sum, FOP_SUB, x3);
      R.op(col(6), px, x3, FOP_SUB, dx);
      R.op(col(8), slope, dx, FOP_MUL, prod);
      R.op(col(7), prod, py, FOP_SUB, y3);
    } else {            // ..., p_x_squared, p_x_squared_times_3, slope_squared, p_x_plus_p_x, x3_ins, p_x_minus_x, y3_ins, slope_times_p_x_minus_x
      uint32_t xx[NL], xx3[NL], three[NL], two[NL];
      for (int k = 0; k < NL; k++) { three[k] = k == 0 ? 3u : 0u; two[k] = k == 0 ? 2u : 0u; }
      R.op(col(3), px, px, FOP_MUL, xx);
      R.op(col(4), xx, three, FOP_MUL, xx3);
      R.op(col(1), field.a, xx3, FOP_ADD, num);
      R.op(col(0), two, py, FOP_MUL, den);
      R.op(col(2), num, den, FOP_DIV, slope);
      R.op(col(5), slope, slope, FOP_MUL, sq);
      R.op(col(6), px, px, FOP_ADD, sum);
      R.op(col(7), sq, sum, FOP_SUB, x3);
      R.op(col(8), px, x3, FOP_SUB, dx);
      R.op(col(10), slope, dx, FOP_MUL, prod);
      R.op(col(9), prod, py, FOP_SUB, y3);
    }
    R.put(0, real ? 1u : 0u);
    R.put(1, real ? e[0] : 0u); R.put(2, real ? e[1] : 0u); R.put(3, real ? e[2] : 0u);
    if (!DOUBLE) R.put(4, real ? e[3] : 0u);
    const uint32_t dummy[6] = {1, 0, 1, 1, 0, 0};
    for (int k = 0; k < W; k++) {
      R.write_cols(P_ACCESS + 13 * k, real ? e + E_P + 6 * k : (DOUBLE && k == NL ? dummy : nullptr));
      if (!DOUBLE) R.read_cols(Q_ACCESS + 9 * k, real ? e + E_Q + 5 * k : nullptr);
      if (real && e[E_P + 6 * k] != (k < NL ? x3[k] : y3[k - NL])) ok = false;
    }
    if (real && !ok) *bad = 1;
  }
  blk.flush(counts);
}

// ---- <Curve>Decompress (syscall/precompiles/weierstrass/weierstrass_decompress.rs:52-79, :121-142, :163-285) for Secp256k1, Secp256r1 (the sign bit is
// y's parity) and Bls12381 (LEX: the bit says y > p - y; three flags and two more FieldLtCols): y = sqrt(x^3 + a x + b), the root
// (x^3 + a x + b)^((p + 1) / 4). x is read at ptr + N, y written at ptr. Padding rows hold the operations of the generator's x.
// Error codes (the lowest wins): 1 sign bit / x not below p, 2 x not on the curve, 3 the y written is not a root, 4 not the root asked for.
template <int NL> struct DecompressCurve { CurveField<NL> f; uint32_t b[NL], generator_x[NL], sqrt_exp[NL]; };
template <int NL, bool LEX>
__global__ __launch_bounds__(bf_threads(NL)) void weierstrass_decompress_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                                              uint32_t* __restrict__ out, uint32_t* counts, int* __restrict__ bad,
                                                                              const DecompressCurve<NL> curve) {
  constexpr int N = 4 * NL, W = NL, G = FieldRow<NL>::G;
  constexpr int X_ACCESS = 5, Y_ACCESS = X_ACCESS + 9 * W, RANGE_X = Y_ACCESS + 13 * W, X_2 = RANGE_X + N + 2, X_3 = X_2 + G, AX_PLUS_B = X_2 + 2 * G,
                X_3_PLUS = X_2 + 3 * G, Y_MULT = X_2 + 4 * G, Y_RANGE = Y_MULT + G, Y_LSB = Y_RANGE + N + 2, NEG_Y = Y_LSB + 1, CHOICE = NEG_Y + G;
  constexpr int EV_WORDS = 4 + 11 * W, E_X = 4, E_Y = 4 + 5 * W;
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * EV_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const FieldRow<NL> R = FieldRow<NL>::make(out, height, row, sink, blk.count && real, curve.f, blk.scratch);
    const bigfield::Modulus<NL>& m = curve.f.m;
    int why = 0;
    uint32_t x[NL];
    for (int k = 0; k < NL; k++) x[k] = real ? e[E_X + 5 * k] : curve.generator_x[k];
    if (bigfield::cmp<NL>(x, m.p) >= 0 || (real && e[3] > 1)) { why = 1; for (int k = 0; k < NL; k++) x[k] = curve.generator_x[k]; }
    uint32_t x2[NL], x3[NL], axb[NL], rhs[NL], y[NL], sq[NL], neg[NL], zero[NL];
    for (int k = 0; k < NL; k++) zero[k] = 0;
    R.lt(RANGE_X, x);
    R.op(X_2, x, x, FOP_MUL, x2);
    R.op(X_3, x2, x, FOP_MUL, x3);
    R.inner_with_one(AX_PLUS_B, curve.f.a, x, curve.b, axb);
    R.op(X_3_PLUS, x3, axb, FOP_ADD, rhs);
    {      // the root rhs^((p + 1) / 4)
      bool zero = true;
      for (int k = 0; k < NL; k++) zero = zero && rhs[k] == 0;
      if (zero) { for (int k = 0; k < NL; k++) y[k] = 0; }
      else if constexpr (NL > 8) bigfield::pow_call<NL>(rhs, curve.sqrt_exp, &m, y);
      else bigfield::pow<NL>(rhs, curve.sqrt_exp, m, y);
      bigfield::mulmod<NL>(y, y, m, sq);
      if (bigfield::cmp<NL>(sq, rhs) != 0) why = why ? why : 2;
    }
    // FieldSqrtCols (field_sqrt.rs:34-85): y * y = rhs in the multiplication's carry / witness columns, the root itself in its result columns
    R.op(Y_MULT, y, y, FOP_MUL, sq);
    R.stage_a(y);
    for (int i = 0; i < N; i++) R.put(Y_MULT + i, (uint32_t)R.A(i));
    if (R.count) lookup(sink, B_AND, y[0], 1);      // the range checks of the product's and the root's bytes: u8_pair_histogram
    R.lt(Y_RANGE, y);
    R.put(Y_LSB, y[0] & 1);
    R.op(NEG_Y, zero, y, FOP_SUB, neg);
    R.put(0, real ? 1u : 0u);
    R.put(1, real ? e[0] : 0u); R.put(2, real ? e[1] : 0u); R.put(3, real ? e[2] : 0u); R.put(4, real ? e[3] : 0u);
    uint32_t d[NL];
    for (int k = 0; k < W; k++) {
      R.read_cols(X_ACCESS + 9 * k, real ? e + E_X + 5 * k : nullptr);
      R.write_cols(Y_ACCESS + 13 * k, real ? e + E_Y + 6 * k : nullptr);
      d[k] = real ? e[E_Y + 6 * k] : 0u;
    }
    if (!real)      // the padding rows show the generator's x as the value read
      for (int i = 0; i < N; i++) R.put(X_ACCESS + 9 * (i / 4) + i % 4, (curve.generator_x[i / 4] >> (8 * (i % 4))) & 0xff);
    const bool wrote_root = bigfield::cmp<NL>(d, y) == 0;
    if (real && !wrote_root && bigfield::cmp<NL>(d, neg) != 0) why = why ? why : 3;
    if (!LEX) {
      if (real && (d[0] & 1) != e[3]) why = why ? why : 4;
    } else {      // LexicographicChoiceCols :196-246: comparison_lt_cols, neg_y_range_check, is_y_eq_sqrt_y_result, when_sqrt_y_res_is_lt, when_neg_y_res_is_lt
      constexpr int CMP = CHOICE, NEG_RANGE = CHOICE + N + 2, FLAGS = CHOICE + 2 * (N + 2);
      if (real && !why) {
        const uint32_t* other = wrote_root ? neg : y;      // p - d
        const int order = bigfield::cmp<NL>(other, d);
        if (order == 0 || (order < 0) != (e[3] != 0)) why = 4;
        R.put(FLAGS, wrote_root ? 1u : 0u);
        R.put(FLAGS + 1, (e[3] != 0) != wrote_root ? 1u : 0u);      // sign: !wrote_root; no sign: wrote_root
        R.put(FLAGS + 2, (e[3] != 0) == wrote_root ? 1u : 0u);
        R.lt(NEG_RANGE, neg);
        if (e[3]) R.lt(CMP, other, d); else R.lt(CMP, d, other);
      } else {
        R.zeros(CHOICE, 2 * (N + 2) + 3);
      }
    }
    if (real && why) atomicMax(bad, 16 - why);
  }
  blk.flush(counts);
}

// ---- U256Field gadgets: Uint256MulMod and U256XU2048Mul. FieldOpCols over U256Field (curves/src/uint256.rs) has 63 witness limbs, because its
// modulus polynomial may have 33 (2^256 = t^32). The identity is x(t) y(t) + addend(t) - result(t) - carry(t) m(t), with m either 32 limbs or
// t^32. Its coefficients are produced one at a time from the top, which is the order the witness recurrence w[k - 1] = van[k] + 256 w[k]
// (operations/field/util.rs:21-66) consumes them; the six operands' limbs are staged in LDS ([word][thread]).
constexpr int U256_THREADS = 256, U256_LDS_WORDS = 48, U256_GADGET = 190;
__host__ __device__ constexpr size_t u256_lds_bytes(bool count) { return (count ? 2 * (size_t)BF_HASH_SLOTS * 4 : 0) + (size_t)U256_THREADS * U256_LDS_WORDS * 4; }
struct U256Operands {
  enum { X = 0, Y = 1, CARRY = 2, MODULUS = 3, RESULT = 4, ADDEND = 5, T = U256_THREADS };
  uint32_t* lds;      // this thread's first word
  __device__ __forceinline__ int32_t byte(int operand, int i) const { return (int32_t)((lds[(size_t)(8 * operand + (i >> 2)) * T] >> (8 * (i & 3))) & 0xff); }
  __device__ __forceinline__ void stage(int operand, const uint32_t* v) const {
#pragma unroll
    for (int k = 0; k < 8; k++) lds[(size_t)(8 * operand + k) * T] = v[k];
  }
  // result(32), carry(32), witness_low(63), witness_high(63) at `base`; their range checks are counted by u8_pair_histogram
  __device__ void gadget(const RowCols& R, int base, bool modulus_is_t32) const {
    constexpr int N = 32, NW = 63;
    int32_t above = 0;
    for (int k = NW; k >= 1; k--) {
      int32_t van = k < N ? byte(ADDEND, k) - byte(RESULT, k) : (modulus_is_t32 ? -byte(CARRY, k - N) : 0);
      for (int i = k < N ? 0 : k - N + 1; i <= (k < N ? k : N - 1); i++) van += byte(X, i) * byte(Y, k - i) - byte(CARRY, i) * byte(MODULUS, k - i);
      above = van + 256 * above;
      const uint32_t shifted = (uint32_t)(above + (1 << 14));
      R.put(base + 2 * N + k - 1, shifted & 0xff);
      R.put(base + 2 * N + NW + k - 1, shifted >> 8);
    }
    for (int i = 0; i < N; i++) { R.put(base + i, (uint32_t)byte(RESULT, i)); R.put(base + N + i, (uint32_t)byte(CARRY, i)); }
  }
};

// Uint256MulMod (syscall/precompiles/uint256/air.rs:57-91, :104-203): x <- x * y mod m, m = 0 standing for 2^256; 480 columns. The modulus changes
// from row to row, so the quotient comes from a binary long division (512 shift-compare-subtract steps) instead of Barrett.
// Error codes: 1 the quotient does not fit 256 bits, 2 the words written to x are not the result.
constexpr int UINT256_MUL_WIDTH = 480, UINT256_MUL_EVENT_WORDS = 132;
__global__ __launch_bounds__(U256_THREADS) void uint256_mul_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                                                 uint32_t* counts, int* __restrict__ bad) {
  enum { SHARD = 0, CLK = 1, X_PTR = 2, Y_PTR = 3, X_MEM = 4, Y_MEM = 108, M_MEM = 180, IS_ZERO = 252, NOT_ZERO = 254, OUTPUT = 255, RANGE = 445, IS_REAL = 479, N = 32 };
  enum { E_X = 4, E_Y = 4 + 48, E_M = 4 + 48 + 40 };
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * UINT256_MUL_EVENT_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const RowCols R{out, height, row, sink, blk.count && real};
    const U256Operands ops{blk.scratch + threadIdx.x};
    uint32_t x[8], y[8], m[8], t[16], q[16], rem[9], zero[8];
    bool no_modulus = true;
    uint32_t byte_sum = 0;
    for (int k = 0; k < 8; k++) {
      x[k] = real ? e[E_X + 6 * k + 3] : 0u;
      y[k] = real ? e[E_Y + 5 * k] : 0u;
      m[k] = real ? e[E_M + 5 * k] : 0u;
      zero[k] = 0;
      no_modulus = no_modulus && m[k] == 0;
      byte_sum += (m[k] & 0xff) + ((m[k] >> 8) & 0xff) + ((m[k] >> 16) & 0xff) + (m[k] >> 24);
    }
    bigfield::mul<8, 8>(x, y, t);
    int why = 0;
    if (no_modulus) {        // modulo 2^256: the low half, carry the high half
      for (int k = 0; k < 8; k++) { rem[k] = t[k]; q[k] = t[8 + k]; q[8 + k] = 0; }
    } else {
      for (int k = 0; k < 9; k++) rem[k] = 0;
#pragma unroll
      for (int l = 15; l >= 0; l--) {
        uint32_t w = t[l], qw = 0;
        for (int b = 0; b < 32; b++) {
          for (int k = 8; k > 0; k--) rem[k] = rem[k] << 1 | rem[k - 1] >> 31;
          rem[0] = rem[0] << 1 | w >> 31;
          w <<= 1;
          const bool ge = rem[8] != 0 || bigfield::cmp<8>(rem, m) >= 0;
          if (ge) { const uint32_t borrow = bigfield::sub<8>(rem, m); rem[8] -= borrow; }
          qw = qw << 1 | (ge ? 1u : 0u);
        }
        q[l] = qw;
      }
      for (int k = 8; k < 16; k++) if (q[k]) why = 1;
    }
    ops.stage(U256Operands::X, x); ops.stage(U256Operands::Y, y); ops.stage(U256Operands::CARRY, q); ops.stage(U256Operands::MODULUS, m);
    ops.stage(U256Operands::RESULT, rem); ops.stage(U256Operands::ADDEND, zero);
    ops.gadget(R, OUTPUT, no_modulus);
    // IsZeroOperation of the sum of the modulus' bytes, modulus_is_not_zero, the range check of the result against the modulus
    R.put(IS_ZERO, real ? small_inverse(byte_sum) : 0u); R.put(IS_ZERO + 1, real && no_modulus ? 1u : 0u);
    R.put(NOT_ZERO, real && !no_modulus ? 1u : 0u);
    {
      int at = -1;
      if (real && !no_modulus)
        for (int i = N - 1; i >= 0 && at < 0; i--)
          if (ops.byte(U256Operands::RESULT, i) < ops.byte(U256Operands::MODULUS, i)) at = i;      // the remainder is below the modulus: the first difference from the top decides
      for (int i = 0; i < N; i++) R.put(RANGE + i, i == at ? 1u : 0u);
      const uint32_t a = at >= 0 ? (uint32_t)ops.byte(U256Operands::RESULT, at) : 0u, b = at >= 0 ? (uint32_t)ops.byte(U256Operands::MODULUS, at) : 0u;
      R.put(RANGE + N, a); R.put(RANGE + N + 1, b);
      if (R.count && at >= 0) lookup(sink, B_LTU, a, b);
    }
    R.put(IS_REAL, real ? 1u : 0u);
    R.put(SHARD, real ? e[0] : 0u); R.put(CLK, real ? e[1] : 0u); R.put(X_PTR, real ? e[2] : 0u); R.put(Y_PTR, real ? e[3] : 0u);
    for (int k = 0; k < 8; k++) {
      R.write_cols(X_MEM + 13 * k, real ? e + E_X + 6 * k : nullptr);
      R.read_cols(Y_MEM + 9 * k, real ? e + E_Y + 5 * k : nullptr);
      R.read_cols(M_MEM + 9 * k, real ? e + E_M + 5 * k : nullptr);
      if (real && e[E_X + 6 * k] != rem[k]) why = why ? why : 2;
    }
    if (real && why) atomicMax(bad, 16 - why);
  }
  blk.flush(counts);
}

// U256XU2048Mul (syscall/precompiles/u256x2048_mul/air.rs:52-87, :101-229): a (256 bits) times b (2048 bits), as eight gadgets chained through their
// carries — a * b_i + carry_{i-1} = result_i + carry_i 2^256 (the schoolbook product, one 256-bit digit of b at a time): 3129 columns. Event
// words: shard, clk, a_ptr, b_ptr, lo_ptr, hi_ptr, the read records of $a2 and $a3, 8 + 64 read records of a and b, 64 + 8 write records
// of lo and hi. Error codes: 1 lo_ptr / hi_ptr are not the registers' values, 2 the words written are not the product.
constexpr int U256X2048_MUL_WIDTH = 3129, U256X2048_MUL_EVENT_WORDS = 808;
__global__ __launch_bounds__(U256_THREADS) void u256x2048_mul_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                                                   uint32_t* counts, int* __restrict__ bad) {
  enum { SHARD = 0, CLK = 1, A_PTR = 2, B_PTR = 3, LO_PTR = 4, HI_PTR = 5, LO_PTR_MEM = 6, HI_PTR_MEM = 15, A_MEM = 24, B_MEM = 96, LO_MEM = 672, HI_MEM = 1504,
         GADGETS = 1608, IS_REAL = 3128, G = U256_GADGET };
  enum { E_LO_PTR = 6, E_HI_PTR = 11, E_A = 16, E_B = 16 + 40, E_LO = 16 + 40 + 320, E_HI = 16 + 40 + 320 + 384 };
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * U256X2048_MUL_EVENT_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const RowCols R{out, height, row, sink, blk.count && real};
    const U256Operands ops{blk.scratch + threadIdx.x};
    uint32_t a[8], b[8], carry[8], t[16], zero[8];
    int why = 0;
    for (int k = 0; k < 8; k++) { a[k] = real ? e[E_A + 5 * k] : 0u; carry[k] = 0; zero[k] = 0; }
    ops.stage(U256Operands::X, a); ops.stage(U256Operands::MODULUS, zero);
    for (int g = 0; g < 8; g++) {
      for (int k = 0; k < 8; k++) b[k] = real ? e[E_B + 5 * (8 * g + k)] : 0u;
      bigfield::mul<8, 8>(a, b, t);
      uint32_t addend[16];
      for (int k = 0; k < 16; k++) addend[k] = k < 8 ? carry[k] : 0u;
      bigfield::add<16>(t, addend);                       // a * b_g + carry < 2^512
      ops.stage(U256Operands::Y, b); ops.stage(U256Operands::ADDEND, carry); ops.stage(U256Operands::RESULT, t); ops.stage(U256Operands::CARRY, t + 8);
      ops.gadget(R, GADGETS + G * g, true);
      for (int k = 0; k < 8; k++) {
        carry[k] = t[8 + k];
        if (real && e[E_LO + 6 * (8 * g + k)] != t[k]) why = 2;
      }
    }
    for (int k = 0; k < 8; k++)
      if (real && e[E_HI + 6 * k] != carry[k]) why = 2;
    if (real && (e[E_LO_PTR] != e[4] || e[E_HI_PTR] != e[5])) why = 1;
    R.put(IS_REAL, real ? 1u : 0u);
    R.put(SHARD, real ? e[0] : 0u); R.put(CLK, real ? e[1] : 0u); R.put(A_PTR, real ? e[2] : 0u); R.put(B_PTR, real ? e[3] : 0u);
    R.put(LO_PTR, real ? e[4] : 0u); R.put(HI_PTR, real ? e[5] : 0u);
    R.read_cols(LO_PTR_MEM, real ? e + E_LO_PTR : nullptr);
    R.read_cols(HI_PTR_MEM, real ? e + E_HI_PTR : nullptr);
    for (int k = 0; k < 8; k++) R.read_cols(A_MEM + 9 * k, real ? e + E_A + 5 * k : nullptr);
    for (int k = 0; k < 64; k++) R.read_cols(B_MEM + 9 * k, real ? e + E_B + 5 * k : nullptr);
    for (int k = 0; k < 64; k++) R.write_cols(LO_MEM + 13 * k, real ? e + E_LO + 6 * k : nullptr);
    for (int k = 0; k < 8; k++) R.write_cols(HI_MEM + 13 * k, real ? e + E_HI + 6 * k : nullptr);
    if (real && why) atomicMax(bad, 16 - why);
  }
  blk.flush(counts);
}

// ---- BooleanCircuitGarble (syscall/precompiles/boolean_circuit_garble/columns.rs:10-35, trace.rs:100-223): 1 + num_gates rows per call, 292 columns; one
// thread per row from its 103-word record (shard, clk, input_address, output_address, is_gate, gate_id, gates_num, pre_check, delta[4], seventeen
// read records, a write record). A gate row is checked against the record before it (same call, next gate, next address, pre_check = the
// conjunction so far — recomputed from that record), so the rows stay independent. Error codes: 1 a header row that does not read the gate
// count / delta, 2 a gate row that does not continue the row before it, 3 a gate type other than 0 / 7, 4 a wrong result, 5 a call cut short.
constexpr int GARBLE_WIDTH = 292, GARBLE_ROW_WORDS = 103;
__device__ __forceinline__ bool garble_gate_ok(const uint32_t* g) {      // g: a gate row's record
  bool ok = true;
  for (int i = 0; i < 4; i++) {
    const uint32_t v = g[12 + 5 * (1 + i)] ^ g[12 + 5 * (5 + i)] ^ g[12 + 5 * (9 + i)] ^ (g[12] ? g[8 + i] : 0u);
    ok = ok && v == g[12 + 5 * (13 + i)];
  }
  return ok;
}
__global__ __launch_bounds__(256) void garble_rows(const uint32_t* __restrict__ records, size_t n_rows, size_t height, uint32_t* __restrict__ out, uint32_t* counts,
                                                   int* __restrict__ bad) {
  enum { SHARD = 0, CLK = 1, IS_REAL = 2, INPUT = 3, OUTPUT = 4, IS_FIRST_ROW = 5, IS_GATE = 6, IS_FIRST_GATE = 7, IS_LAST_GATE = 8, NOT_LAST_GATE = 9, GATE_TYPE = 10,
         GATE_ID = 12, GATES_NUM = 13, DELTA = 14, MEM = 30, RESULT_MEM = 183, AUX1 = 196, AUX2 = 212, AUX3 = 228, IS_EQ = 244, CHECKS = 288 };
  enum { R_INPUT = 2, R_OUTPUT = 3, R_IS_GATE = 4, R_GATE_ID = 5, R_GATES_NUM = 6, R_PRE_CHECK = 7, R_DELTA = 8, R_READS = 12, R_WRITE = 12 + 85 };
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_rows;
    const uint32_t* g = records + row * GARBLE_ROW_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const RowCols R{out, height, row, sink, blk.count && real};
    if (!real) {
      R.zeros(0, GARBLE_WIDTH);
    } else {
      const uint32_t* prev = row ? g - GARBLE_ROW_WORDS : nullptr;
      const bool gate = g[R_IS_GATE] != 0;
      const uint32_t type = gate ? g[R_READS] : 0u;
      const bool last = gate && g[R_GATE_ID] + 1 == g[R_GATES_NUM];
      int why = 0;
      if (!gate) {
        bool ok = g[R_READS] == g[R_GATES_NUM] && g[R_GATES_NUM] != 0;
        for (int k = 0; k < 4; k++) ok = ok && g[R_READS + 5 * (1 + k)] == g[R_DELTA + k];
        if (!ok) why = 1;
        if (prev && !(prev[R_IS_GATE] && prev[R_GATE_ID] + 1 == prev[R_GATES_NUM])) why = why ? why : 5;
        if (row + 1 == n_rows) why = why ? why : 5;
      } else {
        bool chained = prev != nullptr;
        if (chained) {
          chained = prev[0] == g[0] && prev[1] == g[1] && prev[R_GATES_NUM] == g[R_GATES_NUM] && prev[R_OUTPUT] == g[R_OUTPUT] && g[R_GATE_ID] < g[R_GATES_NUM];
          for (int k = 0; k < 4; k++) chained = chained && prev[R_DELTA + k] == g[R_DELTA + k];
          if (g[R_GATE_ID] == 0) chained = chained && !prev[R_IS_GATE] && g[R_INPUT] == prev[R_INPUT] + 20 && g[R_PRE_CHECK] == 1;
          else chained = chained && prev[R_IS_GATE] && prev[R_GATE_ID] + 1 == g[R_GATE_ID] && g[R_INPUT] == prev[R_INPUT] + 68 &&
                         g[R_PRE_CHECK] == (prev[R_PRE_CHECK] && garble_gate_ok(prev) ? 1u : 0u);
        }
        if (!chained) why = 2;
        if (type != 0 && type != 7) why = why ? why : 3;
        if (!last && row + 1 == n_rows) why = why ? why : 5;
      }
      R.put(SHARD, g[0]); R.put(CLK, g[1]); R.put(IS_REAL, 1u); R.put(INPUT, g[R_INPUT]); R.put(OUTPUT, g[R_OUTPUT]);
      R.put(IS_FIRST_ROW, gate ? 0u : 1u); R.put(IS_GATE, gate ? 1u : 0u);
      R.put(IS_FIRST_GATE, gate && g[R_GATE_ID] == 0 ? 1u : 0u); R.put(IS_LAST_GATE, last ? 1u : 0u); R.put(NOT_LAST_GATE, gate && !last ? 1u : 0u);
      R.put(GATE_TYPE, gate && type == 0 ? 1u : 0u); R.put(GATE_TYPE + 1, gate && type != 0 ? 1u : 0u);
      R.put(GATE_ID, gate ? g[R_GATE_ID] : 0u); R.put(GATES_NUM, g[R_GATES_NUM]);
      for (int k = 0; k < 16; k++) R.put(DELTA + k, (g[R_DELTA + k / 4] >> (8 * (k % 4))) & 0xff);
      for (int k = 0; k < 17; k++) R.read_cols(MEM + 9 * k, k < (gate ? 17 : 5) ? g + R_READS + 5 * k : nullptr);
      R.write_cols(RESULT_MEM, last ? g + R_WRITE : nullptr);
      if (!gate) {
        R.zeros(AUX1, GARBLE_WIDTH - AUX1);
      } else {
        uint32_t running = 1, check[4];
        for (int k = 0; k < 4; k++) {
          uint32_t v[3];
          v[0] = g[R_READS + 5 * (1 + k)] ^ g[R_READS + 5 * (5 + k)];
          v[1] = v[0] ^ g[R_READS + 5 * (9 + k)];
          v[2] = v[1] ^ g[R_DELTA + k];
          const uint32_t lhs[3] = {g[R_READS + 5 * (1 + k)], v[0], v[1]}, rhs[3] = {g[R_READS + 5 * (5 + k)], g[R_READS + 5 * (9 + k)], g[R_DELTA + k]};
          for (int a = 0; a < 3; a++)
            for (int c = 0; c < 4; c++) {
              R.put((a == 0 ? AUX1 : a == 1 ? AUX2 : AUX3) + 4 * k + c, (v[a] >> (8 * c)) & 0xff);
              if (R.count) lookup(sink, B_XOR, lhs[a] >> (8 * c), rhs[a] >> (8 * c));
            }
          const uint32_t computed = type ? v[2] : v[1], want = g[R_READS + 5 * (13 + k)];
          uint32_t eq[11];
          is_equal_word_cols(computed, want, eq);
          for (int c = 0; c < 11; c++) R.put(IS_EQ + 11 * k + c, eq[c]);
          running = running && computed == want;
          check[k] = running;
        }
        R.put(CHECKS, check[1]); R.put(CHECKS + 1, check[2]); R.put(CHECKS + 2, check[3]); R.put(CHECKS + 3, check[3] && g[R_PRE_CHECK] ? 1u : 0u);
        if (last && g[R_WRITE] != (check[3] && g[R_PRE_CHECK] ? 1u : 0u)) why = why ? why : 4;
      }
      if (why) atomicMax(bad, 16 - why);
    }
  }
  blk.flush(counts);
}

// ---- SysLinux (syscall/precompiles/sys_linux/columns.rs:20-82, trace.rs:104-233): one Linux syscall per row, 103 columns, from the flattened LinuxEvent
// (shard, clk, a0, a1, v0, syscall_code, the read record of brk / write, the write record of $a3, the write record of HEAP for mmap with
// a0 = 0). Refused (code 1): an event whose v0 / $a3 value is not what the syscall returns, or whose heap does not move by the rounded size.
constexpr int SYS_LINUX_WIDTH = 103, LINUX_EVENT_WORDS = 23;
__global__ __launch_bounds__(256) void sys_linux_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out, uint32_t* counts,
                                                      int* __restrict__ bad) {
  enum { SHARD = 0, CLK = 1, ID = 2, A0 = 3, A1 = 7, RESULT = 11, INOROUT = 15, OUTPUT = 28, D_MMAP = 41, D_MMAP2 = 43, D_CLONE = 45, D_EXIT = 47, D_BRK = 49, D_FCNTL = 51,
         D_READ = 53, D_WRITE = 55, IS_MMAP = 57, D_A0_0 = 58, D_A0_1 = 60, D_A0_2 = 62, D_A1_1 = 64, D_A1_3 = 66, IS_MMAP_A0_0 = 68, IS_FCNTL_A1_1 = 69, IS_FCNTL_A1_3 = 70,
         LO_BITS = 71, HI_BITS = 75, PAGE_ZERO = 79, MMAP_SIZE = 81, SIZE_CARRY = 85, HEAP_ADD = 87, GT = 94, IS_REAL = 102 };
  enum { MMAP = 4210, MMAP2 = 4090, CLONE = 4120, EXIT_GROUP = 4246, BRK = 4045, FCNTL = 4055, READ = 4003, WRITE = 4004 };
  enum { E_A0 = 2, E_A1 = 3, E_V0 = 4, E_CODE = 5, E_READ = 6, E_A3 = 11, E_HEAP = 17 };
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const RowCols R{out, height, row, sink, blk.count && real};
    if (!real) {
      R.zeros(0, SYS_LINUX_WIDTH);
    } else {
      const uint32_t* e = events + row * LINUX_EVENT_WORDS;
      const uint32_t code = e[E_CODE], a0 = e[E_A0], a1 = e[E_A1];
      const bool mmap = code == MMAP || code == MMAP2, fd = a0 <= 2;
      uint32_t r[SYS_LINUX_WIDTH];
      for (int c = 0; c < SYS_LINUX_WIDTH; c++) r[c] = 0;
      uint32_t v0 = 0, a3 = 0;
      if (code == BRK) v0 = a0 > e[E_READ] ? a0 : e[E_READ];
      else if (mmap) v0 = a0 == 0 ? e[E_HEAP + 3] : a0;
      else if (code == CLONE) v0 = 1;
      else if (code == FCNTL) {
        if (a1 == 3) v0 = a0 == 0 ? 0u : fd ? 1u : 0xffffffffu;
        else if (a1 == 1) v0 = fd ? a0 : 0xffffffffu;
        else v0 = 0xffffffffu;
        a3 = v0 == 0xffffffffu ? 9u : 0u;
      } else if (code == READ) { v0 = a0 == 0 ? 0u : 0xffffffffu; a3 = a0 == 0 ? 0u : 9u; }
      else if (code == WRITE) v0 = e[E_READ];
      bool ok = v0 == e[E_V0] && a3 == e[E_A3];
      auto lookups = [&](uint32_t v) { if (R.count) { lookup(sink, B_U8RANGE, v, v >> 8); lookup(sink, B_U8RANGE, v >> 16, v >> 24); } };
      word(r + A0, a0); word(r + A1, a1); word(r + RESULT, e[E_V0]);
      r[SHARD] = e[0] % kb::P; r[CLK] = e[1] % kb::P; r[ID] = code % kb::P; r[IS_REAL] = 1;
      const uint32_t sid = code % kb::P, a0f = a0 % kb::P, a1f = a1 % kb::P;
      is_zero_cols(sid, MMAP, r + D_MMAP); is_zero_cols(sid, MMAP2, r + D_MMAP2); is_zero_cols(sid, CLONE, r + D_CLONE); is_zero_cols(sid, EXIT_GROUP, r + D_EXIT);
      is_zero_cols(sid, BRK, r + D_BRK); is_zero_cols(sid, FCNTL, r + D_FCNTL); is_zero_cols(sid, READ, r + D_READ); is_zero_cols(sid, WRITE, r + D_WRITE);
      r[IS_MMAP] = mmap;
      is_zero_cols(a0f, 0, r + D_A0_0); is_zero_cols(a0f, 1, r + D_A0_1); is_zero_cols(a0f, 2, r + D_A0_2);
      is_zero_cols(a1f, 1, r + D_A1_1); is_zero_cols(a1f, 3, r + D_A1_3);
      r[IS_MMAP_A0_0] = mmap && a0 == 0;
      r[IS_FCNTL_A1_1] = code == FCNTL && a1 == 1;
      r[IS_FCNTL_A1_3] = code == FCNTL && a1 == 3;
      if (code == BRK) {        // GtColsBytes::populate (operations/cmp.rs:34-93) of a0 against the BRK register
        const uint32_t bb = e[E_READ];
        uint32_t res = 0, a_byte = 0, b_byte = 0;
        bool flagged = false;
        for (int k = 3; k >= 0 && !flagged; k--) {
          const uint32_t x = (a0 >> (8 * k)) & 0xff, y = (bb >> (8 * k)) & 0xff;
          if (x != y) { r[GT + k] = 1; a_byte = x; b_byte = y; res = x > y; flagged = true; }
        }
        r[GT + 4] = a_byte; r[GT + 5] = b_byte; r[GT + 6] = res; r[GT + 7] = flagged;
        if (R.count) { lookup(sink, B_LTU, b_byte, a_byte); if (flagged) lookup(sink, B_LTU, a_byte, b_byte); }
        lookups(a0); lookups(bb);
      } else if (mmap) {
        lookups(a0); lookups(a1);
        const uint32_t byte1 = (a1 >> 8) & 0xff, lo = byte1 & 15, hi = byte1 >> 4;
        for (int bit = 0; bit < 4; bit++) { r[LO_BITS + bit] = (lo >> bit) & 1; r[HI_BITS + bit] = (hi >> bit) & 1; }
        const uint32_t page_off = a1 & 0xfff, upper = (a1 >> 12) << 12;
        is_zero_cols(page_off, 0, r + PAGE_ZERO);
        if (a0 == 0) {
          const uint32_t size = page_off == 0 ? upper : upper + 0x1000;
          word(r + MMAP_SIZE, size);
          lookups(size);
          if (page_off != 0 && hi == 15) { r[SIZE_CARRY] = 1; if (((a1 >> 16) & 0xff) == 255) r[SIZE_CARRY + 1] = 1; }
          const uint32_t old_heap = e[E_HEAP + 3], sum = old_heap + size;      // AddOperation::populate (operations/add.rs:23-57)
          ok = ok && e[E_HEAP] == sum;
          word(r + HEAP_ADD, sum);
          uint32_t carry = 0;
          for (int k = 0; k < 3; k++) { carry = (((old_heap >> (8 * k)) & 0xff) + ((size >> (8 * k)) & 0xff) + carry) >> 8; r[HEAP_ADD + 4 + k] = carry; }
          lookups(old_heap); lookups(size); lookups(sum);
        }
      }
      for (int c = 0; c < SYS_LINUX_WIDTH; c++)
        if (!(c >= INOROUT && c < OUTPUT + 13)) R.put(c, r[c]);
      // inorout: the read of BRK / $a2 (the previous value is the value), or the write of HEAP; output: the write of $a3
      if (code == BRK || code == WRITE) {
        uint32_t w4[4];
        word(w4, e[E_READ]);
        for (int c = 0; c < 4; c++) R.put(INOROUT + c, w4[c]);
        R.read_cols(INOROUT + 4, e + E_READ);
      } else if (mmap && a0 == 0) {
        R.write_cols(INOROUT, e + E_HEAP);
      } else {
        R.zeros(INOROUT, 13);
      }
      R.write_cols(OUTPUT, e + E_A3);
      if (!ok) *bad = 1;
    }
  }
  blk.flush(counts);
}

// ---- Field tower (syscall/precompiles/fptower/fp.rs, fp2_addsub.rs, fp2_mul.rs) over the base field of Bn254 (NL = 8) or Bls12381 (NL = 12):
// KIND 0 FpOp (one FieldOpCols, the operation — FieldOperation as a word: Add 0, Mul 1, Sub 2 — per event), 1 Fp2AddSub (two), 2 Fp2Mul (four
// products, a difference, a sum). x is overwritten at clk + 1, y is read at clk. Padding rows: zero inputs with is_add set.
template <int NL, int KIND>
__global__ __launch_bounds__(bf_threads(NL)) void fp_tower_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                                uint32_t* __restrict__ out, uint32_t* counts, int* __restrict__ bad,
                                                                const CurveField<NL> field) {
  constexpr int G = FieldRow<NL>::G, W = KIND == 0 ? NL : 2 * NL, HEAD = KIND == 0 ? 8 : KIND == 1 ? 6 : 5;
  constexpr int X_ACCESS = HEAD, Y_ACCESS = HEAD + 13 * W, GADGETS = HEAD + 22 * W, E_HEAD = KIND == 2 ? 4 : 5, EV_WORDS = E_HEAD + 11 * W;
  extern __shared__ uint32_t bf_lds[];
  const BfBlock blk(bf_lds, counts);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < height) {
    const bool real = row < n_events;
    const uint32_t* e = events + row * EV_WORDS;
    const LookupSink sink{blk.hkeys, blk.hvals, BF_HASH_SLOTS - 1, counts};
    const FieldRow<NL> R = FieldRow<NL>::make(out, height, row, sink, blk.count && real, field, blk.scratch);
    const uint32_t op = KIND == 2 ? 1u : (real ? e[4] : 0u);
    bool ok = KIND == 0 ? op <= 2 : (KIND == 1 ? op == 0 || op == 2 : true);
    const int fop = op == 0 ? FOP_ADD : op == 1 ? FOP_MUL : FOP_SUB;
    uint32_t x0[NL], x1[NL], y0[NL], y1[NL];
    for (int k = 0; k < NL; k++) {
      x0[k] = real ? e[E_HEAD + 6 * k + 3] : 0u;
      y0[k] = real ? e[E_HEAD + 6 * W + 5 * k] : 0u;
      x1[k] = real && KIND != 0 ? e[E_HEAD + 6 * (NL + k) + 3] : 0u;
      y1[k] = real && KIND != 0 ? e[E_HEAD + 6 * W + 5 * (NL + k)] : 0u;
    }
    if (bigfield::cmp<NL>(x0, field.m.p) >= 0 || bigfield::cmp<NL>(y0, field.m.p) >= 0 || bigfield::cmp<NL>(x1, field.m.p) >= 0 ||
        bigfield::cmp<NL>(y1, field.m.p) >= 0) {
      ok = false;
      for (int k = 0; k < NL; k++) x0[k] = x1[k] = y0[k] = y1[k] = 0;
    }
    uint32_t out0[NL], out1[NL];
    auto col = [&](int k) { return GADGETS + G * k; };
    if (KIND == 0) {
      R.op(col(0), x0, y0, fop, out0);
    } else if (KIND == 1) {
      R.op(col(0), x0, y0, fop, out0);
      R.op(col(1), x1, y1, fop, out1);
    } else {            // a0_mul_b0, a1_mul_b1, a0_mul_b1, a1_mul_b0, c0, c1
      uint32_t a0b0[NL], a1b1[NL], a0b1[NL], a1b0[NL];
      R.op(col(0), x0, y0, FOP_MUL, a0b0);
      R.op(col(1), x1, y1, FOP_MUL, a1b1);
      R.op(col(2), x0, y1, FOP_MUL, a0b1);
      R.op(col(3), x1, y0, FOP_MUL, a1b0);
      R.op(col(4), a0b0, a1b1, FOP_SUB, out0);
      R.op(col(5), a0b1, a1b0, FOP_ADD, out1);
    }
    R.put(0, real ? 1u : 0u); R.put(1, real ? e[0] : 0u); R.put(2, real ? e[1] : 0u);
    if (KIND == 0) {
      R.put(3, !real || op == 0 ? 1u : 0u); R.put(4, real && op == 2 ? 1u : 0u); R.put(5, real && op == 1 ? 1u : 0u);
      R.put(6, real ? e[2] : 0u); R.put(7, real ? e[3] : 0u);
    } else if (KIND == 1) {
      R.put(3, !real || op == 0 ? 1u : 0u); R.put(4, real ? e[2] : 0u); R.put(5, real ? e[3] : 0u);
    } else {
      R.put(3, real ? e[2] : 0u); R.put(4, real ? e[3] : 0u);
    }
    for (int k = 0; k < W; k++) {
      R.write_cols(X_ACCESS + 13 * k, real ? e + E_HEAD + 6 * k : nullptr);
      R.read_cols(Y_ACCESS + 9 * k, real ? e + E_HEAD + 6 * W + 5 * k : nullptr);
      if (real && e[E_HEAD + 6 * k] != (k < NL ? out0[k] : out1[k - NL])) ok = false;
    }
    if (real && !ok) *bad = 1;
  }
  blk.flush(counts);
}

// recursion ExpReverseBitsLen chip (crates/recursion/core/src/chips/exp_reverse_bits.rs:175-226): one thread walks one event's bits —
// accum_i = accum_{i-1}^2 * (bit_i ? x : 1) — and writes its rows (x, bit, prev_accum^2, that times the multiplier, accum, accum^2,
// multiplier); Montgomery words in and out. offsets[e] .. offsets[e + 1] are event e's rows; the rest of the matrix is zeroed first.
constexpr int EXP_REVERSE_BITS_WIDTH = 7;
__global__ void exp_reverse_bits_rows(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ bits, const uint32_t* __restrict__ offsets,
                                      size_t n_events, size_t rows, size_t height, uint32_t* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_events) return;
  const uint32_t x = bases[e];
  uint32_t accum = kb::ONE;
  // `rows` bits were handed over and the matrix has `height` rows: offsets that say otherwise (they may come straight from HBM, unread by
  // the host) neither read nor write outside
  const size_t end = offsets[e + 1] < rows ? offsets[e + 1] : rows;
  for (size_t i = offsets[e]; i < end && i < height; i++) {
    const uint32_t bit = bits[i], prev_sq = kb::mul(accum, accum), mult = bit == kb::ONE ? x : kb::ONE;
    accum = kb::mul(prev_sq, mult);
    const uint32_t r[EXP_REVERSE_BITS_WIDTH] = {x, bit, prev_sq, accum, accum, kb::mul(accum, accum), mult};
#pragma unroll
    for (int c = 0; c < EXP_REVERSE_BITS_WIDTH; c++) out[(size_t)c * height + i] = r[c];
  }
}

// recursion Poseidon2Skinny chip (crates/recursion/core/src/chips/poseidon2_skinny/trace.rs:62-118): the same permutation laid out
// as eleven rows of 28 columns (the state entering each step; on the internal-rounds row also lane 0 after each of the first twelve
// internal rounds). One thread per permutation; rows past 11 * n_events are zero (the matrix is cleared first).
constexpr int SKINNY_WIDTH = 28, SKINNY_ROWS = 11;
__global__ __launch_bounds__(THREADS) void poseidon2_skinny_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                                 uint32_t* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_events) return;
  uint32_t s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = events[e * 32 + i];
  const size_t row0 = e * SKINNY_ROWS;
  auto put_state = [&](int row) {
#pragma unroll
    for (int i = 0; i < 16; i++) out[(size_t)i * height + row0 + row] = s[i];
  };
  put_state(0);
  wide_external_layer(s);
  for (int row = 1; row <= 10; row++) {
    put_state(row);
    if (row == 10) break;
    if (row == 5) {
      for (int r = 0; r < 13; r++) {
        s[0] = wide_sbox(kb::add(s[0], p2::d_rc_int[r] + kb::P));
        uint32_t sum = s[0];
#pragma unroll
        for (int i = 1; i < 16; i++) sum = kb::add(sum, s[i]);
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = kb::add(kb::mul(s[i], p2::d_diag[i]), sum);
        if (r < 12) out[(size_t)(16 + r) * height + row0 + 5] = s[0];
      }
    } else {
      const int rd = row < 5 ? row - 1 : row - 2;
#pragma unroll
      for (int i = 0; i < 16; i++) s[i] = wide_sbox(kb::add(s[i], p2::d_rc_ext[rd][i] + kb::P));
      wide_external_layer(s);
    }
  }
}

// ByteChip::generate_trace: out = to_field(counts + extra); extra (may be null) holds the row-major plain counts of the
// chips whose dependencies stay on the host
__global__ void byte_mults_finish(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ extra_row_major,
                                  uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t v = counts[i];
  if (extra_row_major) v += extra_row_major[(i & (BYTE_ROWS - 1)) * NUM_BYTE_OPS + (i >> 16)];
  out[i] = kb::to_monty(v);
}

// ByteChip::trace() (bytes/mod.rs:31-104): row (b << 8 | c) of the preprocessed table, column-major, Montgomery
__global__ void byte_table(uint32_t* out) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= BYTE_ROWS) return;
  const uint32_t b = row >> 8, c = row & 0xff, k = c & 7;
  const uint32_t v[BYTE_PREP_COLS] = {b, c, b & c, b | c, b ^ c, ~(b | c) & 0xff, (b << k) & 0xff, b >> k, b & ((1u << k) - 1),
                                      (uint32_t)(b < c), b >> 7, (b << 8) + c};
#pragma unroll
  for (int j = 0; j < BYTE_PREP_COLS; j++) out[(size_t)j * BYTE_ROWS + row] = kb::to_monty(v[j]);
}

// ---- Global chip (crates/core/machine/src/global/mod.rs): GlobalLookupEvents of eight words (message[7]; is_receive, kind in the
// bytes of the eighth), 99 columns (:54-64). Three steps: (1) global_point_rows — one thread per row maps its message to a curve point
// (GlobalLookupOperation::populate, operations/global_lookup.rs:26-92), writes columns 0..63 and the point into a scan buffer whose
// element 0 is the start digest; (2) an inclusive scan of that buffer under the curve's complete addition (the reference's parallel
// scan, mod.rs:163-167) — short chunks of SCAN_CHUNK per thread (an addition is ~800 dependent multiplications: threads, not long chunks), the chunk sums scanned recursively, one block at the top; (3)
// global_accum_rows — columns 64..98 from the prefix sums (operations/global_accumulation.rs:75-113). Byte lookups: U16Range(message[0])
// per event (mod.rs:75-95): message[0] is the shard number, one or two distinct values, so each wave adds its leader's count once.
constexpr int GLOBAL_WIDTH = 99, SCAN_CHUNK = 8, SCAN_BLOCK = 1024, POINT_WORDS = 16;
namespace globalcols {
enum { MESSAGE = 0, KIND = 7, OFFSET_BITS = 8, X = 16, Y = 23, Y6_BITS = 30, RC_WITNESS = 60, IS_RECEIVE = 61, IS_SEND = 62, IS_REAL = 63,
       INITIAL = 64, SUM_CHECKER = 78, CUMULATIVE = 85 };
}
__constant__ uint32_t d_global_consts[28];   // start digest x, y (septic_digest.rs:9-14), dummy point x, y (septic_curve.rs:18-38); Montgomery
enum : uint32_t { GLOBAL_ERR_NO_POINT = 1, GLOBAL_ERR_INFINITY = 2, GLOBAL_ERR_EQUAL_X = 4, GLOBAL_ERR_NOT_U16 = 8 };
__device__ __forceinline__ septic::Point load_point(const uint32_t* p) {
  septic::Point r;
#pragma unroll
  for (int k = 0; k < 7; k++) { r.x.c[k] = p[k]; r.y.c[k] = p[7 + k]; }
  r.inf = p[14];
  return r;
}
__device__ __forceinline__ void store_point(uint32_t* p, const septic::Point& v) {
#pragma unroll
  for (int k = 0; k < 7; k++) { p[k] = v.x.c[k]; p[7 + k] = v.y.c[k]; }
  p[14] = v.inf;
  p[15] = 0;
}
__global__ void global_point_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                  uint32_t* __restrict__ points, uint32_t* __restrict__ counts, uint32_t* __restrict__ err) {
  using namespace globalcols;
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row == 0) {
    septic::Point s;
    for (int k = 0; k < 7; k++) { s.x.c[k] = d_global_consts[k]; s.y.c[k] = d_global_consts[7 + k]; }
    s.inf = 0;
    store_point(points, s);
  }
  if (row >= height) return;
  auto put = [&](int col, uint32_t monty) { out[(size_t)col * height + row] = monty; };
  if (row >= n_events) {   // populate_dummy: the dummy point, everything else zero
    for (int c = 0; c < INITIAL; c++) put(c, (c >= X && c < X + 14) ? d_global_consts[14 + c - X] : 0u);
    return;
  }
  const uint32_t* e = events + row * 8;
  const uint32_t is_receive = e[7] & 0xff, kind = (e[7] >> 8) & 0xff;
  if (e[0] >> 16) atomicOr(err, GLOBAL_ERR_NOT_U16);     // message[0] is range-checked as a u16 (global/mod.rs): found here, the events may never have been on the host
  septic::S7 m;
  for (int k = 0; k < 7; k++) { m.c[k] = kb::to_monty(e[k]); put(MESSAGE + k, m.c[k]); }
  m.c[0] = kb::add(m.c[0], kb::to_monty(kind << 16));
  put(KIND, kb::to_monty(kind));
  septic::Point pt;
  uint32_t offset = 0;
  if (!septic::lift_x(m, septic::d_frob, &pt, &offset)) {
    atomicOr(err, GLOBAL_ERR_NO_POINT);
    pt.x = m; pt.y = septic::s_zero(); pt.inf = 0;
  }
  if (!is_receive) pt.y = septic::s_neg(pt.y);
  for (int k = 0; k < 8; k++) put(OFFSET_BITS + k, (offset >> k) & 1 ? kb::ONE : 0u);
  for (int k = 0; k < 7; k++) { put(X + k, pt.x.c[k]); put(Y + k, pt.y.c[k]); }
  const uint32_t y6 = kb::from_monty(pt.y.c[6]);
  const uint32_t rc = is_receive ? y6 - 1 : y6 - (kb::P + 1) / 2;
  uint32_t top = 0;
  for (int k = 0; k < 30; k++) {
    const uint32_t bit = (rc >> k) & 1;
    put(Y6_BITS + k, bit ? kb::ONE : 0u);
    if (k >= 23) top += bit;
  }
  put(RC_WITNESS, kb::to_monty(small_inverse(kb::P + top - 7)));
  put(IS_RECEIVE, is_receive ? kb::ONE : 0u);
  put(IS_SEND, is_receive ? 0u : kb::ONE);
  put(IS_REAL, kb::ONE);
  store_point(points + (row + 1) * POINT_WORDS, pt);
  // U16Range(message[0]): lanes that agree with the wave's first active lane are counted with one atomic
  const uint32_t v = e[0] & 0xffff;
  const uint32_t lead = __builtin_amdgcn_readfirstlane(v);
  const unsigned long long same = __ballot(v == lead);
  if (v != lead) atomicAdd(counts + B_U16RANGE * BYTE_ROWS + v, 1u);
  else if ((same & ((1ull << __lane_id()) - 1)) == 0) atomicAdd(counts + B_U16RANGE * BYTE_ROWS + v, (uint32_t)__popcll(same));
}
// sums[t] = pts[t * chunk] + ... (chunk consecutive points)
__global__ void global_scan_reduce(const uint32_t* __restrict__ pts, size_t n, uint32_t* __restrict__ sums, size_t n_sums) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_sums) return;
  const size_t lo = t * SCAN_CHUNK, hi = lo + SCAN_CHUNK < n ? lo + SCAN_CHUNK : n;
  septic::Point acc = load_point(pts + lo * POINT_WORDS);
  for (size_t i = lo + 1; i < hi; i++) acc = septic::add_complete(acc, load_point(pts + i * POINT_WORDS), septic::d_frob);
  store_point(sums + t * POINT_WORDS, acc);
}
// inclusive scan of n <= SCAN_BLOCK points in place, one block (Hillis-Steele over LDS)
__global__ void __launch_bounds__(SCAN_BLOCK) global_scan_block(uint32_t* __restrict__ pts, size_t n) {
  __shared__ uint32_t lds[SCAN_BLOCK * POINT_WORDS];
  const size_t t = threadIdx.x;
  septic::Point mine;
  if (t < n) { mine = load_point(pts + t * POINT_WORDS); store_point(lds + t * POINT_WORDS, mine); }
  __syncthreads();
  for (size_t d = 1; d < n; d <<= 1) {
    septic::Point left;
    const bool on = t < n && t >= d;
    if (on) left = load_point(lds + (t - d) * POINT_WORDS);
    __syncthreads();
    if (on) { mine = septic::add_complete(left, mine, septic::d_frob); store_point(lds + t * POINT_WORDS, mine); }
    __syncthreads();
  }
  if (t < n) store_point(pts + t * POINT_WORDS, mine);
}
// pts[i] <- (scanned sums of the chunks before i's) + pts[lo..=i], in place
__global__ void global_scan_apply(uint32_t* __restrict__ pts, size_t n, const uint32_t* __restrict__ sums, size_t n_sums) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_sums) return;
  const size_t lo = t * SCAN_CHUNK, hi = lo + SCAN_CHUNK < n ? lo + SCAN_CHUNK : n;
  septic::Point acc;
  acc.inf = 1;
  acc.x = acc.y = septic::s_zero();
  if (t) acc = load_point(sums + (t - 1) * POINT_WORDS);
  for (size_t i = lo; i < hi; i++) {
    acc = septic::add_complete(acc, load_point(pts + i * POINT_WORDS), septic::d_frob);
    store_point(pts + i * POINT_WORDS, acc);
  }
}
__global__ void global_accum_rows(const uint32_t* __restrict__ prefix, size_t n_events, size_t height, uint32_t* __restrict__ out,
                                  uint32_t* __restrict__ err) {
  using namespace globalcols;
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= height) return;
  auto put = [&](int col, uint32_t monty) { out[(size_t)col * height + row] = monty; };
  const bool real = row < n_events;
  const septic::Point initial = load_point(prefix + (real ? row : n_events) * POINT_WORDS);
  septic::Point total = initial;
  septic::S7 checker = septic::s_zero();
  if (real) {
    total = load_point(prefix + (row + 1) * POINT_WORDS);
    uint32_t same = 0;
    for (int k = 0; k < 7; k++) same |= initial.x.c[k] ^ out[(size_t)(X + k) * height + row];
    if (same == 0) atomicOr(err, GLOBAL_ERR_EQUAL_X);    // the AIR's addition is the incomplete one
  } else {
    septic::Point dummy;
    for (int k = 0; k < 7; k++) { dummy.x.c[k] = d_global_consts[14 + k]; dummy.y.c[k] = d_global_consts[21 + k]; }
    checker = septic::sum_checker_x(initial, dummy, initial);
  }
  if (initial.inf | total.inf) atomicOr(err, GLOBAL_ERR_INFINITY);
  for (int k = 0; k < 7; k++) {
    put(INITIAL + k, initial.x.c[k]); put(INITIAL + 7 + k, initial.y.c[k]);
    put(SUM_CHECKER + k, checker.c[k]);
    put(CUMULATIVE + k, total.x.c[k]); put(CUMULATIVE + 7 + k, total.y.c[k]);
  }
}

}  // namespace tracegen
