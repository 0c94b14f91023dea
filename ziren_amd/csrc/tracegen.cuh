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
// Values are stored in Montgomery form, the in-memory form of the reference's KoalaBear (RowMajorMatrix<KoalaBear>).
#pragma once
#include "kb31.cuh"

namespace tracegen {

struct AluEvent {  // #[repr(C)] AluEvent, crates/core/executor/src/events/instr.rs:10-26
  uint32_t pc, next_pc;
  uint32_t opcode;  // u8 + three bytes of padding; only the low byte is meaningful
  uint32_t hi, a, b, c;
};

// crates/core/executor/src/opcode.rs:26-48
enum : uint32_t { ADD = 0, SUB = 1, SLL = 9, SRL = 10, SRA = 11, ROR = 12, SLT = 13, SLTU = 14, AND = 15, OR = 16, XOR = 17, NOR = 18 };
enum Chip { ADD_SUB = 0, BITWISE = 1, LT = 2, SHIFT_LEFT = 3, SHIFT_RIGHT = 4, NUM_CHIPS = 5 };

__host__ __device__ constexpr int chip_width(int chip) {
  return chip == ADD_SUB ? 19 : chip == BITWISE ? 18 : chip == LT ? 32 : chip == SHIFT_LEFT ? 44 : chip == SHIFT_RIGHT ? 67 : 0;
}

constexpr int THREADS = 256;

__device__ __forceinline__ uint32_t fe(uint32_t v) { return kb::to_monty(v); }       // from_canonical_u32 (any u32)
__device__ __forceinline__ uint32_t fbool(bool b) { return b ? kb::ONE : 0u; }
__device__ __forceinline__ void word(uint32_t* dst, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) dst[i] = fe((v >> (8 * i)) & 0xff);
}

template <int CHIP> __device__ __forceinline__ void event_row(const AluEvent& e, uint32_t* r);
template <int CHIP> __device__ __forceinline__ void padding_row(uint32_t* r) {}

template <> __device__ __forceinline__ void event_row<ADD_SUB>(const AluEvent& e, uint32_t* r) {
  const bool is_add = e.opcode == ADD;
  const uint32_t op1 = is_add ? e.b : e.a, op2 = e.c;
  r[0] = fe(e.pc);
  r[1] = fe(e.next_pc);
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
  r[0] = fe(e.pc);
  r[1] = fe(e.next_pc);
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
  r[PC] = fe(e.pc);
  r[NEXT_PC] = fe(e.next_pc);
  r[IS_SLT] = fbool(slt);
  r[IS_SLTU] = fbool(e.opcode == SLTU);
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + C, e.c);
  r[B_MASKED] = fe((e.b >> 24) & 0x7f);
  r[C_MASKED] = fe((e.c >> 24) & 0x7f);
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
  r[NOT_EQ_INV] = top >= 0 ? kb::inv(kb::sub(fe(b_byte), fe(c_byte))) : 0u;
  r[CMP_BYTES] = fe(b_byte);
  r[CMP_BYTES + 1] = fe(c_byte);
  const uint32_t msb_b = e.b >> 31, msb_c = e.c >> 31;
  r[MSB_B] = fbool(msb_b);
  r[MSB_C] = fbool(msb_c);
  r[BIT_B] = fbool(msb_b && slt);
  r[BIT_C] = fbool(msb_c && slt);
  r[IS_SIGN_EQ] = fbool(!slt || msb_b == msb_c);
}

template <> __device__ __forceinline__ void event_row<SHIFT_LEFT>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, A = 2, B = 6, C = 10, C_LSB = 14, BY_BITS = 22, MULT = 30, RESULT = 31, CARRY = 35, BY_BYTES = 39, IS_REAL = 43 };
  r[PC] = fe(e.pc);
  r[NEXT_PC] = fe(e.next_pc);
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + C, e.c);
  const uint32_t nbits = e.c & 7, nbytes = (e.c & 31) >> 3;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    r[C_LSB + i] = fbool((e.c >> i) & 1);
    r[BY_BITS + i] = fbool(nbits == (uint32_t)i);
  }
  r[MULT] = fe(1u << nbits);
  // b * 2^nbits byte by byte: limb i keeps 8 bits, the rest carries into limb i+1
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t v = (((e.b >> (8 * i)) & 0xff) << nbits) + carry;
    carry = v >> 8;
    r[RESULT + i] = fe(v & 0xff);
    r[CARRY + i] = fe(carry);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) r[BY_BYTES + i] = fbool(nbytes == (uint32_t)i);
  r[IS_REAL] = kb::ONE;
}
template <> __device__ __forceinline__ void padding_row<SHIFT_LEFT>(uint32_t* r) {
  r[22] = kb::ONE;  // shift_by_n_bits[0]
  r[30] = kb::ONE;  // bit_shift_multiplier
  r[39] = kb::ONE;  // shift_by_n_bytes[0]
}

template <> __device__ __forceinline__ void event_row<SHIFT_RIGHT>(const AluEvent& e, uint32_t* r) {
  enum { PC = 0, NEXT_PC = 1, B = 2, C = 6, BY_BITS = 10, BY_BYTES = 18, BYTE_RES = 22, BIT_RES = 30, SHR_CARRY = 38, SHR_SHIFTED = 46,
         B_MSB = 54, C_LSB = 55, IS_SRL = 63, IS_ROR = 64, IS_SRA = 65, IS_REAL = 66 };
  r[PC] = fe(e.pc);
  r[NEXT_PC] = fe(e.next_pc);
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
    r[BYTE_RES + i] = fe(byte);
    r[SHR_CARRY + i] = fe(carry);
    r[SHR_SHIFTED + i] = fe(shifted);
    r[BIT_RES + i] = fe((shifted + (last_carry << (8 - nbits))) & 0xff);
    last_carry = carry;
  }
  r[B_MSB] = fbool(e.b >> 31);
  r[IS_SRL] = fbool(e.opcode == SRL);
  r[IS_ROR] = fbool(e.opcode == ROR);
  r[IS_SRA] = fbool(e.opcode == SRA);
  r[IS_REAL] = kb::ONE;
}
template <> __device__ __forceinline__ void padding_row<SHIFT_RIGHT>(uint32_t* r) {
  r[10] = kb::ONE;  // shift_by_n_bits[0]
  r[18] = kb::ONE;  // shift_by_n_bytes[0]
}

// events: n_events records of seven words; out: column-major, `height` rows. grid = height / THREADS.
template <int CHIP>
__global__ __launch_bounds__(THREADS) void alu_rows(const uint32_t* __restrict__ events, size_t n_events, size_t height,
                                                    uint32_t* __restrict__ out) {
  constexpr int W = chip_width(CHIP);
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= height) return;
  uint32_t r[W];
#pragma unroll
  for (int c = 0; c < W; c++) r[c] = 0;
  if (row < n_events) {
    const uint32_t* p = events + row * 7;
    AluEvent e{p[0], p[1], p[2] & 0xff, p[3], p[4], p[5], p[6]};
    event_row<CHIP>(e, r);
  } else {
    padding_row<CHIP>(r);
  }
#pragma unroll
  for (int c = 0; c < W; c++) out[(size_t)c * height + row] = r[c];
}

}  // namespace tracegen
