// TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md section 2): CPU restatement of the reference's
// trace generation for the ALU chips that consume `AluEvent`s. Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may call it; the product path is ziren_amd/csrc/tracegen.cuh.
//
// PARITY UNPINNED: the reference holds no golden rows for these chips and its own C++ row builders
// (crates/core/machine/include/*.hpp) need a cbindgen-generated header that is not in the tree, so they cannot
// be compiled here. What pins this file instead: (1) the reference's in-line sanity identities, restated as
// checks in `check_row` below; (2) tests/test_chip_airs.py evaluates the chips' AIR constraints, transcribed from
// the reference's `eval`, on the generated rows.
//
// Follows, per chip (crates/core/machine/src/alu/...):
//   AddSub      add_sub/mod.rs:44-68 (columns), :161-181 (event_to_row), operations/add.rs:13-57; rows past the
//               events are zero (:99)
//   Bitwise     bitwise/mod.rs:36-64, :160-195; zero padding (:110-115)
//   Lt          lt/mod.rs:36-86, :209-274; zero padding (:127)
//   ShiftLeft   sll/mod.rs:70-104, :232-287; padding rows are the template of :157-165
//   ShiftRight  sr/mod.rs:88-137, :232-339; padding rows set shift_by_n_bits[0] = shift_by_n_bytes[0] = 1 (:183-186)
//   CloClz      clo_clz/mod.rs:41-63, :105-133; padding rows a = 32, is_bb_zero = 1 (:147-163)
//   Jump        crates/core/machine/src/control_flow/jump/columns.rs:11-39, trace.rs:92-113 (JumpEvent, not AluEvent),
//               operations/koala_bear_word.rs:9-42 (the word range checker); zero padding
// Row count: utils/mod.rs next_power_of_two — 2^fixed_log2_rows when the shape fixes it, else the next power of
// two, at least 16. Event layout: #[repr(C)] AluEvent, crates/core/executor/src/events/instr.rs:10-26; opcode
// numbers crates/core/executor/src/opcode.rs:26-48.
#pragma once
#include <algorithm>
#include "bigfield.hpp"
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "field.hpp"

#include "poseidon2.hpp"
#include "septic.hpp"
namespace tracegen {
using namespace orc;

struct AluEvent {  // 28 bytes, as the executor lays it out
  uint32_t pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t hi, a, b, c;
};
static_assert(sizeof(AluEvent) == 28, "AluEvent is seven words");

enum Opcode : uint8_t { ADD = 0, SUB = 1, SLL = 9, SRL = 10, SRA = 11, ROR = 12, SLT = 13, SLTU = 14, AND = 15, OR = 16, XOR = 17, NOR = 18, CLZ = 19, CLO = 20 };
enum Chip { ADD_SUB = 0, BITWISE = 1, LT = 2, SHIFT_LEFT = 3, SHIFT_RIGHT = 4, CLO_CLZ = 5, NUM_CHIPS = 6 };

static inline size_t chip_width(int chip) {
  switch (chip) {
    case ADD_SUB: return 19;
    case BITWISE: return 18;
    case LT: return 32;
    case SHIFT_LEFT: return 44;
    case SHIFT_RIGHT: return 67;
    case CLO_CLZ: return 17;
  }
  throw std::runtime_error("tracegen: unknown chip");
}

static inline size_t padded_rows(size_t n_events, int fixed_log2_rows) {
  if (fixed_log2_rows >= 0) {
    size_t h = (size_t)1 << fixed_log2_rows;
    if (n_events > h) throw std::runtime_error("tracegen: fixed log2 rows is too small");
    return h;
  }
  size_t h = 16;
  while (h < n_events) h <<= 1;
  return h;
}

// canonical field value of a u32 (from_canonical_u32 of a value that may exceed p wraps, as the Monty form does)
static inline F fu32(uint32_t x) { return x % P; }
static inline void word(F* dst, uint32_t v) { for (int i = 0; i < 4; i++) dst[i] = (v >> (8 * i)) & 0xff; }

static inline void add_sub_row(const AluEvent& e, F* r) {
  r[0] = fu32(e.pc);
  r[1] = fu32(e.next_pc);
  const bool is_add = e.opcode == ADD;
  const uint32_t op1 = is_add ? e.b : e.a, op2 = e.c;
  const uint32_t sum = op1 + op2;
  word(r + 2, sum);  // add_operation.value
  uint32_t carry = 0;
  for (int i = 0; i < 3; i++) {
    carry = (((op1 >> (8 * i)) & 0xff) + ((op2 >> (8 * i)) & 0xff) + carry) > 255;
    r[6 + i] = carry;  // add_operation.carry
  }
  word(r + 9, op1);
  word(r + 13, op2);
  r[17] = is_add;
  r[18] = e.opcode == SUB;
}

static inline void bitwise_row(const AluEvent& e, F* r) {
  r[0] = fu32(e.pc);
  r[1] = fu32(e.next_pc);
  word(r + 2, e.a);
  word(r + 6, e.b);
  word(r + 10, e.c);
  r[14] = e.opcode == NOR;
  r[15] = e.opcode == XOR;
  r[16] = e.opcode == OR;
  r[17] = e.opcode == AND;
}

static inline void lt_row(const AluEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, IS_SLT = 2, IS_SLTU = 3, A = 4, B = 8, C = 12, BYTE_FLAGS = 16, B_MASKED = 20, C_MASKED = 21,
         NOT_EQ_INV = 22, MSB_B = 23, MSB_C = 24, BIT_B = 25, BIT_C = 26, SLTU_ = 27, IS_COMP_EQ = 28, IS_SIGN_EQ = 29, CMP_BYTES = 30 };
  r[PC] = fu32(e.pc);
  r[NEXT_PC] = fu32(e.next_pc);
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + C, e.c);
  uint8_t b[4], c[4];
  for (int i = 0; i < 4; i++) { b[i] = (e.b >> (8 * i)) & 0xff; c[i] = (e.c >> (8 * i)) & 0xff; }
  const uint8_t masked_b = b[3] & 0x7f, masked_c = c[3] & 0x7f;
  r[B_MASKED] = masked_b;
  r[C_MASKED] = masked_c;
  uint8_t bc[4], cc[4];
  memcpy(bc, b, 4); memcpy(cc, c, 4);
  if (e.opcode == SLT) { bc[3] = masked_b; cc[3] = masked_c; }
  r[SLTU_] = 0;
  r[IS_COMP_EQ] = memcmp(bc, cc, 4) == 0;
  for (int i = 3; i >= 0; i--) {  // most significant differing byte decides
    if (bc[i] != cc[i]) {
      r[BYTE_FLAGS + i] = 1;
      r[SLTU_] = bc[i] < cc[i];
      r[NOT_EQ_INV] = finv(fsub(bc[i], cc[i]));
      r[CMP_BYTES] = bc[i];
      r[CMP_BYTES + 1] = cc[i];
      break;
    }
  }
  r[MSB_B] = b[3] >> 7;
  r[MSB_C] = c[3] >> 7;
  r[IS_SIGN_EQ] = e.opcode == SLT ? (b[3] >> 7) == (c[3] >> 7) : 1;
  r[IS_SLT] = e.opcode == SLT;
  r[IS_SLTU] = e.opcode == SLTU;
  r[BIT_B] = r[MSB_B] * r[IS_SLT];
  r[BIT_C] = r[MSB_C] * r[IS_SLT];
}

static inline void shift_left_row(const AluEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, A = 2, B = 6, C = 10, C_LSB = 14, BY_BITS = 22, MULT = 30, RESULT = 31, CARRY = 35, BY_BYTES = 39, IS_REAL = 43 };
  r[PC] = fu32(e.pc);
  r[NEXT_PC] = fu32(e.next_pc);
  word(r + A, e.a);
  word(r + B, e.b);
  word(r + C, e.c);
  r[IS_REAL] = 1;
  for (int i = 0; i < 8; i++) r[C_LSB + i] = (e.c >> i) & 1;
  const uint32_t nbits = e.c % 8;
  for (uint32_t i = 0; i < 8; i++) r[BY_BITS + i] = nbits == i;
  const uint32_t mult = 1u << nbits;
  r[MULT] = mult;
  uint32_t carry = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t v = ((e.b >> (8 * i)) & 0xff) * mult + carry;
    carry = v >> 8;
    r[RESULT + i] = v & 0xff;
    r[CARRY + i] = carry;
  }
  const uint32_t nbytes = (e.c & 31) / 8;
  for (uint32_t i = 0; i < 4; i++) r[BY_BYTES + i] = nbytes == i;
}
static inline void shift_left_padding(F* r) {
  r[22] = 1;  // shift_by_n_bits[0]
  r[39] = 1;  // shift_by_n_bytes[0]
  r[30] = 1;  // bit_shift_multiplier
}

static inline void shift_right_row(const AluEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, B = 2, C = 6, BY_BITS = 10, BY_BYTES = 18, BYTE_RES = 22, BIT_RES = 30, SHR_CARRY = 38, SHR_SHIFTED = 46,
         B_MSB = 54, C_LSB = 55, IS_SRL = 63, IS_ROR = 64, IS_SRA = 65, IS_REAL = 66 };
  r[PC] = fu32(e.pc);
  r[NEXT_PC] = fu32(e.next_pc);
  word(r + B, e.b);
  word(r + C, e.c);
  r[B_MSB] = (e.b >> 31) & 1;
  r[IS_SRL] = e.opcode == SRL;
  r[IS_SRA] = e.opcode == SRA;
  r[IS_ROR] = e.opcode == ROR;
  r[IS_REAL] = 1;
  for (int i = 0; i < 8; i++) r[C_LSB + i] = (e.c >> i) & 1;
  const uint32_t nbytes = (e.c % 32) / 8, nbits = (e.c % 32) % 8;
  for (uint32_t i = 0; i < 4; i++) r[BY_BYTES + i] = nbytes == i;
  uint64_t ext;
  if (e.opcode == SRA) ext = (uint64_t)(int64_t)(int32_t)e.b;
  else if (e.opcode == ROR) ext = ((uint64_t)e.b << 32) | e.b;
  else ext = e.b;
  uint8_t byte_res[8] = {0};
  for (uint32_t i = 0; i < 8; i++)
    if (i + nbytes < 8) byte_res[i] = (ext >> (8 * (i + nbytes))) & 0xff;
  for (int i = 0; i < 8; i++) r[BYTE_RES + i] = byte_res[i];
  for (uint32_t i = 0; i < 8; i++) r[BY_BITS + i] = nbits == i;
  const uint32_t carry_mult = 1u << (8 - nbits);
  uint32_t last_carry = 0;
  for (int i = 7; i >= 0; i--) {
    uint8_t shifted = byte_res[i], carry = 0;  // shr_carry, crates/core/machine/src/bytes/utils.rs:2-11
    if (nbits != 0) {
      shifted = byte_res[i] >> nbits;
      carry = byte_res[i] & ((1u << nbits) - 1);
    }
    r[SHR_CARRY + i] = carry;
    r[SHR_SHIFTED + i] = shifted;
    r[BIT_RES + i] = (shifted + last_carry * carry_mult) & 0xff;
    last_carry = carry;
  }
}
static inline void shift_right_padding(F* r) {
  r[10] = 1;  // shift_by_n_bits[0]
  r[18] = 1;  // shift_by_n_bytes[0]
}

static inline void clo_clz_row(const AluEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, A = 2, B = 6, BB = 10, IS_BB_ZERO = 14, IS_CLZ = 15, IS_REAL = 16 };
  word(r + A, e.a);
  word(r + B, e.b);
  r[PC] = fu32(e.pc);
  r[NEXT_PC] = fu32(e.next_pc);
  r[IS_REAL] = 1;
  r[IS_CLZ] = e.opcode == CLZ;
  const uint32_t bb = e.opcode == CLZ ? e.b : 0xffffffffu - e.b;
  word(r + BB, bb);
  r[IS_BB_ZERO] = bb == 0;
}
static inline void clo_clz_padding(F* r) {
  r[2] = 32;  // a = Word::from(32)
  r[14] = 1;  // is_bb_zero
}

// ---- Jump chip: JumpEvent (crates/core/executor/src/events/instr.rs:200-217) ---------------------------------------
struct JumpEvent {
  uint32_t pc, next_pc, next_next_pc;
  uint8_t opcode, _pad[3];
  uint32_t a, b, c;
};
static_assert(sizeof(JumpEvent) == 28, "JumpEvent is seven words");
enum { OP_JUMP = 27, OP_JUMPI = 28, OP_JUMPDIRECT = 29 };
static const size_t JUMP_WIDTH = 66;

// KoalaBearWordRangeChecker::populate (operations/koala_bear_word.rs:27-42): 14 columns
static inline void range_checker(F* r, uint32_t value) {
  for (int i = 0; i < 8; i++) r[i] = (value >> (24 + i)) & 1;
  r[8] = r[0] * r[1];
  for (int i = 0; i < 5; i++) r[9 + i] = r[8 + i] * r[2 + i];
}
static inline void jump_row(const JumpEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, NEXT_PC_RC = 5, NEXT_NEXT_PC = 19, NEXT_NEXT_PC_RC = 23, OP_A = 37, OP_B = 41, OP_C = 45, IS_JUMP = 49,
         IS_JUMPI = 50, IS_JUMPDIRECT = 51, OP_A_RC = 52 };
  r[PC] = fu32(e.pc);
  r[IS_JUMP] = e.opcode == OP_JUMP;
  r[IS_JUMPI] = e.opcode == OP_JUMPI;
  r[IS_JUMPDIRECT] = e.opcode == OP_JUMPDIRECT;
  word(r + OP_A, e.a);
  word(r + OP_B, e.b);
  word(r + OP_C, e.c);
  range_checker(r + OP_A_RC, e.a);
  word(r + NEXT_PC, e.next_pc);
  range_checker(r + NEXT_PC_RC, e.next_pc);
  word(r + NEXT_NEXT_PC, e.next_next_pc);
  range_checker(r + NEXT_NEXT_PC_RC, e.next_next_pc);
}
static inline std::vector<F> generate_jump(const JumpEvent* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * JUMP_WIDTH, 0);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_events; i++) jump_row(events[i], t.data() + i * JUMP_WIDTH);
  *height = h;
  return t;
}

// ---- MovCond chip: MovCondEvent (crates/core/executor/src/events/instr.rs:286-302), columns and row builder
// crates/core/machine/src/misc/mov_cond/mod.rs:38-65, :141-160; IsZeroWordOperation operations/is_zero_word.rs:22-38
struct MovCondEvent {
  uint32_t pc, next_pc;
  uint8_t opcode, _pad[3];
  uint32_t a, b, c, prev_a;
};
static_assert(sizeof(MovCondEvent) == 28, "MovCondEvent is seven words");
enum { OP_MEQ = 50, OP_MNE = 51, OP_WSBH = 52 };
static const size_t MOV_COND_WIDTH = 32;

static inline void mov_cond_row(const MovCondEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, OP_A = 2, PREV_A = 6, OP_B = 10, OP_C = 14, C_EQ_0 = 18, IS_MNE = 29, IS_MEQ = 30, IS_WSBH = 31 };
  r[PC] = fu32(e.pc);
  r[NEXT_PC] = fu32(e.next_pc);
  word(r + OP_A, e.a);
  word(r + OP_B, e.b);
  word(r + OP_C, e.c);
  word(r + PREV_A, e.prev_a);
  // c_eq_0: per byte (inverse, result), then is_lower_half_zero, is_upper_half_zero, result
  bool all_zero = true;
  F res[4];
  for (int i = 0; i < 4; i++) {
    const F byte = (e.c >> (8 * i)) & 0xff;
    r[C_EQ_0 + 2 * i] = byte ? finv(byte) : 0;
    res[i] = r[C_EQ_0 + 2 * i + 1] = byte == 0;
    all_zero &= byte == 0;
  }
  r[C_EQ_0 + 8] = res[0] * res[1];
  r[C_EQ_0 + 9] = res[2] * res[3];
  r[C_EQ_0 + 10] = all_zero;
  r[IS_MEQ] = e.opcode == OP_MEQ;
  r[IS_MNE] = e.opcode == OP_MNE;
  r[IS_WSBH] = e.opcode == OP_WSBH;
}
static inline std::vector<F> generate_mov_cond(const MovCondEvent* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * MOV_COND_WIDTH, 0);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_events; i++) mov_cond_row(events[i], t.data() + i * MOV_COND_WIDTH);
  *height = h;
  return t;
}

// The reference's own sanity identities (debug_assert / assert in the row builders); returns false when one fails.
static inline bool check_row(int chip, const AluEvent& e, const F* r) {
  switch (chip) {
    case ADD_SUB: {  // operations/add.rs:44-46: the top limb overflows by 0 or 256
      const uint32_t op1 = e.opcode == ADD ? e.b : e.a, sum = op1 + e.c;
      const uint32_t ov = (op1 >> 24) + (e.c >> 24) + r[8] - (sum >> 24);
      return ov == 0 || ov == 256;
    }
    case LT:  // lt/mod.rs:266: a[0] = bit_b (1 - bit_c) + is_sign_eq * sltu
      return r[4] == fadd(fmul(r[25], fsub(1, r[26])), fmul(r[29], r[27]));
    case SHIFT_LEFT: {  // sll/mod.rs:281-286
      const uint32_t nbytes = (e.c & 31) / 8;
      for (uint32_t i = nbytes; i < 4; i++)
        if (r[31 + i - nbytes] != ((e.a >> (8 * i)) & 0xff)) return false;
      return true;
    }
    case SHIFT_RIGHT:  // sr/mod.rs:327-332
      for (int i = 0; i < 4; i++)
        if (r[30 + i] != ((e.a >> (8 * i)) & 0xff)) return false;
      return true;
    default: return true;
  }
}

// Row-major canonical trace of `chip` over `events`; `height` rows of chip_width(chip) columns.
static inline std::vector<F> generate(int chip, const AluEvent* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t w = chip_width(chip), h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * w, 0);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * w;
    if (i < n_events) {
      switch (chip) {
        case ADD_SUB: add_sub_row(events[i], r); break;
        case BITWISE: bitwise_row(events[i], r); break;
        case LT: lt_row(events[i], r); break;
        case SHIFT_LEFT: shift_left_row(events[i], r); break;
        case SHIFT_RIGHT: shift_right_row(events[i], r); break;
        case CLO_CLZ: clo_clz_row(events[i], r); break;
      }
    } else if (chip == SHIFT_LEFT) {
      shift_left_padding(r);
    } else if (chip == SHIFT_RIGHT) {
      shift_right_padding(r);
    } else if (chip == CLO_CLZ) {
      clo_clz_padding(r);
    }
  }
  *height = h;
  return t;
}

// ---- byte lookups and the Byte chip -----------------------------------------------------------------------------------
// ByteOpcode (crates/core/executor/src/opcode.rs:195-216); table and multiplicity layout: bytes/columns.rs:12-57,
// bytes/mod.rs:31-104 (ByteChip::trace), bytes/trace.rs:46-66 (generate_trace).
enum ByteOp { B_AND = 0, B_OR = 1, B_XOR = 2, B_SLL = 3, B_U8RANGE = 4, B_SHRCARRY = 5, B_LTU = 6, B_MSB = 7, B_U16RANGE = 8, B_NOR = 9 };
static const size_t NUM_BYTE_OPS = 10, BYTE_ROWS = 1 << 16, BYTE_PREP_COLS = 12;

struct ByteLookup { uint8_t op, b, c; };  // a1 / a2 are functions of (op, b, c); only the table row matters for counting

static inline void range_checks(std::vector<ByteLookup>& out, const F* bytes, int n) {  // events/byte.rs:72-82
  int i = 0;
  for (; i + 1 < n; i += 2) out.push_back({B_U8RANGE, (uint8_t)bytes[i], (uint8_t)bytes[i + 1]});
  if (i < n) out.push_back({B_U8RANGE, (uint8_t)bytes[i], 0});
}

// The `blu` events of each chip's event_to_row, read off its (canonical) row.
static inline void row_lookups(int chip, const AluEvent& e, const F* r, std::vector<ByteLookup>& out) {
  switch (chip) {
    case ADD_SUB:  // operations/add.rs:48-53
      range_checks(out, r + 9, 4); range_checks(out, r + 13, 4); range_checks(out, r + 2, 4);
      break;
    case BITWISE: {  // bitwise/mod.rs:183-193, ByteOpcode::from(opcode) events/byte.rs:148-160
      const uint8_t op = e.opcode == AND ? B_AND : e.opcode == OR ? B_OR : e.opcode == XOR ? B_XOR : B_NOR;
      for (int i = 0; i < 4; i++) out.push_back({op, (uint8_t)r[6 + i], (uint8_t)r[10 + i]});
      break;
    }
    case LT:  // lt/mod.rs:227-241, 268-274
      out.push_back({B_AND, (uint8_t)r[11], 0x7f});
      out.push_back({B_AND, (uint8_t)r[15], 0x7f});
      out.push_back({B_LTU, (uint8_t)r[30], (uint8_t)r[31]});
      break;
    case SHIFT_LEFT:  // sll/mod.rs:276-279
      range_checks(out, r + 31, 4); range_checks(out, r + 35, 4);
      break;
    case SHIFT_RIGHT: {  // sr/mod.rs:258-265, 309-316, 334-337
      out.push_back({B_MSB, (uint8_t)r[5], 0});
      const uint8_t nbits = (uint8_t)((e.c % 32) % 8);
      for (int i = 7; i >= 0; i--) out.push_back({B_SHRCARRY, (uint8_t)r[22 + i], nbits});
      range_checks(out, r + 22, 8); range_checks(out, r + 30, 8); range_checks(out, r + 38, 8); range_checks(out, r + 46, 8);
      break;
    }
    case CLO_CLZ:  // clo_clz/mod.rs:123-130
      range_checks(out, r + 10, 4);
      out.push_back({B_LTU, (uint8_t)e.a, 33});
      break;
  }
}

// ByteChip::generate_trace over the lookups of the given event streams (+ optional plain counts, row-major 65536 x 10)
static inline std::vector<F> byte_mults(size_t n_streams, const int* chips, const AluEvent* const* events, const size_t* n_events,
                                        const uint32_t* extra) {
  std::vector<uint64_t> cnt(BYTE_ROWS * NUM_BYTE_OPS, 0);
  std::vector<ByteLookup> lk;
  for (size_t s = 0; s < n_streams; s++) {
    const size_t w = chip_width(chips[s]);
    std::vector<F> row(w);
    for (size_t i = 0; i < n_events[s]; i++) {
      std::fill(row.begin(), row.end(), 0);
      switch (chips[s]) {
        case ADD_SUB: add_sub_row(events[s][i], row.data()); break;
        case BITWISE: bitwise_row(events[s][i], row.data()); break;
        case LT: lt_row(events[s][i], row.data()); break;
        case SHIFT_LEFT: shift_left_row(events[s][i], row.data()); break;
        case SHIFT_RIGHT: shift_right_row(events[s][i], row.data()); break;
        case CLO_CLZ: clo_clz_row(events[s][i], row.data()); break;
      }
      lk.clear();
      row_lookups(chips[s], events[s][i], row.data(), lk);
      for (const ByteLookup& l : lk) cnt[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
    }
  }
  std::vector<F> t(BYTE_ROWS * NUM_BYTE_OPS);
  for (size_t i = 0; i < t.size(); i++) t[i] = (F)((cnt[i] + (extra ? extra[i] : 0)) % P);
  return t;
}

// ByteChip::trace(): 65536 x 12 row-major canonical
static inline std::vector<F> byte_table() {
  std::vector<F> t(BYTE_ROWS * BYTE_PREP_COLS);
  for (uint32_t b = 0; b < 256; b++)
    for (uint32_t c = 0; c < 256; c++) {
      F* r = t.data() + ((size_t)(b << 8) + c) * BYTE_PREP_COLS;
      const uint32_t k = c & 7;
      r[0] = b; r[1] = c; r[2] = b & c; r[3] = b | c; r[4] = b ^ c; r[5] = (uint8_t)~(b | c);
      r[6] = (uint8_t)(b << k);
      r[7] = k ? b >> k : b;                       // shr_carry(b, c)
      r[8] = k ? (uint8_t)(b << (8 - k)) >> (8 - k) : 0;
      r[9] = b < c; r[10] = (b & 0x80) != 0; r[11] = (b << 8) + c;
    }
  return t;
}

// ---- Branch chip: BranchEvent has JumpEvent's layout (events/instr.rs:161-178); columns control_flow/branch/columns.rs:11-66,
// row builder trace.rs:94-141
static const size_t BRANCH_WIDTH = 62;
enum { OP_BEQ = 21, OP_BGEZ = 22, OP_BGTZ = 23, OP_BLEZ = 24, OP_BLTZ = 25, OP_BNE = 26 };
static inline bool branch_row(const JumpEvent& e, F* r) {  // returns `branching`
  enum { PC = 0, NEXT_PC = 1, NEXT_PC_RC = 5, TARGET_PC = 19, NEXT_NEXT_PC = 23, NEXT_NEXT_PC_RC = 27, OP_A = 41, OP_B = 45, OP_C = 49,
         IS_BEQ = 53, IS_BNE = 54, IS_BLTZ = 55, IS_BLEZ = 56, IS_BGTZ = 57, IS_BGEZ = 58, IS_BRANCHING = 59, A_GT_B = 60, A_LT_B = 61 };
  r[PC] = fu32(e.pc);
  r[IS_BEQ] = e.opcode == OP_BEQ; r[IS_BNE] = e.opcode == OP_BNE; r[IS_BLTZ] = e.opcode == OP_BLTZ;
  r[IS_BGTZ] = e.opcode == OP_BGTZ; r[IS_BLEZ] = e.opcode == OP_BLEZ; r[IS_BGEZ] = e.opcode == OP_BGEZ;
  word(r + OP_A, e.a); word(r + OP_B, e.b); word(r + OP_C, e.c);
  const bool eq = e.a == e.b, lt = (int32_t)e.a < (int32_t)e.b, gt = (int32_t)e.a > (int32_t)e.b;
  r[A_LT_B] = lt; r[A_GT_B] = gt;
  bool branching = false;
  switch (e.opcode) {
    case OP_BEQ: branching = eq; break;
    case OP_BNE: branching = !eq; break;
    case OP_BLTZ: branching = lt; break;
    case OP_BLEZ: branching = lt || eq; break;
    case OP_BGTZ: branching = gt; break;
    case OP_BGEZ: branching = eq || gt; break;
    default: throw std::runtime_error("tracegen: invalid branch opcode");
  }
  word(r + NEXT_PC, e.next_pc);
  word(r + TARGET_PC, e.next_pc + e.c);
  word(r + NEXT_NEXT_PC, e.next_next_pc);
  range_checker(r + NEXT_PC_RC, e.next_pc);
  range_checker(r + NEXT_NEXT_PC_RC, e.next_next_pc);
  r[IS_BRANCHING] = branching;
  return branching;
}
static inline std::vector<F> generate_branch(const JumpEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                             uint64_t* byte_counts /* nullable: [row][op] */) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * BRANCH_WIDTH, 0);
  for (size_t i = 0; i < n_events; i++) {
    F* r = t.data() + i * BRANCH_WIDTH;
    const bool branching = branch_row(events[i], r);
    if (!branching && byte_counts) {  // trace.rs:137-140: range checks of next_pc and next_next_pc
      std::vector<ByteLookup> lk;
      range_checks(lk, r + 1, 4);
      range_checks(lk, r + 23, 4);
      for (const ByteLookup& l : lk) byte_counts[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
    }
  }
  *height = h;
  return t;
}


// ---- Mul chip: CompAluEvents (crates/core/executor/src/events/instr.rs:50-73); columns alu/mul/mod.rs:79-141, row builder
// :221-337, the HI-register access columns memory/consistency/columns.rs:18-45 filled by consistency/trace.rs:43-100
struct MemoryWriteRecord { uint32_t value, shard, timestamp, prev_value, prev_shard, prev_timestamp; };
struct CompAluEvent {
  uint32_t shard, clk, pc, next_pc;
  uint8_t opcode; uint8_t pad[3];
  uint32_t hi, a, b, c;
  MemoryWriteRecord hi_record;
  uint8_t hi_record_is_real; uint8_t pad2[3];
};
static_assert(sizeof(CompAluEvent) == 64, "CompAluEvent is sixteen words");
static const size_t MUL_WIDTH = 58;
enum { OP_MUL = 2, OP_MULT = 3, OP_MULTU = 4 };
enum { B_U8RANGE_OP = 4, B_MSB_OP = 7, B_U16RANGE_OP = 8 };   // ByteOpcode, opcode.rs:195-216

// MemoryAccessCols::populate_access (consistency/trace.rs:69-100) for a write; `r` points at prev_value (13 columns)
static inline void memory_write_cols(const MemoryWriteRecord& rec, F* r, std::vector<ByteLookup>* lk) {
  enum { PREV_VALUE = 0, VALUE = 4, PREV_SHARD = 8, PREV_CLK = 9, COMPARE_CLK = 10, DIFF_16 = 11, DIFF_8 = 12 };
  word(r + PREV_VALUE, rec.prev_value);
  word(r + VALUE, rec.value);
  r[PREV_SHARD] = fu32(rec.prev_shard);
  r[PREV_CLK] = fu32(rec.prev_timestamp);
  const bool use_clk = rec.prev_shard == rec.shard;
  r[COMPARE_CLK] = use_clk;
  const uint32_t prev_t = use_clk ? rec.prev_timestamp : rec.prev_shard, cur_t = use_clk ? rec.timestamp : rec.shard;
  const uint32_t diff_minus_one = cur_t - prev_t - 1u;   // wrapping_sub
  const uint32_t d16 = diff_minus_one & 0xffff, d8 = (diff_minus_one >> 16) & 0xff;
  r[DIFF_16] = d16;
  r[DIFF_8] = d8;
  if (lk) {
    lk->push_back(ByteLookup{B_U16RANGE_OP, (uint8_t)(d16 >> 8), (uint8_t)d16});   // the table row of U16Range is its value
    lk->push_back(ByteLookup{B_U8RANGE_OP, 0, (uint8_t)d8});
  }
}

static inline void mul_row(const CompAluEvent& e, F* r, std::vector<ByteLookup>* lk) {
  enum { PC = 0, NEXT_PC = 1, HI = 2, A = 6, B = 10, C = 14, CARRY = 18, PRODUCT = 26, B_MSB = 34, C_MSB = 35, B_SIGN_EXTEND = 36,
         C_SIGN_EXTEND = 37, IS_MUL = 38, IS_MULT = 39, IS_MULTU = 40, IS_REAL = 41, OP_HI_ACCESS = 42, HI_RECORD_IS_REAL = 55,
         SHARD = 56, CLK = 57 };
  if (e.opcode != OP_MUL && e.opcode != OP_MULT && e.opcode != OP_MULTU) throw std::runtime_error("tracegen: invalid mul opcode");
  r[PC] = fu32(e.pc);
  r[NEXT_PC] = fu32(e.next_pc);
  r[HI_RECORD_IS_REAL] = e.hi_record_is_real != 0;
  if (e.hi_record_is_real) {
    memory_write_cols(e.hi_record, r + OP_HI_ACCESS, lk);
    r[SHARD] = fu32(e.shard);
    r[CLK] = fu32(e.clk);
  }
  uint8_t b[8], c[8];
  for (int i = 0; i < 4; i++) { b[i] = (e.b >> (8 * i)) & 0xff; c[i] = (e.c >> (8 * i)) & 0xff; }
  const uint8_t b_msb = b[3] >> 7, c_msb = c[3] >> 7;
  r[B_MSB] = b_msb;
  r[C_MSB] = c_msb;
  int nb = 4, nc = 4;
  if (e.opcode == OP_MULT && b_msb) { r[B_SIGN_EXTEND] = 1; for (int i = 4; i < 8; i++) b[i] = 0xff; nb = 8; }
  if (e.opcode == OP_MULT && c_msb) { r[C_SIGN_EXTEND] = 1; for (int i = 4; i < 8; i++) c[i] = 0xff; nc = 8; }
  if (lk) {
    lk->push_back(ByteLookup{B_MSB_OP, b[3], 0});
    lk->push_back(ByteLookup{B_MSB_OP, c[3], 0});
  }
  uint32_t product[8] = {0}, carry[8];
  for (int i = 0; i < nb; i++)
    for (int j = 0; j < nc; j++)
      if (i + j < 8) product[i + j] += (uint32_t)b[i] * c[j];
  for (int i = 0; i < 8; i++) {
    carry[i] = product[i] >> 8;
    product[i] &= 0xff;
    if (i + 1 < 8) product[i + 1] += carry[i];
    r[CARRY + i] = carry[i];
    r[PRODUCT + i] = product[i];
  }
  word(r + HI, e.hi); word(r + A, e.a); word(r + B, e.b); word(r + C, e.c);
  r[IS_REAL] = 1;
  r[IS_MUL] = e.opcode == OP_MUL; r[IS_MULT] = e.opcode == OP_MULT; r[IS_MULTU] = e.opcode == OP_MULTU;
  if (lk) {
    for (int i = 0; i < 8; i++) lk->push_back(ByteLookup{B_U16RANGE_OP, (uint8_t)(carry[i] >> 8), (uint8_t)carry[i]});
    range_checks(*lk, r + PRODUCT, 8);
  }
}
static inline std::vector<F> generate_mul(const CompAluEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                          uint64_t* byte_counts /* nullable: [row][op] */) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * MUL_WIDTH, 0);
  std::vector<ByteLookup> lk;
  for (size_t i = 0; i < n_events; i++) {
    lk.clear();
    mul_row(events[i], t.data() + i * MUL_WIDTH, byte_counts ? &lk : nullptr);
    for (const ByteLookup& l : lk) byte_counts[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
  }
  *height = h;
  return t;
}

// ---- DivRem chip: CompAluEvents; columns alu/divrem/mod.rs:106-202, row builder :224-381 (get_quotient_and_remainder:
// crates/core/executor/src/utils.rs:33-43), IsZeroWordOperation / IsEqualWordOperation: operations/is_zero_word.rs:25-38,
// is_equal_word.rs:15-28 (inverses of field differences)
static const size_t DIVREM_WIDTH = 106;
enum { OP_DIV = 5, OP_DIVU = 6, OP_MOD = 7, OP_MODU = 8 };
// 11 columns: per byte (inverse, result), is_lower_half_zero, is_upper_half_zero, result
static inline void is_zero_word_cols(const F a[4], F* r) {
  bool z[4];
  for (int i = 0; i < 4; i++) {
    z[i] = a[i] == 0;
    r[2 * i] = z[i] ? 0 : finv(a[i]);
    r[2 * i + 1] = z[i];
  }
  r[8] = z[0] && z[1];
  r[9] = z[2] && z[3];
  r[10] = z[0] && z[1] && z[2] && z[3];
}
static inline void is_equal_word_cols(uint32_t a, uint32_t b, F* r) {
  F diff[4];
  for (int i = 0; i < 4; i++) diff[i] = fsub((a >> (8 * i)) & 0xff, (b >> (8 * i)) & 0xff);
  is_zero_word_cols(diff, r);
}
static inline void divrem_row(const CompAluEvent& e, F* r, std::vector<ByteLookup>* lk) {
  enum { PC = 0, NEXT_PC = 1, B = 2, C = 6, QUOTIENT = 10, REMAINDER = 14, ABS_REMAINDER = 18, ABS_C = 22, MAX_ABS_C_OR_1 = 26,
         C_TIMES_QUOTIENT = 30, CARRY = 38, IS_C_0 = 46, IS_DIV = 57, IS_DIVU = 58, IS_MOD = 59, IS_MODU = 60, IS_OVERFLOW = 61,
         IS_OVERFLOW_B = 62, IS_OVERFLOW_C = 73, B_MSB = 84, REM_MSB = 85, C_MSB = 86, B_NEG = 87, REM_NEG = 88, C_NEG = 89,
         REMAINDER_CHECK_MULTIPLICITY = 90, OP_HI_ACCESS = 91, SHARD = 104, CLK = 105 };
  if (e.opcode < OP_DIV || e.opcode > OP_MODU) throw std::runtime_error("tracegen: invalid divrem opcode");
  const bool is_signed = e.opcode == OP_DIV || e.opcode == OP_MOD;
  word(r + B, e.b); word(r + C, e.c);
  r[PC] = fu32(e.pc); r[NEXT_PC] = fu32(e.next_pc);
  r[IS_DIVU] = e.opcode == OP_DIVU; r[IS_DIV] = e.opcode == OP_DIV; r[IS_MODU] = e.opcode == OP_MODU; r[IS_MOD] = e.opcode == OP_MOD;
  F cw[4]; word(cw, e.c);
  is_zero_word_cols(cw, r + IS_C_0);
  if (e.opcode == OP_DIVU || e.opcode == OP_DIV) {
    memory_write_cols(e.hi_record, r + OP_HI_ACCESS, lk);
    r[SHARD] = fu32(e.shard);
    r[CLK] = fu32(e.clk);
  }
  uint32_t quotient, remainder;
  if (e.c == 0) { quotient = 0xffffffffu; remainder = e.b; }
  else if (is_signed) {
    const int64_t sb = (int32_t)e.b, sc = (int32_t)e.c;   // 64-bit: i32::MIN / -1 wraps to i32::MIN, remainder 0
    quotient = (uint32_t)(sb / sc); remainder = (uint32_t)(sb % sc);
  } else { quotient = e.b / e.c; remainder = e.b % e.c; }
  word(r + QUOTIENT, quotient); word(r + REMAINDER, remainder);
  r[REM_MSB] = remainder >> 31; r[B_MSB] = e.b >> 31; r[C_MSB] = e.c >> 31;
  is_equal_word_cols(e.b, 0x80000000u, r + IS_OVERFLOW_B);
  is_equal_word_cols(e.c, 0xffffffffu, r + IS_OVERFLOW_C);
  auto unsigned_abs = [](uint32_t v) { return (v >> 31) ? (uint32_t)(0u - v) : v; };
  if (is_signed) {
    const uint32_t abs_c = unsigned_abs(e.c);
    r[REM_NEG] = r[REM_MSB]; r[B_NEG] = r[B_MSB]; r[C_NEG] = r[C_MSB];
    r[IS_OVERFLOW] = e.b == 0x80000000u && e.c == 0xffffffffu;
    word(r + ABS_REMAINDER, unsigned_abs(remainder));
    word(r + ABS_C, abs_c);
    word(r + MAX_ABS_C_OR_1, abs_c > 1 ? abs_c : 1);
  } else {
    word(r + ABS_REMAINDER, remainder);
    word(r + ABS_C, e.c);
    word(r + MAX_ABS_C_OR_1, e.c > 1 ? e.c : 1);
  }
  if (lk) {
    lk->push_back(ByteLookup{B_MSB_OP, (uint8_t)(e.b >> 24), 0});
    lk->push_back(ByteLookup{B_MSB_OP, (uint8_t)(e.c >> 24), 0});
    lk->push_back(ByteLookup{B_MSB_OP, (uint8_t)(remainder >> 24), 0});
  }
  r[REMAINDER_CHECK_MULTIPLICITY] = e.c != 0;   // 1 - is_c_0.result
  const uint64_t ctq = is_signed ? (uint64_t)((int64_t)(int32_t)quotient * (int64_t)(int32_t)e.c) : (uint64_t)quotient * e.c;
  const uint64_t rem64 = is_signed ? (uint64_t)(int64_t)(int32_t)remainder : (uint64_t)remainder;
  uint32_t carry = 0;
  for (int i = 0; i < 8; i++) {
    r[C_TIMES_QUOTIENT + i] = (ctq >> (8 * i)) & 0xff;
    const uint32_t x = (uint32_t)((ctq >> (8 * i)) & 0xff) + (uint32_t)((rem64 >> (8 * i)) & 0xff) + carry;
    carry = x >> 8;
    r[CARRY + i] = carry;
  }
  if (lk) {
    range_checks(*lk, r + QUOTIENT, 4);
    range_checks(*lk, r + REMAINDER, 4);
    range_checks(*lk, r + C_TIMES_QUOTIENT, 8);
  }
}
static inline std::vector<F> generate_divrem(const CompAluEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                             uint64_t* byte_counts /* nullable: [row][op] */) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * DIVREM_WIDTH, 0);
  std::vector<ByteLookup> lk;
  for (size_t i = 0; i < n_events; i++) {
    lk.clear();
    divrem_row(events[i], t.data() + i * DIVREM_WIDTH, byte_counts ? &lk : nullptr);
    for (const ByteLookup& l : lk) byte_counts[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
  }
  *height = h;
  return t;
}

// ---- Cpu chip: CpuEventFfi / InstructionFfi (crates/core/executor/src/events/cpu.rs:46-106, instruction.rs:22-33); columns
// cpu/columns/mod.rs:16-77 + instruction.rs:12-32; row builder cpu/trace.rs:117-257, padding rows :57-60; the read access
// columns memory/consistency/trace.rs:22-33
struct MemoryReadRecord { uint32_t value, shard, timestamp, prev_shard, prev_timestamp; };
struct OptionMemoryRecord { uint8_t tag; uint8_t pad[3]; MemoryReadRecord read; MemoryWriteRecord write; };   // tag: Read 0, Write 1, None 2
struct OptionU32 { uint8_t tag; uint8_t pad[3]; uint32_t value; };                                            // tag: Some 0, None 1
struct CpuEvent {
  uint32_t clk, pc, next_pc, next_next_pc, a;
  OptionMemoryRecord a_record;
  uint32_t b;
  OptionMemoryRecord b_record;
  uint32_t c;
  OptionMemoryRecord c_record;
  OptionU32 hi;
  OptionMemoryRecord hi_record, memory_record;
  uint32_t exit_code;
};
static_assert(sizeof(CpuEvent) == 280, "CpuEventFfi is seventy words");
struct Instruction { uint8_t opcode, op_a; uint8_t pad0[2]; uint32_t op_b, op_c; uint8_t imm_b, imm_c; uint8_t pad1[2]; OptionU32 raw; };
static_assert(sizeof(Instruction) == 24, "InstructionFfi is six words");
static const size_t CPU_WIDTH = 67, PROGRAM_PREP_WIDTH = 14;

namespace opc {   // crates/core/executor/src/opcode.rs:26-90 and the predicates of instruction.rs:70-310
static inline bool is_branch(uint8_t o) { return o >= 21 && o <= 26; }
static inline bool is_jump(uint8_t o) { return o >= 27 && o <= 29; }
static inline bool is_syscall(uint8_t o) { return o == 30; }
static inline bool is_memory(uint8_t o) { return o >= 31 && o <= 44; }
static inline bool is_store_except_sc(uint8_t o) { return o >= 39 && o <= 43; }
static inline bool is_mult_div(uint8_t o) { return o == 3 || o == 4 || o == 5 || o == 6; }
static inline bool is_maddsub(uint8_t o) { return o >= 46 && o <= 49; }
static inline bool is_check_memory(uint8_t o) { return is_syscall(o) || is_maddsub(o) || is_memory(o); }
static inline bool is_rw_a(uint8_t o) { return is_syscall(o) || o == 45 || is_maddsub(o) || o == 50 || o == 51 || is_memory(o); }
}  // namespace opc

// MemoryReadCols::populate: value, prev_shard, prev_clk, compare_clk, diff_16bit_limb, diff_8bit_limb (9 columns at `r`)
static inline void memory_access_cols(uint32_t value, uint32_t shard, uint32_t ts, uint32_t prev_shard, uint32_t prev_ts, F* r,
                                      std::vector<ByteLookup>* lk) {
  word(r, value);
  r[4] = fu32(prev_shard);
  r[5] = fu32(prev_ts);
  const bool use_clk = prev_shard == shard;
  r[6] = use_clk;
  const uint32_t diff_minus_one = (use_clk ? ts : shard) - (use_clk ? prev_ts : prev_shard) - 1u;
  const uint32_t d16 = diff_minus_one & 0xffff, d8 = (diff_minus_one >> 16) & 0xff;
  r[7] = d16;
  r[8] = d8;
  if (lk) {
    lk->push_back(ByteLookup{B_U16RANGE_OP, (uint8_t)(d16 >> 8), (uint8_t)d16});
    lk->push_back(ByteLookup{B_U8RANGE_OP, 0, (uint8_t)d8});
  }
}
static inline void instruction_cols(const Instruction& in, F* r) {   // opcode, op_a, op_b[4], op_c[4], op_a_0, imm_b, imm_c
  r[0] = in.opcode; r[1] = in.op_a;
  word(r + 2, in.op_b); word(r + 6, in.op_c);
  r[10] = in.op_a == 0; r[11] = in.imm_b != 0; r[12] = in.imm_c != 0;
}
static inline void cpu_row(const CpuEvent& e, uint32_t shard, const Instruction& in, F* r, std::vector<ByteLookup>* lk) {
  enum { SHARD = 0, CLK_16 = 1, CLK_8 = 2, SHARD_TO_SEND = 3, CLK_TO_SEND = 4, PC = 5, NEXT_PC = 6, NEXT_NEXT_PC = 7, INSTRUCTION = 8,
         NUM_EXTRA_CYCLES = 21, IS_RW_A = 22, IS_CHECK_MEMORY = 23, IS_HALT = 24, IS_SEQUENTIAL = 25, OP_A_VALUE = 26, HI_OR_PREV_A = 30,
         OP_A_ACCESS = 34, OP_B_ACCESS = 47, OP_C_ACCESS = 56, IS_REAL = 65, OP_A_IMMUTABLE = 66 };
  r[SHARD] = fu32(shard);
  const uint32_t clk16 = e.clk & 0xffff, clk8 = (e.clk >> 16) & 0xff;
  r[CLK_16] = clk16; r[CLK_8] = clk8;
  if (lk) {
    lk->push_back(ByteLookup{B_U16RANGE_OP, (uint8_t)((shard & 0xffff) >> 8), (uint8_t)shard});
    lk->push_back(ByteLookup{B_U16RANGE_OP, (uint8_t)(clk16 >> 8), (uint8_t)clk16});
    lk->push_back(ByteLookup{B_U8RANGE_OP, 0, (uint8_t)clk8});
  }
  r[PC] = fu32(e.pc); r[NEXT_PC] = fu32(e.next_pc); r[NEXT_NEXT_PC] = fu32(e.next_next_pc);
  instruction_cols(in, r + INSTRUCTION);
  const uint8_t o = in.opcode;
  r[OP_A_IMMUTABLE] = opc::is_store_except_sc(o) || opc::is_branch(o) || o == 54;   // TEQ
  r[IS_RW_A] = opc::is_rw_a(o);
  const bool check_memory = opc::is_mult_div(o) || opc::is_check_memory(o);
  r[IS_CHECK_MEMORY] = check_memory;
  word(r + OP_A_VALUE, e.a);
  if (e.hi.tag == 0) word(r + HI_OR_PREV_A, e.hi.value);
  word(r + OP_A_ACCESS + 4, e.a);        // op_a_access = prev_value(4), access(9)
  word(r + OP_B_ACCESS, e.b);
  word(r + OP_C_ACCESS, e.c);
  r[SHARD_TO_SEND] = check_memory ? r[SHARD] : 0;
  r[CLK_TO_SEND] = check_memory ? fu32(e.clk) : 0;
  if (e.a_record.tag == 1) {
    const MemoryWriteRecord& w = e.a_record.write;
    word(r + OP_A_ACCESS, w.prev_value);
    memory_access_cols(w.value, w.shard, w.timestamp, w.prev_shard, w.prev_timestamp, r + OP_A_ACCESS + 4, lk);
  } else if (e.a_record.tag == 0) {
    const MemoryReadRecord& rd = e.a_record.read;
    word(r + OP_A_ACCESS, rd.value);
    memory_access_cols(rd.value, rd.shard, rd.timestamp, rd.prev_shard, rd.prev_timestamp, r + OP_A_ACCESS + 4, lk);
  }
  if (e.b_record.tag == 0) {
    const MemoryReadRecord& rd = e.b_record.read;
    memory_access_cols(rd.value, rd.shard, rd.timestamp, rd.prev_shard, rd.prev_timestamp, r + OP_B_ACCESS, lk);
  }
  if (e.c_record.tag == 0) {
    const MemoryReadRecord& rd = e.c_record.read;
    memory_access_cols(rd.value, rd.shard, rd.timestamp, rd.prev_shard, rd.prev_timestamp, r + OP_C_ACCESS, lk);
  }
  bool is_halt = false;
  if (opc::is_syscall(o)) {   // SyscallCode::HALT = 0, SYS_EXT_GROUP = 4246 (crates/core/executor/src/syscalls/code.rs)
    const F id0 = r[OP_A_ACCESS], id1 = r[OP_A_ACCESS + 1];
    is_halt = (id0 == 0 && id1 == 0) || (id0 == (4246 & 0xff) && id1 == (4246 >> 8));
    r[IS_HALT] = is_halt;
    r[NUM_EXTRA_CYCLES] = r[OP_A_ACCESS + 3];
  }
  r[IS_SEQUENTIAL] = !is_halt && !opc::is_branch(o) && !opc::is_jump(o);
  if (lk) {
    lk->push_back(ByteLookup{B_U8RANGE_OP, (uint8_t)r[OP_A_ACCESS + 4], (uint8_t)r[OP_A_ACCESS + 5]});
    lk->push_back(ByteLookup{B_U8RANGE_OP, (uint8_t)r[OP_A_ACCESS + 6], (uint8_t)r[OP_A_ACCESS + 7]});
  }
  r[IS_REAL] = 1;
}
static inline std::vector<F> generate_cpu(const CpuEvent* events, size_t n_events, const Instruction* program, size_t n_instr,
                                          uint32_t pc_base, uint32_t shard, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * CPU_WIDTH, 0);
  std::vector<ByteLookup> lk;
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * CPU_WIDTH;
    if (i >= n_events) { r[8 + 11] = 1; r[8 + 12] = 1; r[22] = 1; continue; }   // imm_b, imm_c, is_rw_a
    const size_t idx = (events[i].pc - pc_base) / 4;   // Program::fetch
    if (events[i].pc < pc_base || idx >= n_instr) throw std::runtime_error("tracegen: pc outside the program");
    lk.clear();
    cpu_row(events[i], shard, program[idx], r, byte_counts ? &lk : nullptr);
    for (const ByteLookup& l : lk) byte_counts[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
  }
  *height = h;
  return t;
}
// ProgramChip: preprocessed (pc, instruction) rows (program/mod.rs:62-101) and the multiplicity column (:113-146)
static inline std::vector<F> generate_program_prep(const Instruction* program, size_t n_instr, uint32_t pc_base, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows(n_instr, fixed_log2_rows);
  std::vector<F> t(h * PROGRAM_PREP_WIDTH, 0);
  for (size_t i = 0; i < n_instr; i++) {
    t[i * PROGRAM_PREP_WIDTH] = fu32(pc_base + 4 * (uint32_t)i);
    instruction_cols(program[i], t.data() + i * PROGRAM_PREP_WIDTH + 1);
  }
  *height = h;
  return t;
}
static inline std::vector<F> generate_program_mult(const CpuEvent* events, size_t n_events, size_t n_instr, uint32_t pc_base, int fixed_log2_rows,
                                                   size_t* height) {
  const size_t h = padded_rows(n_instr, fixed_log2_rows);
  std::vector<F> t(h, 0);
  for (size_t i = 0; i < n_events; i++) {
    const size_t idx = (events[i].pc - pc_base) / 4;
    if (events[i].pc >= pc_base && idx < n_instr) t[idx] = fadd(t[idx], 1);
  }
  *height = h;
  return t;
}

// ---- MemoryLocal chip (memory/local.rs): MemoryLocalEvents (events/memory.rs:226-237), four per row; columns per entry
// :30-50 (addr, initial_shard, final_shard, initial_clk, final_clk, initial_value[4], final_value[4], is_real), rows :147-190
struct MemoryRecordC { uint32_t shard, timestamp, value; };
struct MemoryLocalEvent { uint32_t addr; MemoryRecordC initial, final_; };
static_assert(sizeof(MemoryLocalEvent) == 28, "MemoryLocalEvent is seven words");
static const size_t MEMORY_LOCAL_ENTRIES = 4, MEMORY_LOCAL_ENTRY_COLS = 14, MEMORY_LOCAL_WIDTH = 56;
static inline std::vector<F> generate_memory_local(const MemoryLocalEvent* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows((n_events + MEMORY_LOCAL_ENTRIES - 1) / MEMORY_LOCAL_ENTRIES, fixed_log2_rows);
  std::vector<F> t(h * MEMORY_LOCAL_WIDTH, 0);
  for (size_t i = 0; i < n_events; i++) {
    F* r = t.data() + i * MEMORY_LOCAL_ENTRY_COLS;   // entry k of row i / 4 is at (i / 4) * 56 + (i % 4) * 14 = i * 14
    const MemoryLocalEvent& e = events[i];
    r[0] = fu32(e.addr); r[1] = fu32(e.initial.shard); r[2] = fu32(e.final_.shard); r[3] = fu32(e.initial.timestamp); r[4] = fu32(e.final_.timestamp);
    word(r + 5, e.initial.value); word(r + 9, e.final_.value);
    r[13] = 1;
  }
  *height = h;
  return t;
}

// ---- MemoryInstructions chip: MemInstrEvents (crates/core/executor/src/events/instr.rs:108-136; mem_access is the #[repr(C)] enum
// MemoryRecordEnum: tag Read 0 / Write 1, then the record); columns memory/instructions/columns.rs:12-117, row trace.rs:100-262
struct MemInstrEvent {
  uint32_t shard, clk, pc, next_pc;
  uint8_t opcode; uint8_t pad[3];
  uint32_t a, b, c;
  uint32_t mem_tag;
  uint32_t mem[6];   // Read: value, shard, timestamp, prev_shard, prev_timestamp; Write: value, shard, timestamp, prev_value, prev_shard, prev_timestamp
  uint32_t prev_a_val;
};
static_assert(sizeof(MemInstrEvent) == 64, "MemInstrEvent is sixteen words");
static const size_t MEMORY_INSTRS_WIDTH = 79;
enum { B_AND_OP = 0, B_LTU_OP = 6 };
static inline void memory_instr_row(const MemInstrEvent& e, F* r, std::vector<ByteLookup>* lk) {
  enum { PC = 0, NEXT_PC = 1, SHARD = 2, CLK = 3, OP_A = 4, OP_B = 8, OP_C = 12, IS_LB = 16, ADDR_WORD = 30, ADDR_ALIGNED = 34, ADDR_LS_TWO_BITS = 35,
         LS_IS_ONE = 36, LS_IS_TWO = 37, LS_IS_THREE = 38, ADDR_RC = 39, MEMORY_ACCESS = 53, PREV_A_VAL = 66, UNSIGNED_MEM_VAL = 70,
         MOST_SIG_BIT = 74, MOST_SIG_BYTE = 75, MEM_VALUE_IS_NEG = 76, MOST_SIG_BYTES_ZERO = 77 };
  const uint8_t o = e.opcode;
  if (o < 31 || o > 44) throw std::runtime_error("tracegen: invalid memory opcode");
  if (e.shard == 0) throw std::runtime_error("tracegen: memory instruction in shard 0");
  r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[PC] = fu32(e.pc); r[NEXT_PC] = fu32(e.next_pc);
  word(r + OP_A, e.a); word(r + OP_B, e.b); word(r + OP_C, e.c);
  uint32_t mem_value;
  if (e.mem_tag == 1) {
    MemoryWriteRecord w{e.mem[0], e.mem[1], e.mem[2], e.mem[3], e.mem[4], e.mem[5]};
    memory_write_cols(w, r + MEMORY_ACCESS, lk);
    mem_value = w.value;
  } else {
    word(r + MEMORY_ACCESS, e.mem[0]);   // populate_read: prev_value = value
    memory_access_cols(e.mem[0], e.mem[1], e.mem[2], e.mem[3], e.mem[4], r + MEMORY_ACCESS + 4, lk);
    mem_value = e.mem[0];
  }
  word(r + PREV_A_VAL, e.prev_a_val);
  const uint32_t addr = e.b + e.c, aligned = addr & ~3u, ls = addr & 3;
  word(r + ADDR_WORD, addr);
  range_checker(r + ADDR_RC, addr);
  r[ADDR_ALIGNED] = fu32(aligned);
  r[ADDR_LS_TWO_BITS] = ls; r[LS_IS_ONE] = ls == 1; r[LS_IS_TWO] = ls == 2; r[LS_IS_THREE] = ls == 3;
  if (lk) lk->push_back(ByteLookup{B_AND_OP, (uint8_t)addr, 3});
  if (o <= 38) {   // loads
    uint32_t u;
    switch (o) {
      case 31: case 32: u = (mem_value >> (8 * ls)) & 0xff; break;                       // LB, LBU
      case 33: case 34: u = (ls >> 1) ? mem_value >> 16 : mem_value & 0xffff; break;     // LH, LHU
      case 36: { const uint32_t sh = 24 - 8 * ls; u = (e.prev_a_val & ~(0xffffffffu << sh)) | (mem_value << sh); break; }   // LWL
      case 37: { const uint32_t sh = 8 * ls; u = (e.prev_a_val & ~(0xffffffffu >> sh)) | (mem_value >> sh); break; }         // LWR
      default: u = mem_value;                                                            // LW, LL
    }
    word(r + UNSIGNED_MEM_VAL, u);
    if (o == 31 || o == 33) {
      const uint8_t byte = o == 31 ? (uint8_t)u : (uint8_t)(u >> 8);
      r[MEM_VALUE_IS_NEG] = byte >> 7;
      r[MOST_SIG_BYTE] = byte;
      r[MOST_SIG_BIT] = byte >> 7;
      if (lk) lk->push_back(ByteLookup{B_MSB_OP, byte, 0});
    }
  }
  r[IS_LB + (o - 31)] = 1;   // is_lb, is_lbu, is_lh, is_lhu, is_lw, is_lwl, is_lwr, is_ll, is_sb, is_sh, is_sw, is_swl, is_swr, is_sc: opcode order
  if (lk) lk->push_back(ByteLookup{B_U8RANGE_OP, (uint8_t)(addr >> 8), (uint8_t)(addr >> 16)});
  const F upper = fadd(fadd((addr >> 8) & 0xff, (addr >> 16) & 0xff), addr >> 24);
  r[MOST_SIG_BYTES_ZERO] = upper ? finv(upper) : 0;
  r[MOST_SIG_BYTES_ZERO + 1] = upper == 0;
  if (upper == 0 && lk) lk->push_back(ByteLookup{B_LTU_OP, 35, (uint8_t)addr});   // NUM_REGISTERS - 1 < addr
}
static inline std::vector<F> generate_memory_instrs(const MemInstrEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                                    uint64_t* byte_counts) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * MEMORY_INSTRS_WIDTH, 0);
  std::vector<ByteLookup> lk;
  for (size_t i = 0; i < n_events; i++) {
    lk.clear();
    memory_instr_row(events[i], t.data() + i * MEMORY_INSTRS_WIDTH, byte_counts ? &lk : nullptr);
    for (const ByteLookup& l : lk) byte_counts[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
  }
  *height = h;
  return t;
}

// ---- recursion Poseidon2Wide chip, degree 3 (crates/recursion/core/src/chips/poseidon2_wide/): columns
// columns/permutation.rs:20-36 (PermutationSBox: external_rounds_state[8][16], internal_rounds_state[16], internal_rounds_s0[12],
// output_state[16], external_rounds_sbox_state[8][16], internal_rounds_sbox_state[13]); rows trace.rs:277-420; padding rows are
// the permutation of the zero state (:99-105)
static const size_t POSEIDON2_WIDE_WIDTH = 313;
static inline void poseidon2_wide_row(const F input[16], F* r) {
  enum { EXT_STATE = 0, INT_STATE = 128, INT_S0 = 144, OUTPUT = 156, EXT_SBOX = 172, INT_SBOX = 300 };
  F state[16];
  for (int i = 0; i < 16; i++) r[EXT_STATE + i] = state[i] = input[i];
  auto external_round = [&](int rd) {
    if (rd == 0) orc::external_layer(state);
    const int round = rd < 4 ? rd : rd + 13;
    for (int i = 0; i < 16; i++) {
      state[i] = orc::sbox(fadd(state[i], orc::ORC_RC_16_30[round][i]));
      r[EXT_SBOX + 16 * rd + i] = state[i];
    }
    orc::external_layer(state);
    F* next = rd == 3 ? r + INT_STATE : rd == 7 ? r + OUTPUT : r + EXT_STATE + 16 * (rd + 1);
    for (int i = 0; i < 16; i++) next[i] = state[i];
  };
  for (int rd = 0; rd < 4; rd++) external_round(rd);
  for (int rd = 0; rd < 13; rd++) {
    state[0] = orc::sbox(fadd(state[0], orc::ORC_RC_16_30[4 + rd][0]));
    r[INT_SBOX + rd] = state[0];
    orc::internal_layer(state);
    if (rd < 12) r[INT_S0 + rd] = state[0];
  }
  for (int i = 0; i < 16; i++) r[EXT_STATE + 64 + i] = state[i];   // external_rounds_state[4]
  for (int rd = 4; rd < 8; rd++) external_round(rd);
}
// events: n_events x 32 canonical words (input[16], output[16]); the row is rebuilt from the input, the output must agree
static inline std::vector<F> generate_poseidon2_wide(const F* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * POSEIDON2_WIDE_WIDTH, 0);
  const F zero[16] = {0};
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * POSEIDON2_WIDE_WIDTH;
    poseidon2_wide_row(i < n_events ? events + 32 * i : zero, r);
    if (i < n_events)
      for (int k = 0; k < 16; k++)
        if (r[156 + k] != events[32 * i + 16 + k]) throw std::runtime_error("tracegen: Poseidon2 event output is not the permutation of its input");
  }
  *height = h;
  return t;
}

// ---- SyscallInstrs chip: SyscallEvents (crates/core/executor/src/events/syscall.rs:7-29); columns syscall/instructions/columns.rs:9-58,
// row trace.rs:88-176. IsZeroOperation = (inverse, result) of a field element.
struct SyscallEvent {
  uint32_t pc, next_pc, shard, clk;
  MemoryWriteRecord a_record;
  uint8_t a_record_is_real; uint8_t pad[3];
  uint32_t syscall_id, arg1, arg2;
};
static_assert(sizeof(SyscallEvent) == 56, "SyscallEvent is fourteen words");
static const size_t SYSCALL_INSTRS_WIDTH = 77;
static inline void is_zero_cols(F a, F* r) { r[0] = a ? finv(a) : 0; r[1] = a == 0; }
static inline void syscall_instr_row(const SyscallEvent& e, F* r) {
  enum { PC = 0, NEXT_PC = 1, SHARD = 2, CLK = 3, NUM_EXTRA_CYCLES = 4, IS_HALT = 5, IS_SYS_LINUX = 6, IS_PREV_A1_ZERO = 7, SYSCALL_ID = 9, OP_A = 10,
         OP_B = 14, OP_C = 18, PREV_A = 22, IS_ENTER_UNCONSTRAINED = 26, IS_HINT_LEN = 28, IS_HALT_CHECK = 30, IS_EXIT_GROUP_CHECK = 32, IS_COMMIT = 34,
         IS_COMMIT_DEFERRED = 36, INDEX_BITMAP = 38, OP_B_RC = 46, OP_C_RC = 60, OP_B_CHECK = 74, OP_C_CHECK = 75, IS_REAL = 76 };
  static_assert(IS_REAL + 1 == 77, "layout");
  r[IS_REAL] = 1;
  r[PC] = fu32(e.pc); r[NEXT_PC] = fu32(e.next_pc); r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk);
  const uint32_t prev = e.a_record.prev_value;
  word(r + OP_A, e.a_record.value); word(r + OP_B, e.arg1); word(r + OP_C, e.arg2); word(r + PREV_A, prev);
  r[SYSCALL_ID] = fu32(e.syscall_id);
  const uint32_t id = prev & 0xffff;
  r[NUM_EXTRA_CYCLES] = (prev >> 24) & 0xff;
  const bool is_halt = id == 0 || id == 4246;
  r[IS_HALT] = is_halt;
  r[IS_SYS_LINUX] = (prev & 0xff00) != 0;
  const bool send_to_table = ((prev >> 8) & 0xff) != 0 || ((prev >> 16) & 0xff) == 1;
  is_zero_cols((prev >> 8) & 0xff, r + IS_PREV_A1_ZERO);
  is_zero_cols(fsub(id, 3), r + IS_ENTER_UNCONSTRAINED);
  is_zero_cols(fsub(id, 0xf0), r + IS_HINT_LEN);
  is_zero_cols(fsub(id, 0), r + IS_HALT_CHECK);
  is_zero_cols(fsub(id, 4246), r + IS_EXIT_GROUP_CHECK);
  is_zero_cols(fsub(id, 0x10), r + IS_COMMIT);
  is_zero_cols(fsub(id, 0x1a), r + IS_COMMIT_DEFERRED);
  if (id == 0x10 || id == 0x1a) {
    if (e.arg1 >= 8) throw std::runtime_error("tracegen: commit digest index out of range");
    r[INDEX_BITMAP + e.arg1] = 1;
  }
  if (send_to_table || is_halt) { r[OP_B_CHECK] = 1; range_checker(r + OP_B_RC, e.arg1); }
  if (send_to_table || id == 0x1a) { r[OP_C_CHECK] = 1; range_checker(r + OP_C_RC, e.arg2); }
}
static inline std::vector<F> generate_syscall_instrs(const SyscallEvent* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * SYSCALL_INSTRS_WIDTH, 0);
  for (size_t i = 0; i < n_events; i++) syscall_instr_row(events[i], t.data() + i * SYSCALL_INSTRS_WIDTH);
  *height = h;
  return t;
}

// ---- MiscInstrs chip: MiscEvents (crates/core/executor/src/events/instr.rs:239-261); columns misc/others/columns/*.rs (a union of four
// per-opcode layouts in columns 20..63), row misc/others/trace.rs:90-273, AddDoubleOperation operations/adddouble.rs:19-78
struct MiscEvent {
  uint32_t shard, clk, pc, next_pc;
  uint8_t opcode; uint8_t pad[3];
  uint32_t a, b, c, prev_a;
  MemoryWriteRecord hi_record;
};
static_assert(sizeof(MiscEvent) == 60, "MiscEvent is fifteen words");
static const size_t MISC_INSTRS_WIDTH = 72;
static inline void misc_instr_row(const MiscEvent& e, F* r, std::vector<ByteLookup>* lk) {
  enum { SHARD = 0, CLK = 1, PC = 2, NEXT_PC = 3, OP_A = 4, PREV_A = 8, OP_B = 12, OP_C = 16, SPECIFIC = 20, IS_SEXT = 64, IS_INS = 65, IS_EXT = 66,
         IS_MADDU = 67, IS_MSUBU = 68, IS_MADD = 69, IS_MSUB = 70, IS_TEQ = 71 };
  const uint8_t o = e.opcode;
  r[PC] = fu32(e.pc); r[NEXT_PC] = fu32(e.next_pc); r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk);
  word(r + OP_A, e.a); word(r + OP_B, e.b); word(r + OP_C, e.c); word(r + PREV_A, e.prev_a);
  r[IS_SEXT] = o == 55; r[IS_EXT] = o == 53; r[IS_INS] = o == 45; r[IS_MADDU] = o == 46; r[IS_MSUBU] = o == 47; r[IS_MADD] = o == 48;
  r[IS_MSUB] = o == 49; r[IS_TEQ] = o == 54;
  F* sp = r + SPECIFIC;
  if (o == 55 || o == 54) {   // SextCols: most_sig_bit, sig_byte, a_eq_b (11), is_seb, is_seh
    uint32_t bit, byte;
    if (e.c > 0) { sp[14] = 1; bit = (e.b & 0xffff) >> 15; byte = (e.b >> 8) & 0xff; }
    else { sp[13] = 1; bit = (e.b & 0xff) >> 7; byte = e.b & 0xff; }
    sp[0] = bit; sp[1] = byte;
    is_equal_word_cols(e.a, e.b, sp + 2);
    if (o == 55 && lk) lk->push_back(ByteLookup{B_MSB_OP, (uint8_t)byte, 0});
  } else if (o >= 46 && o <= 49) {   // MaddsubCols: mul_lo, mul_hi, add_operation (value, value_hi, carry[7]), src2_hi, src2_lo, op_hi_access
    const bool is_sign = o == 48 || o == 49, is_add = o == 46 || o == 48;
    const uint64_t multiply = is_sign ? (uint64_t)((int64_t)(int32_t)e.b * (int64_t)(int32_t)e.c) : (uint64_t)e.b * e.c;
    word(sp + 0, (uint32_t)multiply); word(sp + 4, (uint32_t)(multiply >> 32));
    const uint32_t src2_lo = is_add ? e.prev_a : e.a, src2_hi = is_add ? e.hi_record.prev_value : e.hi_record.value;
    const uint64_t bb = ((uint64_t)src2_hi << 32) + src2_lo, expected = multiply + bb;
    word(sp + 8, (uint32_t)expected); word(sp + 12, (uint32_t)(expected >> 32));
    uint32_t carry = 0;
    for (int i = 0; i < 7; i++) {
      carry = (((multiply >> (8 * i)) & 0xff) + ((bb >> (8 * i)) & 0xff) + carry) > 255;
      sp[16 + i] = carry;
    }
    if (lk) {
      F bytes[8];
      for (uint64_t v : {multiply, bb, expected}) {
        for (int i = 0; i < 8; i++) bytes[i] = (v >> (8 * i)) & 0xff;
        range_checks(*lk, bytes, 8);
      }
    }
    word(sp + 23, src2_hi); word(sp + 27, src2_lo);
    memory_write_cols(e.hi_record, sp + 31, lk);
  } else if (o == 53) {   // ExtCols: lsb, msbd, sll_val
    const uint32_t lsb = e.c & 0x1f, msbd = e.c >> 5;
    sp[0] = lsb; sp[1] = fu32(msbd);
    word(sp + 2, e.b << (31 - lsb - msbd));
    if (lk) {
      lk->push_back(ByteLookup{B_U8RANGE_OP, (uint8_t)lsb, (uint8_t)msbd});
      lk->push_back(ByteLookup{B_LTU_OP, (uint8_t)(lsb + msbd), 32});
    }
  } else if (o == 45) {   // InsCols: lsb, msb, ror_val, srl1_val, srl_val, sll_val, add_val
    const uint32_t lsb = e.c & 0x1f, msb = e.c >> 5;
    const uint32_t ror = lsb ? (e.prev_a >> lsb) | (e.prev_a << (32 - lsb)) : e.prev_a, srl1 = ror >> 1, srl = srl1 >> (msb - lsb);
    const uint32_t sll = e.b << (31 - msb + lsb);
    sp[0] = lsb; sp[1] = fu32(msb);
    word(sp + 2, ror); word(sp + 6, srl1); word(sp + 10, srl); word(sp + 14, sll); word(sp + 18, srl + sll);
    if (lk) {
      lk->push_back(ByteLookup{B_U8RANGE_OP, (uint8_t)lsb, (uint8_t)msb});
      lk->push_back(ByteLookup{B_LTU_OP, (uint8_t)lsb, (uint8_t)(msb + 1)});
      lk->push_back(ByteLookup{B_LTU_OP, (uint8_t)msb, 32});
    }
  } else {
    throw std::runtime_error("tracegen: invalid misc opcode");
  }
}
static inline std::vector<F> generate_misc_instrs(const MiscEvent* events, size_t n_events, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * MISC_INSTRS_WIDTH, 0);
  std::vector<ByteLookup> lk;
  for (size_t i = 0; i < n_events; i++) {
    lk.clear();
    misc_instr_row(events[i], t.data() + i * MISC_INSTRS_WIDTH, byte_counts ? &lk : nullptr);
    for (const ByteLookup& l : lk) byte_counts[(((size_t)l.b << 8) + l.c) * NUM_BYTE_OPS + l.op]++;
  }
  *height = h;
  return t;
}

// ---- recursion ExpReverseBitsLen chip (crates/recursion/core/src/chips/exp_reverse_bits.rs:175-226): an event is a base and its exponent
// bits; one row per bit: x, current_bit, prev_accum_squared, prev_accum_squared_times_multiplier, accum, accum_squared, multiplier with
// accum_i = accum_{i-1}^2 * (bit_i ? x : 1), accum_{-1} = 1. bases: n canonical words; bits: all events' bits end to end; offsets: n + 1.
static const size_t EXP_REVERSE_BITS_WIDTH = 7;
static inline std::vector<F> generate_exp_reverse_bits(const F* bases, const F* bits, const uint32_t* offsets, size_t n_events,
                                                       int fixed_log2_rows, size_t* height) {
  const size_t rows = n_events ? offsets[n_events] : 0, h = padded_rows(rows, fixed_log2_rows);
  std::vector<F> t(h * EXP_REVERSE_BITS_WIDTH, 0);
  for (size_t e = 0; e < n_events; e++) {
    F accum = 1;
    for (uint32_t i = offsets[e]; i < offsets[e + 1]; i++) {
      F* r = t.data() + (size_t)i * EXP_REVERSE_BITS_WIDTH;
      const F prev_sq = fmul(accum, accum), mult = bits[i] == 1 ? bases[e] : 1;
      accum = fmul(prev_sq, mult);
      r[0] = bases[e]; r[1] = bits[i]; r[2] = prev_sq; r[3] = accum; r[4] = accum; r[5] = fmul(accum, accum); r[6] = mult;
    }
  }
  *height = h;
  return t;
}

// ---- recursion Poseidon2Skinny chip (crates/recursion/core/src/chips/poseidon2_skinny/trace.rs:62-118): eleven rows per permutation —
// the state entering: the initial linear layer, external rounds 0..3, the internal rounds (that row also holds lane 0 after each of the
// first twelve), external rounds 4..7, and the output row; 28 columns (state_var[16], internal_rounds_s0[12]); zero padding
static const size_t SKINNY_WIDTH = 28, SKINNY_ROWS = 11;
static inline void poseidon2_skinny_rows(const F input[16], F* r /* 11 rows */) {
  F state[16];
  for (int i = 0; i < 16; i++) r[i] = state[i] = input[i];
  orc::external_layer(state);
  for (int row = 1; row <= 10; row++) {
    F* cur = r + row * SKINNY_WIDTH;
    for (int i = 0; i < 16; i++) cur[i] = state[i];
    if (row == 10) break;
    if (row == 5) {
      for (int rd = 0; rd < 13; rd++) {
        state[0] = orc::sbox(fadd(state[0], orc::ORC_RC_16_30[4 + rd][0]));
        orc::internal_layer(state);
        if (rd < 12) cur[16 + rd] = state[0];
      }
    } else {
      const int round = row < 5 ? row - 1 : row - 2 + 13;
      for (int i = 0; i < 16; i++) state[i] = orc::sbox(fadd(state[i], orc::ORC_RC_16_30[round][i]));
      orc::external_layer(state);
    }
  }
}
static inline std::vector<F> generate_poseidon2_skinny(const F* events, size_t n_events, int fixed_log2_rows, size_t* height) {
  const size_t h = padded_rows(n_events * SKINNY_ROWS, fixed_log2_rows);
  std::vector<F> t(h * SKINNY_WIDTH, 0);
  for (size_t e = 0; e < n_events; e++) {
    F* r = t.data() + e * SKINNY_ROWS * SKINNY_WIDTH;
    poseidon2_skinny_rows(events + 32 * e, r);
    for (int k = 0; k < 16; k++)
      if (r[10 * SKINNY_WIDTH + k] != events[32 * e + 16 + k]) throw std::runtime_error("tracegen: Poseidon2 event output is not the permutation of its input");
  }
  *height = h;
  return t;
}

// ---- Global chip (crates/core/machine/src/global/mod.rs): GlobalLookupEvents (crates/core/executor/src/events/global.rs:6-15,
// #[repr(C)]: message[7], is_receive, kind); columns :54-64 (message[7], kind, GlobalLookupOperation {offset_bits[8], x[7], y[7],
// y6_bit_decomp[30], range_check_witness}, is_receive, is_send, is_real, GlobalAccumulationOperation<1> {initial_digest[14],
// sum_checker[7], cumulative_sum[14]}); rows :120-197, operations/global_lookup.rs:26-92, operations/global_accumulation.rs:75-113.
// The running sum is a plain left-to-right loop (the reference scans in parallel with the complete addition law; two equal x
// in a row, probability ~2^-217, throw here). Byte lookups: U16Range(message[0]) per event (:75-95).
struct GlobalLookupEvent { uint32_t message[7]; uint8_t is_receive, kind, pad[2]; };
static_assert(sizeof(GlobalLookupEvent) == 32, "GlobalLookupEvent is eight words");
static const size_t GLOBAL_WIDTH = 99;
static inline std::vector<F> generate_global(const GlobalLookupEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                             uint64_t* byte_counts) {
  enum { MESSAGE = 0, KIND = 7, OFFSET_BITS = 8, X = 16, Y = 23, Y6_BITS = 30, RC_WITNESS = 60, IS_RECEIVE = 61, IS_SEND = 62, IS_REAL = 63,
         INITIAL = 64, SUM_CHECKER = 78, CUMULATIVE = 85 };
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * GLOBAL_WIDTH, 0);
  septic::Point sum, dummy;
  for (int k = 0; k < 7; k++) {
    sum.x.c[k] = SEPTIC_START_X[k]; sum.y.c[k] = SEPTIC_START_Y[k];
    dummy.x.c[k] = septic::DUMMY_X[k]; dummy.y.c[k] = septic::DUMMY_Y[k];
  }
  for (size_t i = 0; i < n_events; i++) {
    const GlobalLookupEvent& e = events[i];
    F* r = t.data() + i * GLOBAL_WIDTH;
    if (e.message[0] >> 16) throw std::runtime_error("global lookup: message[0] is not a u16");
    septic::S7 m;
    for (int k = 0; k < 7; k++) { r[MESSAGE + k] = fu32(e.message[k]); m.c[k] = r[MESSAGE + k]; }
    m.c[0] = fadd(m.c[0], (F)e.kind << 16);
    r[KIND] = e.kind;
    uint8_t offset;
    septic::Point pt = septic::lift_x(m, &offset);
    if (!e.is_receive) pt.y = septic::s_neg(pt.y);
    for (int k = 0; k < 8; k++) r[OFFSET_BITS + k] = (offset >> k) & 1;
    for (int k = 0; k < 7; k++) { r[X + k] = pt.x.c[k]; r[Y + k] = pt.y.c[k]; }
    const uint32_t rc = e.is_receive ? pt.y.c[6] - 1 : pt.y.c[6] - (P + 1) / 2;
    F top = 0;
    for (int k = 0; k < 30; k++) { r[Y6_BITS + k] = (rc >> k) & 1; if (k >= 23) top = fadd(top, r[Y6_BITS + k]); }
    r[RC_WITNESS] = finv(fsub(top, 7));
    r[IS_RECEIVE] = e.is_receive ? 1 : 0; r[IS_SEND] = e.is_receive ? 0 : 1; r[IS_REAL] = 1;
    septic::Point next = septic::add_incomplete(sum, pt);
    for (int k = 0; k < 7; k++) {
      r[INITIAL + k] = sum.x.c[k]; r[INITIAL + 7 + k] = sum.y.c[k];
      r[CUMULATIVE + k] = next.x.c[k]; r[CUMULATIVE + 7 + k] = next.y.c[k];
    }
    sum = next;
    if (byte_counts) byte_counts[(size_t)e.message[0] * NUM_BYTE_OPS + B_U16RANGE_OP]++;   // the table row of U16Range is its value
  }
  const septic::S7 final_checker = septic::sum_checker_x(sum, dummy, sum);
  for (size_t i = n_events; i < h; i++) {
    F* r = t.data() + i * GLOBAL_WIDTH;
    for (int k = 0; k < 7; k++) {
      r[X + k] = dummy.x.c[k]; r[Y + k] = dummy.y.c[k];
      r[INITIAL + k] = sum.x.c[k]; r[INITIAL + 7 + k] = sum.y.c[k];
      r[SUM_CHECKER + k] = final_checker.c[k];
      r[CUMULATIVE + k] = sum.x.c[k]; r[CUMULATIVE + 7 + k] = sum.y.c[k];
    }
  }
  *height = h;
  return t;
}

// ---- MemoryGlobalInit / MemoryGlobalFinalize chips (memory/global.rs): MemoryInitializeFinalizeEvents (crates/core/executor/src/events/
// memory.rs:180-209: addr, value, shard, timestamp); columns MemoryInitCols :221-259 (shard, timestamp, addr, lt_cols.bit_flags[32],
// addr_bits: bits[32] + six running products of the top byte's bits, value[32] bits, is_real, is_next_comp, is_prev_addr_zero (inverse,
// result), is_first_comp, is_last_addr); rows generate_trace :113-185: events sorted by address, row 0 compared with the previous shard's
// last address (public values), every later row with its predecessor. The C++ twin is include/memory_global.hpp:9-43.
struct MemoryInitFinalizeEvent { uint32_t addr, value, shard, timestamp; };
static_assert(sizeof(MemoryInitFinalizeEvent) == 16, "MemoryInitializeFinalizeEvent is four words");
static const size_t MEMORY_GLOBAL_WIDTH = 111;
// AssertLtColsBits::populate (operations/cmp.rs:301-319): the flag of the most significant bit where a < b
static inline void assert_lt_bits(uint32_t a, uint32_t b, F* flags) {
  if (!(a < b)) throw std::runtime_error("tracegen: memory init/finalize addresses are not strictly increasing");
  for (int i = 31; i >= 0; i--) {
    const uint32_t ab = (a >> i) & 1, bb = (b >> i) & 1;
    if (ab < bb) { flags[i] = 1; break; }
  }
}
static inline std::vector<F> generate_memory_global(const MemoryInitFinalizeEvent* events_in, size_t n_events, uint32_t previous_addr,
                                                    int fixed_log2_rows, size_t* height) {
  enum { SHARD = 0, TIMESTAMP = 1, ADDR = 2, LT = 3, ADDR_BITS = 35, AND_DECOMP = 67, VALUE = 73, IS_REAL = 105, IS_NEXT_COMP = 106,
         IS_PREV_ADDR_ZERO = 107, IS_FIRST_COMP = 109, IS_LAST_ADDR = 110 };
  static_assert(IS_LAST_ADDR + 1 == 111, "layout");
  std::vector<MemoryInitFinalizeEvent> events(events_in, events_in + n_events);
  std::stable_sort(events.begin(), events.end(), [](const MemoryInitFinalizeEvent& a, const MemoryInitFinalizeEvent& b) { return a.addr < b.addr; });
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * MEMORY_GLOBAL_WIDTH, 0);
  for (size_t i = 0; i < n_events; i++) {
    const MemoryInitFinalizeEvent& e = events[i];
    F* r = t.data() + i * MEMORY_GLOBAL_WIDTH;
    r[ADDR] = fu32(e.addr);
    for (int k = 0; k < 32; k++) r[ADDR_BITS + k] = (e.addr >> k) & 1;
    r[AND_DECOMP] = r[ADDR_BITS + 24] * r[ADDR_BITS + 25];
    for (int k = 0; k < 5; k++) r[AND_DECOMP + 1 + k] = r[AND_DECOMP + k] * r[ADDR_BITS + 26 + k];
    r[SHARD] = fu32(e.shard);
    r[TIMESTAMP] = fu32(e.timestamp);
    for (int k = 0; k < 32; k++) r[VALUE + k] = (e.value >> k) & 1;
    r[IS_REAL] = 1;
    if (i == 0) {
      is_zero_cols(fu32(previous_addr), r + IS_PREV_ADDR_ZERO);
      r[IS_FIRST_COMP] = previous_addr != 0;
      if (previous_addr != 0) assert_lt_bits(previous_addr, e.addr, r + LT);
    } else {
      r[IS_NEXT_COMP] = 1;
      assert_lt_bits(events[i - 1].addr, e.addr, r + LT);
    }
    if (i == n_events - 1) r[IS_LAST_ADDR] = 1;
  }
  *height = h;
  return t;
}

// ---- SyscallCore / SyscallPrecompile chips (syscall/chip.rs): SyscallEvents; columns SyscallCols :71-107 (shard, clk, syscall_id,
// arg1_lo, arg1_hi, arg2_lo, arg2_hi, result_lo, result_hi, is_linux, is_real); rows generate_trace :211-276. Core: the shard's syscall
// events whose code (a_record.prev_value) has the send-to-table byte set or names a Linux syscall; Precompile: the syscall events filed
// with the shard's precompile events (a default a_record — except for Linux events, chip.rs:223-238: is_linux and the result come from the LinuxEvent;
// across this boundary the syscall event of a Linux call carries them in its a_record: prev_value = the code, value = v0).
// The C++ twin is include/syscall.hpp:9-60. Byte lookups (generate_dependencies :115-187): U16Range of the four argument half-words.
static const size_t SYSCALL_WIDTH = 11;
static inline bool syscall_goes_to_table(const SyscallEvent& e) {
  const uint32_t prev = e.a_record.prev_value;
  return ((prev >> 16) & 0xff) == 1 || ((prev >> 8) & 0xff) != 0;
}
static inline std::vector<F> generate_syscall(const SyscallEvent* events, size_t n_events, bool precompile, int fixed_log2_rows, size_t* height,
                                              uint64_t* byte_counts) {
  std::vector<const SyscallEvent*> kept;
  for (size_t i = 0; i < n_events; i++)
    if (precompile || syscall_goes_to_table(events[i])) kept.push_back(events + i);
  const size_t h = padded_rows(kept.size(), fixed_log2_rows);
  std::vector<F> t(h * SYSCALL_WIDTH, 0);
  for (size_t i = 0; i < kept.size(); i++) {
    const SyscallEvent& e = *kept[i];
    F* r = t.data() + i * SYSCALL_WIDTH;
    r[0] = fu32(e.shard); r[1] = fu32(e.clk); r[2] = fu32(e.syscall_id);
    r[3] = e.arg1 & 0xffff; r[4] = e.arg1 >> 16; r[5] = e.arg2 & 0xffff; r[6] = e.arg2 >> 16;
    const bool is_linux = ((e.a_record.prev_value >> 8) & 0xff) != 0;      // Precompile: the a_record of a Linux event's syscall event carries its code and v0
    r[9] = is_linux;
    if (is_linux) { r[7] = e.a_record.value & 0xffff; r[8] = e.a_record.value >> 16; }
    r[10] = 1;
    if (byte_counts)
      for (int k = 3; k < 7; k++) byte_counts[(size_t)r[k] * NUM_BYTE_OPS + B_U16RANGE_OP]++;   // the table row of U16Range is its value
  }
  *height = h;
  return t;
}

// ---- Poseidon2Permute precompile chip (syscall/precompiles/poseidon2/): columns Poseidon2MemCols columns.rs:9-27 = the degree-3
// permutation columns of operations/poseidon2 (the same 313 as the recursion chip above: permutation.rs:43-80, trace.rs:13-157), shard, clk,
// state_addr, sixteen MemoryWriteCols (13 each), KoalaBearWordRangeChecker of every pre- and post-state word (14 each), is_real; rows
// trace.rs:31-128: the padding row carries the permutation of the zero state and nothing else. Event: Poseidon2PermuteEvent
// (crates/core/executor/src/events/precompiles/poseidon2_permute.rs:9-27) flattened: shard, clk, state_addr, then the sixteen
// MemoryWriteRecords of the state words (pre_state = their prev_value, post_state = their value, as syscalls/precompiles/poseidon2/
// permute.rs:30-47 makes them).
struct Poseidon2PermuteEvent { uint32_t shard, clk, state_addr; MemoryWriteRecord state_records[16]; };
static_assert(sizeof(Poseidon2PermuteEvent) == 4 * 99, "flattened Poseidon2PermuteEvent is 99 words");
static const size_t POSEIDON2_PERMUTE_WIDTH = 973;
static inline std::vector<F> generate_poseidon2_permute(const Poseidon2PermuteEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                                        uint64_t* byte_counts) {
  enum { SHARD = 313, CLK = 314, STATE_ADDR = 315, STATE_MEM = 316, PRE_RC = 524, POST_RC = 748, IS_REAL = 972 };
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * POSEIDON2_PERMUTE_WIDTH, 0);
  std::vector<ByteLookup> lk;
  const F zero[16] = {0};
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * POSEIDON2_PERMUTE_WIDTH;
    if (i >= n_events) { poseidon2_wide_row(zero, r); continue; }
    const Poseidon2PermuteEvent& e = events[i];
    F input[16];
    for (int k = 0; k < 16; k++) {
      if (e.state_records[k].prev_value >= P || e.state_records[k].value >= P) throw std::runtime_error("tracegen: Poseidon2 state word is not a field element");
      input[k] = e.state_records[k].prev_value;
    }
    poseidon2_wide_row(input, r);
    for (int k = 0; k < 16; k++)
      if (r[156 + k] != e.state_records[k].value) throw std::runtime_error("tracegen: Poseidon2PermuteEvent post-state is not the permutation of its pre-state");
    r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[STATE_ADDR] = fu32(e.state_addr); r[IS_REAL] = 1;
    for (int k = 0; k < 16; k++) {
      memory_write_cols(e.state_records[k], r + STATE_MEM + 13 * k, &lk);
      range_checker(r + PRE_RC + 14 * k, e.state_records[k].prev_value);
      range_checker(r + POST_RC + 14 * k, e.state_records[k].value);
    }
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}


// ---- KeccakSponge precompile chip (syscall/precompiles/keccak_sponge/): columns KeccakSpongeCols columns.rs:17-37 = the 2633 KeccakCols of
// p3-keccak-air, thirty-six MemoryReadCols of the block (9 each), shard, clk, is_real, read_block, input_address, output_address, input_len,
// already_absorbed_u32s, is_absorbed, receive_syscall, write_output, is_first_input_block, is_final_input_block, the fifty words of the state
// the block is absorbed into, thirty-six XorOperations, the MemoryReadCols of the input length and sixteen MemoryWriteCols of the output:
// 3531 columns, twenty-four rows per 36-word block (trace.rs:102-195); padding rows are the rounds of the permutation of the zero state,
// row i carrying round i mod 24 (trace.rs:79-93).
//
// p3-keccak-air (git dependency github.com/ProjectZKM/Plonky3, not vendored under /root/reference: Cargo.toml:62) is restated here from the
// published crate (columns.rs, generate.rs, constants.rs): KeccakCols = step_flags[24], export, preimage[5][5][4], a[5][5][4], c[5][64],
// c_prime[5][64], a_prime[5][5][64], a_prime_prime[5][5][4], a_prime_prime_0_0_bits[64], a_prime_prime_prime_0_0_limbs[4], arrays indexed
// [y][x], 16-bit limbs. Its column count is pinned by the reference's cost table (mips_costs.json: KeccakSponge 102216 = 24 x 4259, which
// only a 2633-column KeccakCols gives), its values by Keccak-256 known answers (tests); `export` is left 0 as generate_trace_rows leaves it.
// PARITY UNPINNED for anything else the fork may have changed in that crate.
static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
static const int KECCAK_ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
static inline uint64_t rotl64(uint64_t v, int r) { return r ? (v << r) | (v >> (64 - r)) : v; }
enum { KC_STEP = 0, KC_EXPORT = 24, KC_PREIMAGE = 25, KC_A = 125, KC_C = 225, KC_C_PRIME = 545, KC_A_PRIME = 865, KC_A_PP = 2465, KC_A_PP_00_BITS = 2565,
       KC_A_PPP_00 = 2629, NUM_KECCAK_COLS = 2633 };
// One round of keccak-f[1600] on a[y * 5 + x], writing the round's KeccakCols at r (generate_trace_row_for_round); returns with `a` = the next round's input.
static inline void keccak_round_cols(uint64_t a[25], const uint64_t preimage[25], int round, F* r) {
  r[KC_STEP + round] = 1;
  for (int i = 0; i < 25; i++)
    for (int l = 0; l < 4; l++) {
      r[KC_PREIMAGE + 4 * i + l] = (preimage[i] >> (16 * l)) & 0xffff;
      r[KC_A + 4 * i + l] = (a[i] >> (16 * l)) & 0xffff;
    }
  uint64_t c[5], cp[5], ap[25], app[25];
  for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[5 + x] ^ a[10 + x] ^ a[15 + x] ^ a[20 + x];
  for (int x = 0; x < 5; x++) {
    const uint64_t d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);     // bit z of rotl(c, 1) is bit z - 1 of c
    cp[x] = c[x] ^ d;
    for (int y = 0; y < 5; y++) ap[5 * y + x] = a[5 * y + x] ^ d;
  }
  for (int x = 0; x < 5; x++)
    for (int z = 0; z < 64; z++) {
      r[KC_C + 64 * x + z] = (c[x] >> z) & 1;
      r[KC_C_PRIME + 64 * x + z] = (cp[x] >> z) & 1;
    }
  for (int i = 0; i < 25; i++)
    for (int z = 0; z < 64; z++) r[KC_A_PRIME + 64 * i + z] = (ap[i] >> z) & 1;
  // B[x, y] = rotl(A'[(x + 3y) mod 5, x], R[(x + 3y) mod 5][x]);  A''[x, y] = B[x, y] ^ (~B[x + 1, y] & B[x + 2, y])
  auto b = [&](int x, int y) { const int xa = (x + 3 * y) % 5; return rotl64(ap[5 * x + xa], KECCAK_ROT[xa][x]); };
  for (int y = 0; y < 5; y++)
    for (int x = 0; x < 5; x++) app[5 * y + x] = b(x, y) ^ (~b((x + 1) % 5, y) & b((x + 2) % 5, y));
  for (int i = 0; i < 25; i++)
    for (int l = 0; l < 4; l++) r[KC_A_PP + 4 * i + l] = (app[i] >> (16 * l)) & 0xffff;
  for (int z = 0; z < 64; z++) r[KC_A_PP_00_BITS + z] = (app[0] >> z) & 1;
  app[0] ^= KECCAK_RC[round];
  for (int l = 0; l < 4; l++) r[KC_A_PPP_00 + l] = (app[0] >> (16 * l)) & 0xffff;
  for (int i = 0; i < 25; i++) a[i] = app[i];
}
// One 36-word block of a KeccakSpongeEvent (crates/core/executor/src/events/precompiles/keccak_sponge.rs:15-46), the event's Vecs cut per
// block for the C ABI: the state after the block is xored in (xored_state_list[block_index] as u32 pairs), the block's read records
// (their values are the input words), and — used on the first / last block only — the record of the input length and the output writes.
struct KeccakSpongeBlock {
  uint32_t shard, clk, input_addr, output_addr, input_len_u32s, block_index;
  uint32_t xored_state[50];
  MemoryReadRecord input_read_records[36];
  MemoryReadRecord input_length_record;
  MemoryWriteRecord output_write_records[16];
};
static_assert(sizeof(KeccakSpongeBlock) == 4 * 337, "flattened KeccakSpongeEvent block is 337 words");
static const size_t KECCAK_SPONGE_WIDTH = 3531;
static inline std::vector<F> generate_keccak_sponge(const KeccakSpongeBlock* blocks, size_t n_blocks, int fixed_log2_rows, size_t* height,
                                                    uint64_t* byte_counts) {
  enum { BLOCK_MEM = 2633, SHARD = 2957, CLK = 2958, IS_REAL = 2959, READ_BLOCK = 2960, INPUT_ADDRESS = 2961, OUTPUT_ADDRESS = 2962, INPUT_LEN = 2963,
         ALREADY_ABSORBED = 2964, IS_ABSORBED = 2965, RECEIVE_SYSCALL = 2966, WRITE_OUTPUT = 2967, IS_FIRST = 2968, IS_FINAL = 2969, ORIGINAL_STATE = 2970,
         XORED_RATE = 3170, INPUT_LENGTH_MEM = 3314, OUTPUT_MEM = 3323 };
  static_assert(OUTPUT_MEM + 16 * 13 == 3531, "layout");
  const size_t h = padded_rows(24 * n_blocks, fixed_log2_rows);
  std::vector<F> t(h * KECCAK_SPONGE_WIDTH, 0);
  std::vector<ByteLookup> lk;
  for (size_t k = 0; k < n_blocks; k++) {
    const KeccakSpongeBlock& e = blocks[k];
    if (e.input_len_u32s == 0 || e.input_len_u32s % 36 != 0 || e.block_index >= e.input_len_u32s / 36)
      throw std::runtime_error("tracegen: KeccakSponge block index / input length");
    const uint32_t n_ev_blocks = e.input_len_u32s / 36;
    const bool first = e.block_index == 0, final = e.block_index == n_ev_blocks - 1;
    uint32_t before[50];                      // the state the block is absorbed into
    for (int j = 0; j < 50; j++) before[j] = e.xored_state[j] ^ (j < 36 ? e.input_read_records[j].value : 0u);
    if (first)
      for (int j = 0; j < 50; j++)
        if (before[j]) throw std::runtime_error("tracegen: KeccakSponge first block is not absorbed into the zero state");
    if (first && e.input_length_record.value != e.input_len_u32s) throw std::runtime_error("tracegen: KeccakSponge input length record");
    uint64_t a[25], pre[25];
    for (int i = 0; i < 25; i++) a[i] = pre[i] = (uint64_t)e.xored_state[2 * i] | ((uint64_t)e.xored_state[2 * i + 1] << 32);
    for (int round = 0; round < 24; round++) {
      F* r = t.data() + (24 * k + round) * KECCAK_SPONGE_WIDTH;
      keccak_round_cols(a, pre, round, r);
      r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[IS_REAL] = 1;
      r[INPUT_LEN] = fu32(e.input_len_u32s);
      r[ALREADY_ABSORBED] = fu32(36 * e.block_index);
      r[IS_ABSORBED] = round == 23 && !final;
      r[IS_FIRST] = first; r[IS_FINAL] = final;
      r[READ_BLOCK] = round == 0;
      r[RECEIVE_SYSCALL] = first && round == 0;
      r[WRITE_OUTPUT] = final && round == 23;
      r[OUTPUT_ADDRESS] = fu32(e.output_addr);
      r[INPUT_ADDRESS] = fu32(e.input_addr + e.block_index * 144);
      for (int j = 0; j < 50; j++) word(r + ORIGINAL_STATE + 4 * j, before[j]);
      if (round == 0)
        for (int j = 0; j < 36; j++) {
          const MemoryReadRecord& m = e.input_read_records[j];
          memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + BLOCK_MEM + 9 * j, &lk);
        }
      if (round == 0)
        for (int j = 0; j < 36; j++) {       // XorOperation::populate (operations/xor.rs:19-37)
          word(r + XORED_RATE + 4 * j, e.xored_state[j]);
          for (int b = 0; b < 4; b++) lk.push_back(ByteLookup{B_XOR, (uint8_t)(before[j] >> (8 * b)), (uint8_t)(e.input_read_records[j].value >> (8 * b))});
        }
      if (first && round == 0) {
        const MemoryReadRecord& m = e.input_length_record;
        memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + INPUT_LENGTH_MEM, &lk);
      }
      if (final && round == 23)
        for (int j = 0; j < 16; j++) {
          if (e.output_write_records[j].value != (uint32_t)(a[j / 2] >> (32 * (j & 1)))) throw std::runtime_error("tracegen: KeccakSponge output is not the squeezed state");
          memory_write_cols(e.output_write_records[j], r + OUTPUT_MEM + 13 * j, &lk);
        }
    }
    if (!final) {      // the next block of the event follows and is absorbed into this block's permuted state
      if (k + 1 >= n_blocks) throw std::runtime_error("tracegen: KeccakSponge event is cut short");
      const KeccakSpongeBlock& nx = blocks[k + 1];
      bool ok = nx.block_index == e.block_index + 1 && nx.input_len_u32s == e.input_len_u32s && nx.shard == e.shard && nx.clk == e.clk;
      for (int j = 0; j < 50 && ok; j++)
        ok = (nx.xored_state[j] ^ (j < 36 ? nx.input_read_records[j].value : 0u)) == (uint32_t)(a[j / 2] >> (32 * (j & 1)));
      if (!ok) throw std::runtime_error("tracegen: KeccakSponge blocks of one event do not chain");
    }
  }
  const uint64_t zero[25] = {0};
  std::vector<F> dummy(24 * NUM_KECCAK_COLS, 0);
  {
    uint64_t a[25] = {0};
    for (int round = 0; round < 24; round++) keccak_round_cols(a, zero, round, dummy.data() + round * NUM_KECCAK_COLS);
  }
  for (size_t i = 24 * n_blocks; i < h; i++)
    std::copy(dummy.begin() + (i % 24) * NUM_KECCAK_COLS, dummy.begin() + (i % 24 + 1) * NUM_KECCAK_COLS, t.begin() + i * KECCAK_SPONGE_WIDTH);
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

// ---- SHA-256 precompile chips (syscall/precompiles/sha256/): ShaExtend (48 rows per call: one message-schedule word each) and
// ShaCompress (80 rows per call: 8 that load the state, 64 rounds, 8 that add and store). Operation column groups (operations/):
// FixedRotateRightOperation / FixedShiftRightOperation = value, shift, carry words (fixed_rotate_right.rs:13-22, fixed_shift_right.rs);
// Xor / And / Not = the result word; Add4 / Add5 = value, one-hot carry flags per byte, carry (add4.rs:13-28, add5.rs); AddOperation =
// value + three carries (add.rs). Byte lookups as each populate() records them.
struct ShaOps {
  std::vector<ByteLookup>* lk;
  void range(uint32_t v) const {       // ByteRecord::add_u8_range_checks of a word: two lookups of byte pairs
    lk->push_back(ByteLookup{B_U8RANGE, (uint8_t)v, (uint8_t)(v >> 8)});
    lk->push_back(ByteLookup{B_U8RANGE, (uint8_t)(v >> 16), (uint8_t)(v >> 24)});
  }
  // fixed_rotate_right.rs:37-88 / fixed_shift_right.rs:37-84: bytes moved by rotation / 8, then every byte shifted by rotation % 8 with
  // the bits that fall out carried into the byte below
  uint32_t shift_or_rotate(F* r, uint32_t x, int rotation, bool rotate) const {
    const int nbytes = rotation / 8, nbits = rotation % 8;
    uint8_t in[4];
    for (int i = 0; i < 4; i++) in[i] = rotate ? (uint8_t)(x >> (8 * ((i + nbytes) % 4))) : (i + nbytes < 4 ? (uint8_t)(x >> (8 * (i + nbytes))) : 0);
    uint32_t first_shift = 0, last_carry = 0;
    for (int i = 3; i >= 0; i--) {
      const uint32_t shift = nbits ? in[i] >> nbits : in[i], carry = nbits ? in[i] & ((1u << nbits) - 1) : 0;
      lk->push_back(ByteLookup{B_SHRCARRY, in[i], (uint8_t)nbits});
      r[4 + i] = shift; r[8 + i] = carry;
      if (i == 3) first_shift = shift; else r[i] = shift + last_carry * (1u << (8 - nbits));
      last_carry = carry;
    }
    r[3] = rotate ? first_shift + last_carry * (1u << (8 - nbits)) : first_shift;
    const uint32_t out = (uint32_t)r[0] | ((uint32_t)r[1] << 8) | ((uint32_t)r[2] << 16) | ((uint32_t)r[3] << 24);
    if (out != (rotate ? (x >> rotation) | (x << (32 - rotation)) : x >> rotation)) throw std::runtime_error("tracegen: fixed rotate / shift");
    return out;
  }
  uint32_t bitwise(F* r, int op, uint32_t x, uint32_t y) const {    // xor.rs / and.rs: the result word, one lookup per byte
    const uint32_t out = op == B_XOR ? x ^ y : x & y;
    word(r, out);
    for (int i = 0; i < 4; i++) lk->push_back(ByteLookup{(uint8_t)op, (uint8_t)(x >> (8 * i)), (uint8_t)(y >> (8 * i))});
    return out;
  }
  uint32_t not_(F* r, uint32_t x) const { word(r, ~x); range(x); return ~x; }     // not.rs:16-25
  // add4.rs:31-73 / add5.rs: value, is_carry_0..k (one-hot per byte), carry
  uint32_t add_many(F* r, const uint32_t* v, int n) const {
    uint32_t sum = 0;
    for (int k = 0; k < n; k++) sum += v[k];
    word(r, sum);
    uint32_t carry = 0;
    for (int i = 0; i < 4; i++) {
      uint32_t res = carry;
      for (int k = 0; k < n; k++) res += (v[k] >> (8 * i)) & 0xff;
      carry = res >> 8;
      for (int c = 0; c < n; c++) r[4 + 4 * c + i] = carry == (uint32_t)c;
      r[4 + 4 * n + i] = carry;
    }
    for (int k = 0; k < n; k++) range(v[k]);
    range(sum);
    return sum;
  }
  uint32_t add(F* r, uint32_t a, uint32_t b) const {                // add.rs:21-57: value, three carries
    word(r, a + b);
    uint32_t carry = 0;
    for (int i = 0; i < 3; i++) {
      carry = (((a >> (8 * i)) & 0xff) + ((b >> (8 * i)) & 0xff) + carry) > 255;
      r[4 + i] = carry;
    }
    range(a); range(b); range(a + b);
    return a + b;
  }
};

// ShaExtendEvent (crates/core/executor/src/events/precompiles/sha256_extend.rs:9-24) with its Vecs at their fixed length of 48
struct ShaExtendEvent {
  uint32_t shard, clk, w_ptr;
  MemoryReadRecord w_i_minus_15_reads[48], w_i_minus_2_reads[48], w_i_minus_16_reads[48], w_i_minus_7_reads[48];
  MemoryWriteRecord w_i_writes[48];
};
static_assert(sizeof(ShaExtendEvent) == 4 * 1251, "flattened ShaExtendEvent is 1251 words");
static const size_t SHA_EXTEND_WIDTH = 176;
// ShaExtendCols::populate_flags (extend/flags.rs:13-38): g = the generator of the order-16 subgroup
static inline void sha_extend_flags(F* r, size_t i, bool is_real) {
  enum { I = 3, CYCLE_16 = 4, CYCLE_16_START = 5, CYCLE_16_END = 7, CYCLE_48 = 9, CYCLE_48_START = 12, CYCLE_48_END = 13 };
  const F g = two_adic_generator(4);
  r[CYCLE_16] = fpow(g, (i + 1) % 16);
  is_zero_cols(fsub(r[CYCLE_16], g), r + CYCLE_16_START);
  is_zero_cols(fsub(r[CYCLE_16], 1), r + CYCLE_16_END);
  const size_t j = 16 + (i % 48);
  r[I] = j;
  r[CYCLE_48] = j < 32; r[CYCLE_48 + 1] = j >= 32 && j < 48; r[CYCLE_48 + 2] = j >= 48;
  r[CYCLE_48_START] = r[CYCLE_48] * r[CYCLE_16_START + 1] * is_real;
  r[CYCLE_48_END] = r[CYCLE_48 + 2] * r[CYCLE_16_END + 1] * is_real;
}
static inline std::vector<F> generate_sha_extend(const ShaExtendEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                                 uint64_t* byte_counts) {
  enum { SHARD = 0, CLK = 1, W_PTR = 2, W_I_MINUS_15 = 14, RR_7 = 23, RR_18 = 35, RS_3 = 47, S0_INTERMEDIATE = 59, S0 = 63, W_I_MINUS_2 = 67, RR_17 = 76,
         RR_19 = 88, RS_10 = 100, S1_INTERMEDIATE = 112, S1 = 116, W_I_MINUS_16 = 120, W_I_MINUS_7 = 129, S2 = 138, W_I = 162, IS_REAL = 175 };
  const size_t h = padded_rows(48 * n_events, fixed_log2_rows);
  std::vector<F> t(h * SHA_EXTEND_WIDTH, 0);
  std::vector<ByteLookup> lk;
  const ShaOps ops{&lk};
  auto read = [&](const MemoryReadRecord& m, F* r) { memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r, &lk); };
  for (size_t row = 0; row < h; row++) {
    F* r = t.data() + row * SHA_EXTEND_WIDTH;
    if (row >= 48 * n_events) { sha_extend_flags(r, row, false); continue; }
    const ShaExtendEvent& e = events[row / 48];
    const size_t j = row % 48;
    r[IS_REAL] = 1;
    sha_extend_flags(r, j, true);
    r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[W_PTR] = fu32(e.w_ptr);
    read(e.w_i_minus_15_reads[j], r + W_I_MINUS_15);
    read(e.w_i_minus_2_reads[j], r + W_I_MINUS_2);
    read(e.w_i_minus_16_reads[j], r + W_I_MINUS_16);
    read(e.w_i_minus_7_reads[j], r + W_I_MINUS_7);
    const uint32_t w15 = e.w_i_minus_15_reads[j].value, w2 = e.w_i_minus_2_reads[j].value;
    const uint32_t s0 = ops.bitwise(r + S0, B_XOR, ops.bitwise(r + S0_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + RR_7, w15, 7, true),
                                                               ops.shift_or_rotate(r + RR_18, w15, 18, true)),
                                    ops.shift_or_rotate(r + RS_3, w15, 3, false));
    const uint32_t s1 = ops.bitwise(r + S1, B_XOR, ops.bitwise(r + S1_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + RR_17, w2, 17, true),
                                                               ops.shift_or_rotate(r + RR_19, w2, 19, true)),
                                    ops.shift_or_rotate(r + RS_10, w2, 10, false));
    const uint32_t four[4] = {e.w_i_minus_16_reads[j].value, s0, e.w_i_minus_7_reads[j].value, s1};
    const uint32_t w_i = ops.add_many(r + S2, four, 4);
    if (e.w_i_writes[j].value != w_i) throw std::runtime_error("tracegen: ShaExtendEvent write is not the schedule word");
    memory_write_cols(e.w_i_writes[j], r + W_I, &lk);
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

// ShaCompressEvent (events/precompiles/sha256_compress.rs:9-25) flattened: w and h are the values of the read records
struct ShaCompressEvent {
  uint32_t shard, clk, w_ptr, h_ptr;
  MemoryReadRecord h_read_records[8], w_i_read_records[64];
  MemoryWriteRecord h_write_records[8];
};
static_assert(sizeof(ShaCompressEvent) == 4 * 412, "flattened ShaCompressEvent is 412 words");
static const size_t SHA_COMPRESS_WIDTH = 262;
static const uint32_t SHA_COMPRESS_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline std::vector<F> generate_sha_compress(const ShaCompressEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                                   uint64_t* byte_counts) {
  enum { SHARD = 0, CLK = 1, W_PTR = 2, H_PTR = 3, START = 4, OCTET = 5, OCTET_NUM = 13, MEM = 23, MEM_ADDR = 36, A = 37, K = 69, E_RR_6 = 73, E_RR_11 = 85,
         E_RR_25 = 97, S1_INTERMEDIATE = 109, S1 = 113, E_AND_F = 117, E_NOT = 121, E_NOT_AND_G = 125, CH = 129, TEMP1 = 133, A_RR_2 = 161, A_RR_13 = 173,
         A_RR_22 = 185, S0_INTERMEDIATE = 197, S0 = 201, A_AND_B = 205, A_AND_C = 209, B_AND_C = 213, MAJ_INTERMEDIATE = 217, MAJ = 221, TEMP2 = 225,
         D_ADD_TEMP1 = 232, TEMP1_ADD_TEMP2 = 239, FINALIZED_OPERAND = 246, FINALIZE_ADD = 250, IS_INITIALIZE = 257, IS_COMPRESSION = 258,
         IS_FINALIZE = 259, IS_LAST_ROW = 260, IS_REAL = 261 };
  const size_t h = padded_rows(80 * n_events, fixed_log2_rows);
  std::vector<F> t(h * SHA_COMPRESS_WIDTH, 0);
  std::vector<ByteLookup> lk;
  const ShaOps ops{&lk};
  for (size_t ev = 0; ev < n_events; ev++) {
    const ShaCompressEvent& e = events[ev];
    uint32_t v[8], og[8];
    for (int i = 0; i < 8; i++) v[i] = og[i] = e.h_read_records[i].value;
    for (int step = 0; step < 80; step++) {
      F* r = t.data() + (80 * ev + step) * SHA_COMPRESS_WIDTH;
      const int octet = step % 8, octet_num = step / 8;
      r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[W_PTR] = fu32(e.w_ptr); r[H_PTR] = fu32(e.h_ptr);
      r[OCTET + octet] = 1; r[OCTET_NUM + octet_num] = 1; r[IS_REAL] = 1;
      r[START] = step == 0;
      if (octet_num == 0) {              // the state words are read (trace.rs:138-168)
        r[IS_INITIALIZE] = 1;
        const MemoryReadRecord& m = e.h_read_records[octet];
        word(r + MEM, m.value);
        memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + MEM + 4, &lk);
        r[MEM_ADDR] = fu32(e.h_ptr + 4 * octet);
        for (int i = 0; i < 8; i++) word(r + A + 4 * i, v[i]);
      } else if (octet_num < 9) {        // one round (:171-246)
        const int j = step - 8;
        word(r + K, SHA_COMPRESS_K[j]);
        r[IS_COMPRESSION] = 1;
        const MemoryReadRecord& m = e.w_i_read_records[j];
        word(r + MEM, m.value);
        memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + MEM + 4, &lk);
        r[MEM_ADDR] = fu32(e.w_ptr + 4 * j);
        for (int i = 0; i < 8; i++) word(r + A + 4 * i, v[i]);
        const uint32_t a = v[0], b = v[1], c = v[2], d = v[3], ee = v[4], f = v[5], g = v[6], hh = v[7];
        const uint32_t s1 = ops.bitwise(r + S1, B_XOR, ops.bitwise(r + S1_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + E_RR_6, ee, 6, true),
                                                                   ops.shift_or_rotate(r + E_RR_11, ee, 11, true)),
                                        ops.shift_or_rotate(r + E_RR_25, ee, 25, true));
        const uint32_t e_and_f = ops.bitwise(r + E_AND_F, B_AND, ee, f);
        const uint32_t e_not = ops.not_(r + E_NOT, ee);
        const uint32_t ch = ops.bitwise(r + CH, B_XOR, e_and_f, ops.bitwise(r + E_NOT_AND_G, B_AND, e_not, g));
        const uint32_t five[5] = {hh, s1, ch, m.value, SHA_COMPRESS_K[j]};
        const uint32_t temp1 = ops.add_many(r + TEMP1, five, 5);
        const uint32_t s0 = ops.bitwise(r + S0, B_XOR, ops.bitwise(r + S0_INTERMEDIATE, B_XOR, ops.shift_or_rotate(r + A_RR_2, a, 2, true),
                                                                   ops.shift_or_rotate(r + A_RR_13, a, 13, true)),
                                        ops.shift_or_rotate(r + A_RR_22, a, 22, true));
        const uint32_t a_and_b = ops.bitwise(r + A_AND_B, B_AND, a, b), a_and_c = ops.bitwise(r + A_AND_C, B_AND, a, c);
        const uint32_t b_and_c = ops.bitwise(r + B_AND_C, B_AND, b, c);
        const uint32_t maj = ops.bitwise(r + MAJ, B_XOR, ops.bitwise(r + MAJ_INTERMEDIATE, B_XOR, a_and_b, a_and_c), b_and_c);
        const uint32_t temp2 = ops.add(r + TEMP2, s0, maj);
        const uint32_t new_e = ops.add(r + D_ADD_TEMP1, d, temp1), new_a = ops.add(r + TEMP1_ADD_TEMP2, temp1, temp2);
        v[7] = g; v[6] = f; v[5] = ee; v[4] = new_e; v[3] = c; v[2] = b; v[1] = a; v[0] = new_a;
      } else {                           // the state is added to what was read and written back (:252-302)
        r[IS_FINALIZE] = 1;
        const uint32_t out = ops.add(r + FINALIZE_ADD, og[octet], v[octet]);
        const MemoryWriteRecord& m = e.h_write_records[octet];
        if (m.value != out || m.prev_value != og[octet]) throw std::runtime_error("tracegen: ShaCompressEvent write is not state + digest");
        memory_write_cols(m, r + MEM, &lk);
        r[MEM_ADDR] = fu32(e.h_ptr + 4 * octet);
        for (int i = 0; i < 8; i++) word(r + A + 4 * i, v[i]);
        word(r + FINALIZED_OPERAND, v[octet]);
        r[IS_LAST_ROW] = octet == 7;
      }
    }
  }
  for (size_t row = 80 * n_events; row < h; row++) {       // padding rows keep the octet counters and k going (trace.rs:58-80)
    F* r = t.data() + row * SHA_COMPRESS_WIDTH;
    const size_t step = (row - 80 * n_events) % 80, octet = step % 8, octet_num = step / 8;
    r[OCTET + octet] = 1; r[OCTET_NUM + octet_num] = 1;
    if (octet_num != 0 && octet_num != 9) word(r + K, SHA_COMPRESS_K[(octet_num - 1) * 8 + octet]);
    r[IS_LAST_ROW] = octet == 7 && octet_num == 9;
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

// ---- EdAddAssign precompile (syscall/precompiles/edwards/ed_add.rs): one Ed25519 point addition per row. Columns EdAddAssignCols :41-57 =
// is_real, shard, clk, p_ptr, q_ptr, sixteen MemoryWriteCols of p, sixteen MemoryReadCols of q, then eight field gadgets of 188 columns
// (result 32, carry 32, witness_low 62, witness_high 62): x3_numerator and y3_numerator (FieldInnerProductCols), x1_mul_y1, x2_mul_y2, f,
// d_mul_f (FieldOpCols, Mul), x3_ins, y3_ins (FieldDenCols). A gadget checks op = result + carry * p as a polynomial identity in x = 2^8:
// the difference vanishes at 256, its quotient by (x - 256), shifted by WITNESS_OFFSET = 2^14, is the witness
// (operations/field/util.rs:21-66). Padding rows hold the gadgets of the all-zero inputs (:130-147): the witness of zero is the offset.
struct EdAddEvent { uint32_t shard, clk, p_ptr, q_ptr; MemoryWriteRecord p_memory_records[16]; MemoryReadRecord q_memory_records[16]; };
static_assert(sizeof(EdAddEvent) == 4 * 180, "flattened EllipticCurveAddEvent is 180 words");
static const size_t ED_ADD_WIDTH = 1861;
static const uint8_t ED25519_MODULUS[32] = {237, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
                                            255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 127};
static const uint8_t ED25519_D[32] = {163, 120, 89, 19, 202, 77, 235, 117, 171, 216, 65, 65, 77, 10, 112, 0,
                                      152, 232, 121, 119, 121, 64, 199, 140, 115, 254, 111, 43, 238, 108, 3, 82};
// result, carry, witness_low, witness_high at r, from `lhs_minus_rhs` (the identity's polynomial before the carry term) and the two integers
// (`nw`: the witness length, 2 N - 2 for the fields and 2 N - 1 for U256Field, whose modulus 2^256 has N + 1 limbs: curves/src/uint256.rs:31-36)
static inline void field_gadget_cols(F* r, const bigfield::Poly& lhs_minus_rhs, const bigfield::Big& result, const bigfield::Big& carry,
                                     const bigfield::Big& p, int n_limbs, int64_t offset, std::vector<ByteLookup>* lk, int nw = -1) {
  using namespace bigfield;
  const int modulus_limbs = nw < 0 ? n_limbs : n_limbs + 1;
  if (nw < 0) nw = 2 * n_limbs - 2;
  Poly van = padd(lhs_minus_rhs, pmul(poly(carry, n_limbs), poly(p, modulus_limbs)), -1);
  van.resize(nw + 1, 0);
  std::vector<int64_t> w(nw, 0);          // van = w * (x - 256): w[k - 1] = van[k] + 256 * w[k] from the top down, and the constant term closes
  int64_t above = 0;
  for (int k = nw; k >= 1; k--) { w[k - 1] = van[k] + 256 * above; above = w[k - 1]; }
  if (van[0] + 256 * w[0] != 0) throw std::runtime_error("tracegen: field gadget identity does not hold");
  for (int i = 0; i < n_limbs; i++) { r[i] = limb(result, i); r[n_limbs + i] = limb(carry, i); }
  for (size_t i = n_limbs; i < std::max(result.size(), carry.size()); i++)
    if (limb(result, i) || limb(carry, i)) throw std::runtime_error("tracegen: field gadget result / carry does not fit its limbs");
  for (int i = 0; i < nw; i++) {
    const int64_t shifted = w[i] + offset;
    if (shifted < 0 || shifted >= 65536) throw std::runtime_error("tracegen: field gadget witness out of range");
    r[2 * n_limbs + i] = shifted & 0xff;
    r[2 * n_limbs + nw + i] = shifted >> 8;
  }
  if (lk) {
    auto ranges = [&](const F* c, int n) {
      for (int i = 0; i + 1 < n; i += 2) lk->push_back(ByteLookup{B_U8RANGE, (uint8_t)c[i], (uint8_t)c[i + 1]});
      if (n & 1) lk->push_back(ByteLookup{B_U8RANGE, (uint8_t)c[n - 1], 0});
    };
    ranges(r, n_limbs); ranges(r + n_limbs, n_limbs); ranges(r + 2 * n_limbs, nw); ranges(r + 2 * n_limbs + nw, nw);
  }
}
struct FieldGadgets {
  bigfield::Big p;
  int n;
  int64_t offset;
  std::vector<ByteLookup>* lk;
  // FieldOpCols::populate, Mul (field_op.rs:97-152): a * b = result + carry * p
  bigfield::Big mul(F* r, const bigfield::Big& a, const bigfield::Big& b) const {
    using namespace bigfield;
    Big q, res;
    divmod(bigfield::mul(a, b), p, q, res);
    field_gadget_cols(r, padd(pmul(poly(a, n), poly(b, n)), poly(res, n), -1), res, q, p, n, offset, lk);
    return res;
  }
  // FieldInnerProductCols::populate (field_inner_product.rs:27-79): a0 * b0 + a1 * b1 = result + carry * p
  bigfield::Big inner_product(F* r, const bigfield::Big& a0, const bigfield::Big& b0, const bigfield::Big& a1, const bigfield::Big& b1) const {
    using namespace bigfield;
    Big q, res;
    divmod(add(bigfield::mul(a0, b0), bigfield::mul(a1, b1)), p, q, res);
    field_gadget_cols(r, padd(padd(pmul(poly(a0, n), poly(b0, n)), pmul(poly(a1, n), poly(b1, n))), poly(res, n), -1), res, q, p, n, offset, lk);
    return res;
  }
  // FieldDenCols::populate (field_den.rs:27-81): result = a / (1 + b) (sign) or a / (1 - b)
  bigfield::Big den(F* r, const bigfield::Big& a, const bigfield::Big& b, bool sign) const {
    using namespace bigfield;
    const Big denominator = mod(add(sign ? b : sub(p, mod(b, p)), from_u64(1)), p);
    const Big res = is_zero(a) ? Big() : mod(bigfield::mul(a, inv_mod(denominator, p)), p);
    const Big lhs = add(bigfield::mul(b, res), sign ? res : a), rhs = sign ? a : res;
    Big q, rem;
    divmod(sub(lhs, rhs), p, q, rem);
    if (!is_zero(rem)) throw std::runtime_error("tracegen: FieldDen identity");
    Poly lhs_p = padd(pmul(poly(b, n), poly(res, n)), poly(sign ? res : a, n));
    field_gadget_cols(r, padd(lhs_p, poly(rhs, n), -1), res, q, p, n, offset, lk);
    return res;
  }
};
static inline std::vector<F> generate_ed_add(const EdAddEvent* events, size_t n_events, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  using bigfield::Big;
  enum { IS_REAL = 0, SHARD = 1, CLK = 2, P_PTR = 3, Q_PTR = 4, P_ACCESS = 5, Q_ACCESS = 5 + 16 * 13, GADGETS = 5 + 16 * 13 + 16 * 9, G = 188 };
  static_assert(GADGETS + 8 * G == 1861, "layout");
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * ED_ADD_WIDTH, 0);
  std::vector<ByteLookup> lk;
  const Big d = bigfield::from_bytes(ED25519_D, 32);
  auto fill = [&](F* r, const Big& x1, const Big& y1, const Big& x2, const Big& y2, std::vector<ByteLookup>* sink) {   // populate_field_ops :68-95
    const FieldGadgets g{bigfield::from_bytes(ED25519_MODULUS, 32), 32, 1 << 14, sink};
    const Big x3n = g.inner_product(r + GADGETS, x1, y2, x2, y1);
    const Big y3n = g.inner_product(r + GADGETS + G, y1, y2, x1, x2);
    const Big x1y1 = g.mul(r + GADGETS + 2 * G, x1, y1), x2y2 = g.mul(r + GADGETS + 3 * G, x2, y2);
    const Big f = g.mul(r + GADGETS + 4 * G, x1y1, x2y2);
    const Big df = g.mul(r + GADGETS + 5 * G, f, d);
    g.den(r + GADGETS + 6 * G, x3n, df, true);
    g.den(r + GADGETS + 7 * G, y3n, df, false);
  };
  std::vector<F> padding(ED_ADD_WIDTH, 0);
  fill(padding.data(), Big(), Big(), Big(), Big(), nullptr);
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * ED_ADD_WIDTH;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const EdAddEvent& e = events[i];
    uint32_t p[16], q[16];
    for (int k = 0; k < 16; k++) { p[k] = e.p_memory_records[k].prev_value; q[k] = e.q_memory_records[k].value; }
    r[IS_REAL] = 1; r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[P_PTR] = fu32(e.p_ptr); r[Q_PTR] = fu32(e.q_ptr);
    fill(r, bigfield::from_words(p, 8), bigfield::from_words(p + 8, 8), bigfield::from_words(q, 8), bigfield::from_words(q + 8, 8), &lk);
    for (int k = 0; k < 16; k++) {
      const MemoryReadRecord& m = e.q_memory_records[k];
      memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + Q_ACCESS + 9 * k, &lk);
    }
    for (int k = 0; k < 16; k++) {
      memory_write_cols(e.p_memory_records[k], r + P_ACCESS + 13 * k, &lk);
      for (int c = 0; c < 4; c++)      // the words written are the sum's limbs (ed_add.rs:299-307)
        if (r[P_ACCESS + 13 * k + 4 + c] != r[GADGETS + (k < 8 ? 6 : 7) * G + 4 * (k % 8) + c]) throw std::runtime_error("tracegen: EdAdd event does not write p + q");
    }
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

// ---- EdDecompress precompile (syscall/precompiles/edwards/ed_decompress.rs): x from y and a sign bit. Columns EdDecompressCols :39-57 = is_real,
// shard, clk, ptr, sign, eight MemoryWriteCols of x, eight MemoryReadCols of y, y_range (FieldLtCols: 32 byte flags + the two compared bytes),
// yy, u, dyy, v, u_div_v (FieldOpCols), x (FieldSqrtCols = a FieldOpCols whose result columns hold the root, a FieldLtCols, the root's low
// bit), neg_x (FieldOpCols): 1566 columns. Padding rows hold the field operations of y = 0 (:239-249), whose root is sqrt(-1).
struct EdDecompressEvent { uint32_t shard, clk, ptr, sign; MemoryWriteRecord x_memory_records[8]; MemoryReadRecord y_memory_records[8]; };
static_assert(sizeof(EdDecompressEvent) == 4 * 92, "flattened EdDecompressEvent is 92 words");
static const size_t ED_DECOMPRESS_WIDTH = 1566;
enum FieldOpKind { FOP_ADD, FOP_SUB, FOP_MUL, FOP_DIV };
// FieldOpCols::populate_with_modulus (field_op.rs:154-224): Sub and Div fill carry and witness from the reversed identity result op' b = a
static inline bigfield::Big field_op_cols(const FieldGadgets& g, F* r, const bigfield::Big& a, const bigfield::Big& b, FieldOpKind op) {
  using namespace bigfield;
  const int n = g.n;
  Big res, lhs_value, rhs = a;
  Poly lhs_poly;
  switch (op) {
    case FOP_ADD: res = mod(add(a, b), g.p); break;
    case FOP_MUL: res = mod(bigfield::mul(a, b), g.p); break;
    case FOP_SUB: res = mod(sub(add(g.p, a), mod(b, g.p)), g.p); break;
    case FOP_DIV:
      if (is_zero(b) && !is_zero(a)) throw std::runtime_error("tracegen: division by zero is allowed only when dividing zero");
      res = is_zero(a) ? Big() : mod(bigfield::mul(a, inv_mod(b, g.p)), g.p);
      break;
  }
  if (op == FOP_ADD || op == FOP_MUL) {           // a op b = res + carry p
    lhs_value = op == FOP_ADD ? add(a, b) : bigfield::mul(a, b);
    lhs_poly = op == FOP_ADD ? padd(poly(a, n), poly(b, n)) : pmul(poly(a, n), poly(b, n));
    rhs = res;
  } else {                                        // res op' b = a + carry p
    lhs_value = op == FOP_SUB ? add(res, b) : bigfield::mul(res, b);
    lhs_poly = op == FOP_SUB ? padd(poly(res, n), poly(b, n)) : pmul(poly(res, n), poly(b, n));
  }
  Big q, rem;
  divmod(sub(lhs_value, rhs), g.p, q, rem);
  if (!is_zero(rem)) throw std::runtime_error("tracegen: field operation identity");
  field_gadget_cols(r, padd(lhs_poly, poly(rhs, n), -1), res, q, g.p, n, g.offset, g.lk);
  return res;
}
// FieldLtCols::populate (range.rs:27-60): the flag of the most significant byte where lhs < rhs, and the two bytes
static inline void field_lt_cols(F* r, const bigfield::Big& lhs, const bigfield::Big& rhs, int n, std::vector<ByteLookup>* lk) {
  if (bigfield::cmp(lhs, rhs) >= 0) throw std::runtime_error("tracegen: field element is not below the modulus");
  for (int i = n - 1; i >= 0; i--) {
    const uint32_t a = bigfield::limb(lhs, i), b = bigfield::limb(rhs, i);
    if (a < b) {
      r[i] = 1; r[n] = a; r[n + 1] = b;
      if (lk) lk->push_back(ByteLookup{B_LTU_OP, (uint8_t)a, (uint8_t)b});
      return;
    }
  }
}
static inline std::vector<F> generate_ed_decompress(const EdDecompressEvent* events, size_t n_events, int fixed_log2_rows, size_t* height,
                                                    uint64_t* byte_counts) {
  using namespace bigfield;
  enum { IS_REAL = 0, SHARD = 1, CLK = 2, PTR = 3, SIGN = 4, X_ACCESS = 5, Y_ACCESS = 109, Y_RANGE = 181, YY = 215, U = 403, DYY = 591, V = 779, U_DIV_V = 967,
         X_MULT = 1155, X_RANGE = 1343, X_LSB = 1377, NEG_X = 1378 };
  static_assert(NEG_X + 188 == 1566, "layout");
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * ED_DECOMPRESS_WIDTH, 0);
  std::vector<ByteLookup> lk;
  const Big p = from_bytes(ED25519_MODULUS, 32), d = from_bytes(ED25519_D, 32), one = from_u64(1);
  // ed25519_sqrt (curves/src/edwards/ed25519.rs:75-113): a^((p + 3) / 8), times sqrt(-1) when that squares to -a; the even root
  auto sqrt = [&](const Big& a) {
    Big e, rem;
    divmod(add(p, from_u64(3)), from_u64(8), e, rem);
    Big beta = pow_mod(a, e, p);
    const Big sq = mod(bigfield::mul(beta, beta), p), neg_a = mod(sub(p, a), p);
    if (cmp(sq, neg_a) == 0 && cmp(sq, a) != 0) {
      const Big e4 = [&] { Big q2, r2; divmod(sub(p, one), from_u64(4), q2, r2); return q2; }();
      beta = mod(bigfield::mul(beta, pow_mod(from_u64(2), e4, p)), p);        // 2^((p - 1) / 4) is a square root of -1
    } else if (cmp(sq, a) != 0) {
      throw std::runtime_error("tracegen: EdDecompress: not a square");
    }
    if (bit(beta, 0)) beta = mod(sub(p, beta), p);
    return beta;
  };
  auto fill = [&](F* r, const Big& y, std::vector<ByteLookup>* sink) {      // populate_field_ops :85-101; returns the even root
    const FieldGadgets g{p, 32, 1 << 14, sink};
    field_lt_cols(r + Y_RANGE, y, p, 32, sink);
    const Big yy = field_op_cols(g, r + YY, y, y, FOP_MUL);
    const Big u = field_op_cols(g, r + U, yy, one, FOP_SUB);
    const Big dyy = field_op_cols(g, r + DYY, d, yy, FOP_MUL);
    const Big v = field_op_cols(g, r + V, one, dyy, FOP_ADD);
    const Big u_div_v = field_op_cols(g, r + U_DIV_V, u, v, FOP_DIV);
    const Big x = sqrt(u_div_v);                                           // FieldSqrtCols::populate (field_sqrt.rs:34-85)
    if (cmp(field_op_cols(g, r + X_MULT, x, x, FOP_MUL), u_div_v) != 0) throw std::runtime_error("tracegen: EdDecompress: root");
    for (int i = 0; i < 32; i++) r[X_MULT + i] = limb(x, i);               // the result columns are overwritten with the root
    field_lt_cols(r + X_RANGE, x, p, 32, sink);
    r[X_LSB] = limb(x, 0) & 1;
    if (sink) {
      sink->push_back(ByteLookup{B_AND_OP, (uint8_t)limb(x, 0), 1});
      for (int i = 0; i < 32; i += 2) sink->push_back(ByteLookup{B_U8RANGE, (uint8_t)limb(x, i), (uint8_t)limb(x, i + 1)});
    }
    field_op_cols(g, r + NEG_X, Big(), x, FOP_SUB);
    return x;
  };
  std::vector<F> padding(ED_DECOMPRESS_WIDTH, 0);
  fill(padding.data(), Big(), nullptr);
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * ED_DECOMPRESS_WIDTH;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const EdDecompressEvent& e = events[i];
    if (e.sign > 1) throw std::runtime_error("tracegen: EdDecompress sign bit");
    r[IS_REAL] = 1; r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[PTR] = fu32(e.ptr); r[SIGN] = e.sign;
    uint32_t yw[8];
    for (int k = 0; k < 8; k++) {
      memory_write_cols(e.x_memory_records[k], r + X_ACCESS + 13 * k, &lk);
      const MemoryReadRecord& m = e.y_memory_records[k];
      memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + Y_ACCESS + 9 * k, &lk);
      yw[k] = m.value;
    }
    const Big x = fill(r, from_words(yw, 8), &lk);
    for (int k = 0; k < 32; k++)        // what is written is the root, or its negative when the sign bit is set (:170-176)
      if (r[X_ACCESS + 13 * (k / 4) + 4 + k % 4] != (e.sign ? r[NEG_X + k] : r[X_MULT + k])) throw std::runtime_error("tracegen: EdDecompress event does not write x");
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

// ---- Short-Weierstrass precompiles (syscall/precompiles/weierstrass/weierstrass_add.rs:43-62, :75-129; weierstrass_double.rs:43-62, :75-150) for
// any of the four curves: the curve's base field comes in as its modulus bytes, limb count and witness offset, a doubling also needs `a`.
// Events are the flattened EllipticCurveAddEvent (shard, clk, p_ptr, q_ptr, W write records of p, W read records of q) or
// EllipticCurveDoubleEvent (shard, clk, p_ptr, W write records of p), W = n_limbs / 2 words per point. Padding rows of an addition: the
// operations of the zero inputs (0 / 0 = 0 is the one division by zero FieldOpCols allows); of a doubling: see below.
static inline std::vector<F> generate_weierstrass(const uint32_t* events, size_t n_events, bool dbl, int n_limbs, const uint8_t* modulus_bytes,
                                                  const uint8_t* a_bytes, int64_t offset, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  using namespace bigfield;
  const int W = n_limbs / 2, G = 6 * n_limbs - 4;
  const int P_ACCESS = dbl ? 4 : 5, Q_ACCESS = P_ACCESS + 13 * W, GADGETS = P_ACCESS + 13 * W + (dbl ? 0 : 9 * W);
  const size_t width = GADGETS + (dbl ? 11 : 9) * G, ev_words = dbl ? 3 + 6 * W : 4 + 11 * W;
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * width, 0);
  std::vector<ByteLookup> lk;
  const Big p = from_bytes(modulus_bytes, n_limbs), a = from_bytes(a_bytes, n_limbs);
  auto fill = [&](F* r, const Big& px, const Big& py, const Big& qx, const Big& qy, std::vector<ByteLookup>* sink, Big& x3, Big& y3) {
    const FieldGadgets g{p, n_limbs, offset, sink};
    auto col = [&](int k) { return r + GADGETS + G * k; };
    Big slope, slope_sq, sum_x;
    if (!dbl) {          // gadget order: slope_denominator, slope_numerator, slope, slope_squared, p_x_plus_q_x, x3_ins, p_x_minus_x, y3_ins, slope_times_p_x_minus_x
      const Big num = field_op_cols(g, col(1), qy, py, FOP_SUB), den = field_op_cols(g, col(0), qx, px, FOP_SUB);
      slope = field_op_cols(g, col(2), num, den, FOP_DIV);
      slope_sq = field_op_cols(g, col(3), slope, slope, FOP_MUL);
      sum_x = field_op_cols(g, col(4), px, qx, FOP_ADD);
      x3 = field_op_cols(g, col(5), slope_sq, sum_x, FOP_SUB);
      const Big dx = field_op_cols(g, col(6), px, x3, FOP_SUB);
      const Big prod = field_op_cols(g, col(8), slope, dx, FOP_MUL);
      y3 = field_op_cols(g, col(7), prod, py, FOP_SUB);
    } else {             // slope_denominator, slope_numerator, slope, p_x_squared, p_x_squared_times_3, slope_squared, p_x_plus_p_x, x3_ins, p_x_minus_x, y3_ins, slope_times_p_x_minus_x
      const Big sq = field_op_cols(g, col(3), px, px, FOP_MUL);
      const Big sq3 = field_op_cols(g, col(4), sq, from_u64(3), FOP_MUL);
      const Big num = field_op_cols(g, col(1), a, sq3, FOP_ADD);
      const Big den = field_op_cols(g, col(0), from_u64(2), py, FOP_MUL);
      slope = field_op_cols(g, col(2), num, den, FOP_DIV);
      slope_sq = field_op_cols(g, col(5), slope, slope, FOP_MUL);
      sum_x = field_op_cols(g, col(6), px, px, FOP_ADD);
      x3 = field_op_cols(g, col(7), slope_sq, sum_x, FOP_SUB);
      const Big dx = field_op_cols(g, col(8), px, x3, FOP_SUB);
      const Big prod = field_op_cols(g, col(10), slope, dx, FOP_MUL);
      y3 = field_op_cols(g, col(9), prod, py, FOP_SUB);
    }
  };
  std::vector<F> padding(width, 0);
  Big x3, y3;
  if (dbl) {      // the doubling's padding row is the point (0, 1) — a / 0 would not be allowed — with a dummy write record on the first word of y
                  // (weierstrass_double.rs:225-239: value 1, shard 0, timestamp 1, previous value 1 at (0, 0))
    fill(padding.data(), Big(), from_u64(1), Big(), Big(), nullptr, x3, y3);
    const MemoryWriteRecord dummy{1, 0, 1, 1, 0, 0};
    memory_write_cols(dummy, padding.data() + P_ACCESS + 13 * (W / 2), nullptr);
  } else {
    fill(padding.data(), Big(), Big(), Big(), Big(), nullptr, x3, y3);
  }
  std::vector<uint32_t> pw(W), qw(W);
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * width;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const uint32_t* e = events + i * ev_words;
    const MemoryWriteRecord* prec = (const MemoryWriteRecord*)(e + (dbl ? 3 : 4));
    const MemoryReadRecord* qrec = (const MemoryReadRecord*)(e + 4 + 6 * W);
    r[0] = 1; r[1] = fu32(e[0]); r[2] = fu32(e[1]); r[3] = fu32(e[2]);
    if (!dbl) r[4] = fu32(e[3]);
    for (int k = 0; k < W; k++) { pw[k] = prec[k].prev_value; qw[k] = dbl ? 0u : qrec[k].value; }
    const Big px = from_words(pw.data(), W / 2), py = from_words(pw.data() + W / 2, W / 2);
    if (cmp(px, p) >= 0 || cmp(py, p) >= 0) throw std::runtime_error("tracegen: Weierstrass point coordinate is not below the modulus");
    fill(r, px, py, from_words(qw.data(), W / 2), from_words(qw.data() + W / 2, W / 2), &lk, x3, y3);
    if (!dbl)
      for (int k = 0; k < W; k++) {
        const MemoryReadRecord& m = qrec[k];
        memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + Q_ACCESS + 9 * k, &lk);
      }
    for (int k = 0; k < W; k++) {
      memory_write_cols(prec[k], r + P_ACCESS + 13 * k, &lk);
      const Big& coord = k < W / 2 ? x3 : y3;
      for (int c = 0; c < 4; c++)
        if (r[P_ACCESS + 13 * k + 4 + c] != limb(coord, 4 * (k % (W / 2)) + c)) throw std::runtime_error("tracegen: Weierstrass event does not write the result point");
    }
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

// ---- <Curve>Decompress (syscall/precompiles/weierstrass/weierstrass_decompress.rs:52-79, :121-142, :163-285): y = sqrt(x^3 + a x + b) from x and a
// sign bit. Events: shard, clk, ptr, sign_bit, W read records of x, W write records of y (W = N / 4). Columns: is_real, shard, clk, ptr,
// sign_bit, x_access (W x 9), y_access (W x 13), range_x (FieldLtCols), x_2, x_3 (FieldOpCols), ax_plus_b (FieldInnerProductCols),
// x_3_plus_b_plus_ax, y (FieldSqrtCols), neg_y; with the lexicographic rule (Bls12381) also comparison_lt_cols, neg_y_range_check and three
// flags. The root is a^((p + 1) / 4) — what k256 / p256 / amcl compute for these p = 3 (mod 4); those crates are not in the tree (unpinned).
// Padding rows: the field operations of the generator's x, which also sits in the value columns of x_access (:254-271).
static inline std::vector<F> generate_weierstrass_decompress(const uint32_t* events, size_t n_events, int n_limbs, const uint8_t* modulus_bytes,
                                                             const uint8_t* a_bytes, const uint8_t* b_bytes, const uint8_t* generator_x_bytes, int64_t offset,
                                                             bool lexicographic, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  using namespace bigfield;
  const int N = n_limbs, W = N / 4, G = 6 * N - 4;
  const int X_ACCESS = 5, Y_ACCESS = X_ACCESS + 9 * W, RANGE_X = Y_ACCESS + 13 * W, X_2 = RANGE_X + N + 2, X_3 = X_2 + G, AX_PLUS_B = X_2 + 2 * G,
            X_3_PLUS = X_2 + 3 * G, Y_MULT = X_2 + 4 * G, Y_RANGE = Y_MULT + G, Y_LSB = Y_RANGE + N + 2, NEG_Y = Y_LSB + 1, CHOICE = NEG_Y + G;
  const size_t width = CHOICE + (lexicographic ? 2 * (N + 2) + 3 : 0), ev_words = 4 + 11 * W;
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * width, 0);
  std::vector<ByteLookup> lk;
  const Big p = from_bytes(modulus_bytes, N), a = from_bytes(a_bytes, N), b = from_bytes(b_bytes, N), one = from_u64(1);
  Big sqrt_exp, rem;
  divmod(add(p, one), from_u64(4), sqrt_exp, rem);
  if (!is_zero(rem)) throw std::runtime_error("tracegen: decompress needs p = 3 (mod 4)");
  auto fill = [&](F* r, const Big& x, std::vector<ByteLookup>* sink) {          // populate_field_ops :121-142; returns the root
    const FieldGadgets g{p, N, offset, sink};
    field_lt_cols(r + RANGE_X, x, p, N, sink);
    const Big x2 = field_op_cols(g, r + X_2, x, x, FOP_MUL);
    const Big x3 = field_op_cols(g, r + X_3, x2, x, FOP_MUL);
    const Big ax_b = g.inner_product(r + AX_PLUS_B, a, x, b, one);
    const Big rhs = field_op_cols(g, r + X_3_PLUS, x3, ax_b, FOP_ADD);
    const Big y = pow_mod(rhs, sqrt_exp, p);                                    // FieldSqrtCols::populate (field_sqrt.rs:34-85)
    if (cmp(field_op_cols(g, r + Y_MULT, y, y, FOP_MUL), rhs) != 0) throw std::runtime_error("tracegen: decompress: x is not on the curve");
    for (int i = 0; i < N; i++) r[Y_MULT + i] = limb(y, i);
    field_lt_cols(r + Y_RANGE, y, p, N, sink);
    r[Y_LSB] = limb(y, 0) & 1;
    if (sink) {
      sink->push_back(ByteLookup{B_AND_OP, (uint8_t)limb(y, 0), 1});
      for (int i = 0; i < N; i += 2) sink->push_back(ByteLookup{B_U8RANGE, (uint8_t)limb(y, i), (uint8_t)limb(y, i + 1)});
    }
    field_op_cols(g, r + NEG_Y, Big(), y, FOP_SUB);
    return y;
  };
  std::vector<F> padding(width, 0);
  {
    const Big gx = from_bytes(generator_x_bytes, N);
    for (int i = 0; i < N; i++) padding[X_ACCESS + 9 * (i / 4) + i % 4] = limb(gx, i);
    fill(padding.data(), gx, nullptr);
  }
  std::vector<uint32_t> xw(W), yw(W);
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * width;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const uint32_t* e = events + i * ev_words;
    const MemoryReadRecord* xrec = (const MemoryReadRecord*)(e + 4);
    const MemoryWriteRecord* yrec = (const MemoryWriteRecord*)(e + 4 + 5 * W);
    if (e[3] > 1) throw std::runtime_error("tracegen: decompress sign bit");
    r[0] = 1; r[1] = fu32(e[0]); r[2] = fu32(e[1]); r[3] = fu32(e[2]); r[4] = e[3];
    for (int k = 0; k < W; k++) {
      const MemoryReadRecord& m = xrec[k];
      memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + X_ACCESS + 9 * k, &lk);
      memory_write_cols(yrec[k], r + Y_ACCESS + 13 * k, &lk);
      xw[k] = m.value; yw[k] = yrec[k].value;
    }
    const Big x = from_words(xw.data(), W), decompressed = from_words(yw.data(), W);
    if (cmp(x, p) >= 0) throw std::runtime_error("tracegen: decompress: x is not below the modulus");
    const Big y = fill(r, x, &lk);
    const Big neg = mod(sub(p, y), p);
    const bool wrote_root = cmp(decompressed, y) == 0;
    if (!wrote_root && cmp(decompressed, neg) != 0) throw std::runtime_error("tracegen: decompress event does not write a root");
    if (!lexicographic) {
      if ((limb(decompressed, 0) & 1) != e[3]) throw std::runtime_error("tracegen: decompress event does not write the root the sign bit asks for");
    } else {        // LexicographicChoiceCols :196-246: comparison_lt_cols, neg_y_range_check, is_y_eq_sqrt_y_result, when_sqrt_y_res_is_lt, when_neg_y_res_is_lt
      const Big other = mod(sub(p, decompressed), p);                           // `neg_y` of :201
      const bool larger = cmp(other, decompressed) < 0;
      if (larger != (e[3] != 0) || cmp(other, decompressed) == 0) throw std::runtime_error("tracegen: decompress event does not write the root the sign bit asks for");
      const int CMP = CHOICE, NEG_RANGE = CHOICE + N + 2, FLAGS = CHOICE + 2 * (N + 2);
      r[FLAGS] = wrote_root ? 1 : 0;
      field_lt_cols(r + NEG_RANGE, wrote_root ? other : decompressed, p, N, &lk);      // either way: the root's negative
      if (e[3]) {
        r[FLAGS + 1] = !wrote_root; r[FLAGS + 2] = wrote_root;
        field_lt_cols(r + CMP, other, decompressed, N, &lk);
      } else {
        r[FLAGS + 1] = wrote_root; r[FLAGS + 2] = !wrote_root;
        field_lt_cols(r + CMP, decompressed, other, N, &lk);
      }
    }
  }
  if (byte_counts)
    for (const ByteLookup& bl : lk) byte_counts[((size_t)bl.b * 256 + bl.c) * NUM_BYTE_OPS + bl.op]++;
  *height = h;
  return t;
}

// ---- Uint256MulMod (syscall/precompiles/uint256/air.rs:57-91, :104-203): x <- x * y mod m, m = 0 standing for 2^256. Columns: shard, clk, x_ptr, y_ptr,
// eight MemoryWriteCols of x, eight MemoryReadCols of y, eight of the modulus, modulus_is_zero (IsZeroOperation of the sum of the modulus' bytes),
// modulus_is_not_zero, output (FieldOpCols over U256Field: 32 + 32 + 63 + 63), output_range_check (FieldLtCols, filled only when there is a
// modulus), is_real: 480 columns. Padding rows: the product of zeros (witness_high = 2^14 >> 8).
struct Uint256MulEvent { uint32_t shard, clk, x_ptr, y_ptr; MemoryWriteRecord x_memory_records[8]; MemoryReadRecord y_memory_records[8], modulus_memory_records[8]; };
static_assert(sizeof(Uint256MulEvent) == 4 * 132, "flattened Uint256MulEvent is 132 words");
static const size_t UINT256_MUL_WIDTH = 480;
static inline std::vector<F> generate_uint256_mul(const Uint256MulEvent* events, size_t n_events, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  using namespace bigfield;
  enum { SHARD = 0, CLK = 1, X_PTR = 2, Y_PTR = 3, X_MEM = 4, Y_MEM = 108, M_MEM = 180, IS_ZERO = 252, NOT_ZERO = 254, OUTPUT = 255, RANGE = 445, IS_REAL = 479, N = 32, NW = 63 };
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * UINT256_MUL_WIDTH, 0);
  std::vector<ByteLookup> lk;
  Big two_256(33, 0);
  two_256[32] = 1;
  auto product = [&](F* r, const Big& x, const Big& y, const Big& modulus, std::vector<ByteLookup>* sink) {      // populate_with_modulus, Mul (field_op.rs:154-224)
    Big q, res;
    divmod(bigfield::mul(x, y), modulus, q, res);
    field_gadget_cols(r + OUTPUT, padd(pmul(poly(x, N), poly(y, N)), poly(res, N), -1), res, q, modulus, N, 1 << 14, sink, NW);
    return res;
  };
  std::vector<F> padding(UINT256_MUL_WIDTH, 0);
  product(padding.data(), Big(), Big(), two_256, nullptr);
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * UINT256_MUL_WIDTH;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const Uint256MulEvent& e = events[i];
    r[IS_REAL] = 1; r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[X_PTR] = fu32(e.x_ptr); r[Y_PTR] = fu32(e.y_ptr);
    uint32_t xw[8], yw[8], mw[8], byte_sum = 0;
    for (int k = 0; k < 8; k++) {
      memory_write_cols(e.x_memory_records[k], r + X_MEM + 13 * k, &lk);
      const MemoryReadRecord& y = e.y_memory_records[k];
      memory_access_cols(y.value, y.shard, y.timestamp, y.prev_shard, y.prev_timestamp, r + Y_MEM + 9 * k, &lk);
      const MemoryReadRecord& m = e.modulus_memory_records[k];
      memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + M_MEM + 9 * k, &lk);
      xw[k] = e.x_memory_records[k].prev_value; yw[k] = y.value; mw[k] = m.value;
      for (int c = 0; c < 4; c++) byte_sum += (m.value >> (8 * c)) & 0xff;
    }
    is_zero_cols(byte_sum, r + IS_ZERO);
    const Big modulus = from_words(mw, 8);
    const bool no_modulus = is_zero(modulus);
    const Big res = product(r, from_words(xw, 8), from_words(yw, 8), no_modulus ? two_256 : modulus, &lk);
    r[NOT_ZERO] = no_modulus ? 0 : 1;
    if (!no_modulus) field_lt_cols(r + RANGE, res, modulus, N, &lk);
    for (int k = 0; k < N; k++)
      if (r[X_MEM + 13 * (k / 4) + 4 + k % 4] != r[OUTPUT + k]) throw std::runtime_error("tracegen: Uint256Mul event does not write x * y mod m");
  }
  if (byte_counts)
    for (const ByteLookup& bl : lk) byte_counts[((size_t)bl.b * 256 + bl.c) * NUM_BYTE_OPS + bl.op]++;
  *height = h;
  return t;
}

// ---- U256XU2048Mul (syscall/precompiles/u256x2048_mul/air.rs:52-87, :101-229): a (256 bits) times b (2048 bits); eight FieldOpCols over U256Field chained
// through their carries (populate_mul_and_carry, field_op.rs:47-95): a * b_i + carry_{i-1} = result_i + carry_i 2^256. Columns: shard, clk, a_ptr,
// b_ptr, lo_ptr, hi_ptr, the reads of registers $a2 / $a3 (lo_ptr_memory, hi_ptr_memory), 8 + 64 MemoryReadCols of a, b, 64 + 8 MemoryWriteCols
// of lo, hi, the eight gadgets, is_real: 3129. Padding rows: products of zeros.
struct U256x2048MulEvent {
  uint32_t shard, clk, a_ptr, b_ptr, lo_ptr, hi_ptr;
  MemoryReadRecord lo_ptr_memory, hi_ptr_memory, a_memory_records[8], b_memory_records[64];
  MemoryWriteRecord lo_memory_records[64], hi_memory_records[8];
};
static_assert(sizeof(U256x2048MulEvent) == 4 * 808, "flattened U256xU2048MulEvent is 808 words");
static const size_t U256X2048_MUL_WIDTH = 3129;
static inline std::vector<F> generate_u256x2048_mul(const U256x2048MulEvent* events, size_t n_events, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  using namespace bigfield;
  enum { SHARD = 0, CLK = 1, A_PTR = 2, B_PTR = 3, LO_PTR = 4, HI_PTR = 5, LO_PTR_MEM = 6, HI_PTR_MEM = 15, A_MEM = 24, B_MEM = 96, LO_MEM = 672, HI_MEM = 1504,
         GADGETS = 1608, IS_REAL = 3128, N = 32, NW = 63, G = 190 };
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * U256X2048_MUL_WIDTH, 0);
  std::vector<ByteLookup> lk;
  Big two_256(33, 0);
  two_256[32] = 1;
  auto gadgets = [&](F* r, const Big& a, const uint32_t* b_words, std::vector<ByteLookup>* sink) {
    Big carry;
    for (int g = 0; g < 8; g++) {
      const Big b = b_words ? from_words(b_words + 8 * g, 8) : Big();
      Big q, res;
      divmod(add(bigfield::mul(a, b), carry), two_256, q, res);
      field_gadget_cols(r + GADGETS + G * g, padd(padd(pmul(poly(a, N), poly(b, N)), poly(carry, N)), poly(res, N), -1), res, q, two_256, N, 1 << 14, sink, NW);
      carry = q;
    }
  };
  std::vector<F> padding(U256X2048_MUL_WIDTH, 0);
  gadgets(padding.data(), Big(), nullptr, nullptr);
  auto read = [&](const MemoryReadRecord& m, F* r) { memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r, &lk); };
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * U256X2048_MUL_WIDTH;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const U256x2048MulEvent& e = events[i];
    r[IS_REAL] = 1; r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[A_PTR] = fu32(e.a_ptr); r[B_PTR] = fu32(e.b_ptr); r[LO_PTR] = fu32(e.lo_ptr); r[HI_PTR] = fu32(e.hi_ptr);
    if (e.lo_ptr_memory.value != e.lo_ptr || e.hi_ptr_memory.value != e.hi_ptr) throw std::runtime_error("tracegen: U256x2048Mul: lo_ptr / hi_ptr are not what the registers hold");
    read(e.lo_ptr_memory, r + LO_PTR_MEM); read(e.hi_ptr_memory, r + HI_PTR_MEM);
    uint32_t aw[8], bw[64];
    for (int k = 0; k < 8; k++) { read(e.a_memory_records[k], r + A_MEM + 9 * k); aw[k] = e.a_memory_records[k].value; }
    for (int k = 0; k < 64; k++) { read(e.b_memory_records[k], r + B_MEM + 9 * k); bw[k] = e.b_memory_records[k].value; }
    for (int k = 0; k < 64; k++) memory_write_cols(e.lo_memory_records[k], r + LO_MEM + 13 * k, &lk);
    for (int k = 0; k < 8; k++) memory_write_cols(e.hi_memory_records[k], r + HI_MEM + 13 * k, &lk);
    gadgets(r, from_words(aw, 8), bw, &lk);
    for (int k = 0; k < 256; k++)
      if (r[LO_MEM + 13 * (k / 4) + 4 + k % 4] != r[GADGETS + G * (k / 32) + k % 32]) throw std::runtime_error("tracegen: U256x2048Mul event does not write the product");
    for (int k = 0; k < 32; k++)
      if (r[HI_MEM + 13 * (k / 4) + 4 + k % 4] != r[GADGETS + G * 7 + N + k]) throw std::runtime_error("tracegen: U256x2048Mul event does not write the product");
  }
  if (byte_counts)
    for (const ByteLookup& bl : lk) byte_counts[((size_t)bl.b * 256 + bl.c) * NUM_BYTE_OPS + bl.op]++;
  *height = h;
  return t;
}

// ---- BooleanCircuitGarble (syscall/precompiles/boolean_circuit_garble/columns.rs:10-35, trace.rs:100-223): a call is a header row (the reads of the
// gate count and of delta in gates_input_mem[0..5]) and one row per gate (seventeen reads: type, h0, h1, label_b, expected; aux1 = h0 ^ h1,
// aux2 = aux1 ^ label_b, aux3 = aux2 ^ delta; is_equal_words of aux2 (AND gate, type 0) or aux3 (OR gate, type 7) with the expected words;
// checks = the conjunctions of those, checks[3] also of all the gates before; the last gate's row holds the write of the result). Input: one
// GarbleRow per row, cut from the BooleanCircuitGarbleEvent. The reference pads this table to the next power of two with no floor of 16
// (trace.rs:84-85) when no shape fixes its size; here the floor is 16 as for every other table, and fixed_log2_rows gives any size.
struct GarbleRow { uint32_t shard, clk, input_address, output_address, is_gate, gate_id, gates_num, pre_check, delta[4]; MemoryReadRecord reads[17]; MemoryWriteRecord write; };
static_assert(sizeof(GarbleRow) == 4 * 103, "a BooleanCircuitGarble row record is 103 words");
static const size_t GARBLE_WIDTH = 292;
static inline bool garble_gate_ok(const GarbleRow& g) {
  bool ok = true;
  for (int i = 0; i < 4; i++) {
    const uint32_t v = g.reads[1 + i].value ^ g.reads[5 + i].value ^ g.reads[9 + i].value ^ (g.reads[0].value ? g.delta[i] : 0u);
    ok = ok && v == g.reads[13 + i].value;
  }
  return ok;
}
static inline std::vector<F> generate_boolean_circuit_garble(const GarbleRow* rows, size_t n_rows, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  enum { SHARD = 0, CLK = 1, IS_REAL = 2, INPUT = 3, OUTPUT = 4, IS_FIRST_ROW = 5, IS_GATE = 6, IS_FIRST_GATE = 7, IS_LAST_GATE = 8, NOT_LAST_GATE = 9, GATE_TYPE = 10,
         GATE_ID = 12, GATES_NUM = 13, DELTA = 14, MEM = 30, RESULT_MEM = 183, AUX1 = 196, AUX2 = 212, AUX3 = 228, IS_EQ = 244, CHECKS = 288 };
  const size_t h = padded_rows(n_rows, fixed_log2_rows);
  std::vector<F> t(h * GARBLE_WIDTH, 0);
  std::vector<ByteLookup> lk;
  auto read = [&](const MemoryReadRecord& m, F* r) { memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r, &lk); };
  auto xor_word = [&](uint32_t x, uint32_t y, F* r) {      // XorOperation::populate (operations/xor.rs:22-37)
    for (int k = 0; k < 4; k++) {
      r[k] = ((x ^ y) >> (8 * k)) & 0xff;
      lk.push_back(ByteLookup{B_XOR, (uint8_t)(x >> (8 * k)), (uint8_t)(y >> (8 * k))});
    }
    return x ^ y;
  };
  for (size_t i = 0; i < n_rows; i++) {
    const GarbleRow& g = rows[i];
    F* r = t.data() + i * GARBLE_WIDTH;
    r[SHARD] = fu32(g.shard); r[CLK] = fu32(g.clk); r[IS_REAL] = 1; r[INPUT] = fu32(g.input_address); r[OUTPUT] = fu32(g.output_address);
    r[GATES_NUM] = fu32(g.gates_num);
    for (int k = 0; k < 16; k++) r[DELTA + k] = (g.delta[k / 4] >> (8 * (k % 4))) & 0xff;
    if (!g.is_gate) {
      if (g.reads[0].value != g.gates_num || g.gates_num == 0) throw std::runtime_error("tracegen: garble: the header row does not read the gate count");
      for (int k = 0; k < 4; k++)
        if (g.reads[1 + k].value != g.delta[k]) throw std::runtime_error("tracegen: garble: the header row does not read delta");
      r[IS_FIRST_ROW] = 1;
      for (int k = 0; k < 5; k++) read(g.reads[k], r + MEM + 9 * k);
      continue;
    }
    // a gate row continues the row before it
    if (i == 0) throw std::runtime_error("tracegen: garble: a gate row without a header row");
    const GarbleRow& prev = rows[i - 1];
    const bool chained = prev.shard == g.shard && prev.clk == g.clk && prev.gates_num == g.gates_num && prev.output_address == g.output_address &&
                         std::equal(g.delta, g.delta + 4, prev.delta) &&
                         (g.gate_id == 0 ? !prev.is_gate && g.input_address == prev.input_address + 20 && g.pre_check == 1
                                         : prev.is_gate && prev.gate_id + 1 == g.gate_id && g.input_address == prev.input_address + 68 &&
                                               g.pre_check == (prev.pre_check && garble_gate_ok(prev) ? 1u : 0u));
    if (!chained || g.gate_id >= g.gates_num) throw std::runtime_error("tracegen: garble: a gate row does not continue the row before it");
    const uint32_t type = g.reads[0].value;
    if (type != 0 && type != 7) throw std::runtime_error("tracegen: garble: gate type");
    const bool last = g.gate_id + 1 == g.gates_num;
    r[IS_GATE] = 1; r[IS_FIRST_GATE] = g.gate_id == 0; r[IS_LAST_GATE] = last; r[NOT_LAST_GATE] = !last; r[GATE_TYPE + (type ? 1 : 0)] = 1; r[GATE_ID] = fu32(g.gate_id);
    for (int k = 0; k < 17; k++) read(g.reads[k], r + MEM + 9 * k);
    uint32_t running = 1, check[4];
    for (int k = 0; k < 4; k++) {
      const uint32_t inter1 = xor_word(g.reads[1 + k].value, g.reads[5 + k].value, r + AUX1 + 4 * k);
      const uint32_t inter2 = xor_word(inter1, g.reads[9 + k].value, r + AUX2 + 4 * k);
      const uint32_t inter3 = xor_word(inter2, g.delta[k], r + AUX3 + 4 * k);
      is_equal_word_cols(type ? inter3 : inter2, g.reads[13 + k].value, r + IS_EQ + 11 * k);
      running = running && (type ? inter3 : inter2) == g.reads[13 + k].value;
      check[k] = running;
    }
    r[CHECKS] = check[1]; r[CHECKS + 1] = check[2]; r[CHECKS + 2] = check[3]; r[CHECKS + 3] = check[3] && g.pre_check;
    if (last) {
      if (g.write.value != (check[3] && g.pre_check ? 1u : 0u)) throw std::runtime_error("tracegen: garble: the last gate's row does not write the result");
      memory_write_cols(g.write, r + RESULT_MEM, &lk);
    }
  }
  if (n_rows && rows[n_rows - 1].is_gate && rows[n_rows - 1].gate_id + 1 != rows[n_rows - 1].gates_num) throw std::runtime_error("tracegen: garble: the last call is cut short");
  if (n_rows && !rows[n_rows - 1].is_gate) throw std::runtime_error("tracegen: garble: the last call is cut short");
  for (size_t i = 0; i + 1 < n_rows; i++)
    if (!rows[i + 1].is_gate && (!rows[i].is_gate || rows[i].gate_id + 1 != rows[i].gates_num)) throw std::runtime_error("tracegen: garble: a call is cut short");
  if (byte_counts)
    for (const ByteLookup& bl : lk) byte_counts[((size_t)bl.b * 256 + bl.c) * NUM_BYTE_OPS + bl.op]++;
  *height = h;
  return t;
}

// ---- SysLinux (syscall/precompiles/sys_linux/columns.rs:20-82, trace.rs:104-233): one Linux syscall per row, 103 columns: shard, clk, syscall_id, a0, a1,
// result (words), inorout and output (MemoryReadWriteCols: the branch's own access — register BRK, $a2 or HEAP — and the write of $a3), eight
// IsZeroOperations decoding the syscall id, is_mmap, five decoding a0 / a1, three composite flags, the mmap columns (the two nibbles of a1's
// second byte as bits, IsZero of the page offset, the size as a word, two carries, an AddOperation for the new heap), GtColsBytes for brk,
// is_real. Padding rows are zero.
struct LinuxEvent { uint32_t shard, clk, a0, a1, v0, syscall_code; MemoryReadRecord read_record; MemoryWriteRecord a3_record, heap_record; };
static_assert(sizeof(LinuxEvent) == 4 * 23, "flattened LinuxEvent is 23 words");
static const size_t SYS_LINUX_WIDTH = 103;
static inline std::vector<F> generate_sys_linux(const LinuxEvent* events, size_t n_events, int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  enum { SHARD = 0, CLK = 1, ID = 2, A0 = 3, A1 = 7, RESULT = 11, INOROUT = 15, OUTPUT = 28, D_MMAP = 41, D_MMAP2 = 43, D_CLONE = 45, D_EXIT = 47, D_BRK = 49, D_FCNTL = 51,
         D_READ = 53, D_WRITE = 55, IS_MMAP = 57, D_A0_0 = 58, D_A0_1 = 60, D_A0_2 = 62, D_A1_1 = 64, D_A1_3 = 66, IS_MMAP_A0_0 = 68, IS_FCNTL_A1_1 = 69, IS_FCNTL_A1_3 = 70,
         LO_BITS = 71, HI_BITS = 75, PAGE_ZERO = 79, MMAP_SIZE = 81, SIZE_CARRY = 85, HEAP_ADD = 87, GT = 94, IS_REAL = 102 };
  enum { MMAP = 4210, MMAP2 = 4090, CLONE = 4120, EXIT_GROUP = 4246, BRK = 4045, FCNTL = 4055, READ = 4003, WRITE = 4004 };
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * SYS_LINUX_WIDTH, 0);
  std::vector<ByteLookup> lk;
  auto range = [&](uint32_t v) {
    lk.push_back(ByteLookup{B_U8RANGE, (uint8_t)v, (uint8_t)(v >> 8)});
    lk.push_back(ByteLookup{B_U8RANGE, (uint8_t)(v >> 16), (uint8_t)(v >> 24)});
  };
  for (size_t i = 0; i < n_events; i++) {
    const LinuxEvent& e = events[i];
    F* r = t.data() + i * SYS_LINUX_WIDTH;
    const uint32_t code = e.syscall_code;
    // what the executor would have done (syscalls/precompiles/sys_linux/*.rs): the value returned and the value written to $a3
    uint32_t v0 = 0, a3 = 0;
    const bool mmap = code == MMAP || code == MMAP2, fd = e.a0 <= 2;
    if (code == BRK) v0 = std::max(e.a0, e.read_record.value);
    else if (mmap) v0 = e.a0 == 0 ? e.heap_record.prev_value : e.a0;
    else if (code == CLONE) v0 = 1;
    else if (code == FCNTL) {
      if (e.a1 == 3) v0 = e.a0 == 0 ? 0 : fd ? 1 : 0xffffffffu;
      else if (e.a1 == 1) v0 = fd ? e.a0 : 0xffffffffu;
      else v0 = 0xffffffffu;
      a3 = v0 == 0xffffffffu ? 9 : 0;
    } else if (code == READ) { v0 = e.a0 == 0 ? 0 : 0xffffffffu; a3 = e.a0 == 0 ? 0 : 9; }
    else if (code == WRITE) v0 = e.read_record.value;
    if (v0 != e.v0 || a3 != e.a3_record.value) throw std::runtime_error("tracegen: SysLinux event does not return what the syscall returns");
    word(r + A0, e.a0); word(r + A1, e.a1); word(r + RESULT, e.v0);
    r[SHARD] = fu32(e.shard); r[CLK] = fu32(e.clk); r[ID] = fu32(code); r[IS_REAL] = 1;
    memory_write_cols(e.a3_record, r + OUTPUT, &lk);
    const F sid = fu32(code);
    const int decoders[8][2] = {{MMAP, D_MMAP}, {MMAP2, D_MMAP2}, {CLONE, D_CLONE}, {EXIT_GROUP, D_EXIT}, {BRK, D_BRK}, {FCNTL, D_FCNTL}, {READ, D_READ}, {WRITE, D_WRITE}};
    for (auto& d : decoders) is_zero_cols(fsub(sid, d[0]), r + d[1]);
    r[IS_MMAP] = mmap;
    const F a0f = fu32(e.a0), a1f = fu32(e.a1);
    is_zero_cols(a0f, r + D_A0_0); is_zero_cols(fsub(a0f, 1), r + D_A0_1); is_zero_cols(fsub(a0f, 2), r + D_A0_2);
    is_zero_cols(fsub(a1f, 1), r + D_A1_1); is_zero_cols(fsub(a1f, 3), r + D_A1_3);
    r[IS_MMAP_A0_0] = mmap && e.a0 == 0;
    r[IS_FCNTL_A1_1] = code == FCNTL && e.a1 == 1;
    r[IS_FCNTL_A1_3] = code == FCNTL && e.a1 == 3;
    auto read_inorout = [&]() {      // MemoryReadWriteCols::populate_read: the previous value is the value
      const MemoryReadRecord& m = e.read_record;
      word(r + INOROUT, m.value);
      memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + INOROUT + 4, &lk);
    };
    if (code == BRK) {
      // GtColsBytes::populate (operations/cmp.rs:34-93) of a0 against the BRK register
      const uint32_t a = e.a0, bb = e.read_record.value;
      uint32_t res = 0, a_byte = 0, b_byte = 0;
      bool flagged = false;
      for (int k = 3; k >= 0 && !flagged; k--) {
        const uint32_t x = (a >> (8 * k)) & 0xff, y = (bb >> (8 * k)) & 0xff;
        if (x != y) { r[GT + k] = 1; a_byte = x; b_byte = y; res = x > y; flagged = true; }
      }
      r[GT + 4] = a_byte; r[GT + 5] = b_byte; r[GT + 6] = res; r[GT + 7] = flagged;
      lk.push_back(ByteLookup{B_LTU, (uint8_t)b_byte, (uint8_t)a_byte});
      if (flagged) lk.push_back(ByteLookup{B_LTU, (uint8_t)a_byte, (uint8_t)b_byte});
      range(a); range(bb);
      read_inorout();
    } else if (mmap) {
      range(e.a0); range(e.a1);
      const uint32_t byte1 = (e.a1 >> 8) & 0xff, lo = byte1 & 15, hi = byte1 >> 4;
      for (int bit = 0; bit < 4; bit++) { r[LO_BITS + bit] = (lo >> bit) & 1; r[HI_BITS + bit] = (hi >> bit) & 1; }
      const uint32_t page_off = e.a1 & 0xfff, upper = (e.a1 >> 12) << 12;
      is_zero_cols(page_off, r + PAGE_ZERO);
      if (e.a0 == 0) {
        memory_write_cols(e.heap_record, r + INOROUT, &lk);
        const uint32_t size = page_off == 0 ? upper : upper + 0x1000;
        word(r + MMAP_SIZE, size);
        range(size);
        if (page_off != 0 && hi == 15) {
          r[SIZE_CARRY] = 1;
          if (((e.a1 >> 16) & 0xff) == 255) r[SIZE_CARRY + 1] = 1;
        }
        // AddOperation::populate (operations/add.rs:23-57) of the old heap and the size
        const uint32_t old_heap = e.heap_record.prev_value, sum = old_heap + size;
        if (e.heap_record.value != sum) throw std::runtime_error("tracegen: SysLinux mmap does not move the heap by the rounded size");
        word(r + HEAP_ADD, sum);
        uint32_t carry = 0;
        for (int k = 0; k < 3; k++) {
          carry = (((old_heap >> (8 * k)) & 0xff) + ((size >> (8 * k)) & 0xff) + carry) >> 8;
          r[HEAP_ADD + 4 + k] = carry;
        }
        range(old_heap); range(size); range(sum);
      }
    } else if (code == WRITE) {
      read_inorout();
    }
  }
  if (byte_counts)
    for (const ByteLookup& bl : lk) byte_counts[((size_t)bl.b * 256 + bl.c) * NUM_BYTE_OPS + bl.op]++;
  *height = h;
  return t;
}

// ---- Field-tower precompiles (syscall/precompiles/fptower/): FpOp (kind 0: one FieldOpCols, the operation chosen per event), Fp2AddSub (kind 1: two,
// add or subtract per event), Fp2Mul (kind 2: four products, a difference, a sum) over the base field of Bn254 or Bls12381. Events: shard, clk,
// x_ptr, y_ptr, [op — FieldOperation as a word: Add 0, Mul 1, Sub 2 —] W write records of x, W read records of y (W = N / 4 for FpOp, N / 2 for
// the Fp2 chips; no op word for Fp2Mul). x is the records' previous values, y their values. Padding rows: the operations of the zero inputs with
// the flag is_add set (fp.rs:150-166, fp2_addsub.rs:158-178; Fp2Mul has no flag: fp2_mul.rs:186-204).
static inline std::vector<F> generate_fp_tower(const uint32_t* events, size_t n_events, int kind, int n_limbs, const uint8_t* modulus_bytes, int64_t offset,
                                               int fixed_log2_rows, size_t* height, uint64_t* byte_counts) {
  using namespace bigfield;
  const int W = kind == 0 ? n_limbs / 4 : n_limbs / 2, G = 6 * n_limbs - 4, HEAD = kind == 0 ? 8 : kind == 1 ? 6 : 5;
  const int X_ACCESS = HEAD, Y_ACCESS = HEAD + 13 * W, GADGETS = HEAD + 22 * W, NG = kind == 0 ? 1 : kind == 1 ? 2 : 6;
  const size_t width = GADGETS + NG * G, ev_words = (kind == 2 ? 4 : 5) + 11 * W;
  const size_t h = padded_rows(n_events, fixed_log2_rows);
  std::vector<F> t(h * width, 0);
  std::vector<ByteLookup> lk;
  const Big p = from_bytes(modulus_bytes, n_limbs);
  auto kind_of = [](uint32_t op) { return op == 0 ? FOP_ADD : op == 1 ? FOP_MUL : FOP_SUB; };
  auto fill = [&](F* r, const Big* x, const Big* y, uint32_t op, std::vector<ByteLookup>* sink, Big* out) {
    const FieldGadgets g{p, n_limbs, offset, sink};
    auto col = [&](int k) { return r + GADGETS + G * k; };
    if (kind == 0) {
      out[0] = field_op_cols(g, col(0), x[0], y[0], kind_of(op));
    } else if (kind == 1) {
      out[0] = field_op_cols(g, col(0), x[0], y[0], kind_of(op));
      out[1] = field_op_cols(g, col(1), x[1], y[1], kind_of(op));
    } else {        // a0_mul_b0, a1_mul_b1, a0_mul_b1, a1_mul_b0, c0, c1
      const Big a0b0 = field_op_cols(g, col(0), x[0], y[0], FOP_MUL), a1b1 = field_op_cols(g, col(1), x[1], y[1], FOP_MUL);
      const Big a0b1 = field_op_cols(g, col(2), x[0], y[1], FOP_MUL), a1b0 = field_op_cols(g, col(3), x[1], y[0], FOP_MUL);
      out[0] = field_op_cols(g, col(4), a0b0, a1b1, FOP_SUB);
      out[1] = field_op_cols(g, col(5), a0b1, a1b0, FOP_ADD);
    }
  };
  std::vector<F> padding(width, 0);
  Big zero2[2], out[2];
  fill(padding.data(), zero2, zero2, 0, nullptr, out);
  if (kind != 2) padding[3] = 1;              // is_add
  std::vector<uint32_t> xw(W), yw(W);
  for (size_t i = 0; i < h; i++) {
    F* r = t.data() + i * width;
    if (i >= n_events) { std::copy(padding.begin(), padding.end(), r); continue; }
    const uint32_t* e = events + i * ev_words;
    const uint32_t op = kind == 2 ? 1u : e[4];
    if (kind == 0 ? op > 2 : (kind == 1 && op != 0 && op != 2)) throw std::runtime_error("tracegen: field-tower operation");
    const MemoryWriteRecord* xrec = (const MemoryWriteRecord*)(e + (kind == 2 ? 4 : 5));
    const MemoryReadRecord* yrec = (const MemoryReadRecord*)(e + (kind == 2 ? 4 : 5) + 6 * W);
    r[0] = 1; r[1] = fu32(e[0]); r[2] = fu32(e[1]);
    if (kind == 0) { r[3] = op == 0; r[4] = op == 2; r[5] = op == 1; r[6] = fu32(e[2]); r[7] = fu32(e[3]); }
    else if (kind == 1) { r[3] = op == 0; r[4] = fu32(e[2]); r[5] = fu32(e[3]); }
    else { r[3] = fu32(e[2]); r[4] = fu32(e[3]); }
    for (int k = 0; k < W; k++) { xw[k] = xrec[k].prev_value; yw[k] = yrec[k].value; }
    const int per = n_limbs / 4;
    Big x[2], y[2];
    x[0] = from_words(xw.data(), per); y[0] = from_words(yw.data(), per);
    if (kind != 0) { x[1] = from_words(xw.data() + per, per); y[1] = from_words(yw.data() + per, per); }
    for (int k = 0; k < (kind == 0 ? 1 : 2); k++)
      if (cmp(x[k], p) >= 0 || cmp(y[k], p) >= 0) throw std::runtime_error("tracegen: field-tower operand is not below the modulus");
    fill(r, x, y, op, &lk, out);
    for (int k = 0; k < W; k++) {
      const MemoryReadRecord& m = yrec[k];
      memory_access_cols(m.value, m.shard, m.timestamp, m.prev_shard, m.prev_timestamp, r + Y_ACCESS + 9 * k, &lk);
    }
    for (int k = 0; k < W; k++) {
      memory_write_cols(xrec[k], r + X_ACCESS + 13 * k, &lk);
      const Big& coord = out[k / per];
      for (int c = 0; c < 4; c++)
        if (r[X_ACCESS + 13 * k + 4 + c] != limb(coord, 4 * (k % per) + c)) throw std::runtime_error("tracegen: field-tower event does not write the result");
    }
  }
  if (byte_counts)
    for (const ByteLookup& b : lk) byte_counts[((size_t)b.b * 256 + b.c) * NUM_BYTE_OPS + b.op]++;
  *height = h;
  return t;
}

}  // namespace tracegen
