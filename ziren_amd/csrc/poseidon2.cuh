// Poseidon2 width-16 over KoalaBear (Montgomery form) for gfx950 kernels and the host
// transcript. Same permutation as zkm_primitives::poseidon2_init
// (crates/primitives/src/lib.rs:1107-1122; layers crates/recursion/core/include/poseidon2.hpp:21-71):
// initial external layer, 4 full rounds, 13 partial rounds on lane 0, 4 full rounds, S-box x^3.
//
// The 16-word state lives in VGPRs. The external layer is the add-only circ(2 M4, M4, M4, M4)
// form; the internal layer multiplies by the diagonal
//   [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/8, 2^-24, -2^-8, -1/8, -1/16, -2^-24]
// whose small-integer entries are done with adds; only the inverse powers of two use a
// Montgomery multiply. Round constants come from the reference's table (poseidon2_constants.inc).
#pragma once
#include "kb31.cuh"

namespace p2 {

#include "poseidon2_constants.inc"

// device copies live in constant memory (scalar loads: indices are wave-uniform); the round constants are
// stored as rc - p (two's complement), the form sbox_rc() consumes
__constant__ uint32_t d_rc_ext[8][16];
__constant__ uint32_t d_rc_int[13];
__constant__ uint32_t d_diag[16];

KB_HD void m4(uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3) {
  // [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
  uint32_t t01 = kb::add(s0, s1), t23 = kb::add(s2, s3);
  uint32_t t0123 = kb::add(t01, t23);
  uint32_t t01123 = kb::add(t0123, s1), t01233 = kb::add(t0123, s3);
  uint32_t n3 = kb::add(t01233, kb::dbl(s0));
  uint32_t n1 = kb::add(t01123, kb::dbl(s2));
  uint32_t n0 = kb::add(t01123, t01);
  uint32_t n2 = kb::add(t01233, t23);
  s0 = n0; s1 = n1; s2 = n2; s3 = n3;
}

KB_HD void external_layer(uint32_t s[16]) {
#pragma unroll
  for (int i = 0; i < 16; i += 4) m4(s[i], s[i + 1], s[i + 2], s[i + 3]);
  uint32_t c0 = kb::add(kb::add(s[0], s[4]), kb::add(s[8], s[12]));
  uint32_t c1 = kb::add(kb::add(s[1], s[5]), kb::add(s[9], s[13]));
  uint32_t c2 = kb::add(kb::add(s[2], s[6]), kb::add(s[10], s[14]));
  uint32_t c3 = kb::add(kb::add(s[3], s[7]), kb::add(s[11], s[15]));
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    s[i] = kb::add(s[i], c0); s[i + 1] = kb::add(s[i + 1], c1);
    s[i + 2] = kb::add(s[i + 2], c2); s[i + 3] = kb::add(s[i + 3], c3);
  }
}

KB_HD uint32_t sbox(uint32_t x) { return kb::mul(kb::sqr(x), x); }

KB_HD int32_t mulhi_s32(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __mulhi(a, b);
#else
  return (int32_t)(((int64_t)a * b) >> 32);
#endif
}
// (s + rc)^3 with the round-constant addition and both intermediate corrections folded away:
// `rcm` = rc - p (two's complement), so y = s + rcm lies in [-p, p) as a signed word; a signed Montgomery
// step maps x in (-p^2, p^2) to (x - t p) / 2^32 in (-p, p) with t = x * p^-1 taken as a signed word,
// so y^2 and then (y^2) * y stay signed and only the final value is brought back to [0, p).
// 11 instructions instead of 15 (add-reduce + two full multiplies).
KB_HD uint32_t sbox_rc(uint32_t s, uint32_t rcm) {
  const int32_t y = (int32_t)(s + rcm);
  const int64_t x1 = (int64_t)y * y;
  const int32_t t1 = (int32_t)((uint32_t)x1 * kb::MU);
  uint32_t hi1 = (uint32_t)((uint64_t)x1 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(hi1));  // keep the next line a 32-bit subtract (the optimiser otherwise widens it to a borrow chain)
#endif
  const int32_t z = (int32_t)(hi1 - (uint32_t)mulhi_s32(t1, (int32_t)kb::P));
  const int64_t x2 = (int64_t)z * y;
  const int32_t t2 = (int32_t)((uint32_t)x2 * kb::MU);
  const uint32_t r = (uint32_t)((uint64_t)x2 >> 32) - (uint32_t)mulhi_s32(t2, (int32_t)kb::P);
  return kb::umin32(r, r + kb::P);
}

// acc += x (64-bit accumulate of a 32-bit value): one v_mad_u64_u32 on the device
KB_HD void acc_add(uint64_t& acc, uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(x) : "vcc");
#else
  acc += x;
#endif
}
// x / 2 mod p
KB_HD uint32_t half(uint32_t x) { return (x + ((0u - (x & 1u)) & kb::P)) >> 1; }

template <class DiagFn>
KB_HD void internal_layer(uint32_t s[16], DiagFn diag) {
  // sum of the 16 lanes: eight unreduced pair sums (< 2p < 2^32), accumulated in 64 bits, one reduction:
  // acc = q 2^31 + l  =>  acc - q p = q (2^24 - 1) + l < 2p
  uint64_t acc = (uint64_t)(s[0] + s[1]);
#pragma unroll
  for (int i = 2; i < 16; i += 2) acc_add(acc, s[i] + s[i + 1]);
  uint32_t q = (uint32_t)(acc >> 31);
  uint32_t r = (uint32_t)acc - q * kb::P;
  uint32_t sum = kb::umin32(r, r - kb::P);
  // s_i <- s_i * V_i + sum. Where V_i costs more than one doubling, the addition rides inside the Montgomery
  // reduction: (s_i V_i + sum * R) / R = s_i V_i / R + sum, one 64-bit multiply-add and one reduction.
  const uint64_t sum_r = (uint64_t)sum * kb::ONE;
  s[0] = kb::sub(sum, kb::dbl(s[0]));                 // -2
  s[1] = kb::add(sum, s[1]);                          //  1
  s[2] = kb::add(sum, kb::dbl(s[2]));                 //  2
  s[3] = kb::add(sum, half(s[3]));                    //  1/2
  s[4] = kb::monty_reduce((uint64_t)s[4] * diag(4) + sum_r);   //  3
  s[5] = kb::monty_reduce((uint64_t)s[5] * diag(5) + sum_r);   //  4
  s[6] = kb::sub(sum, half(s[6]));                    // -1/2
  s[7] = kb::monty_reduce((uint64_t)s[7] * diag(7) + sum_r);   // -3
  s[8] = kb::monty_reduce((uint64_t)s[8] * diag(8) + sum_r);   // -4
#pragma unroll
  for (int i = 9; i < 16; i++) s[i] = kb::monty_reduce((uint64_t)s[i] * diag(i) + sum_r);
}

template <class RcExt, class RcInt, class DiagFn>
KB_HD void permute_impl(uint32_t s[16], RcExt rc_ext, RcInt rc_int, DiagFn diag) {
  external_layer(s);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    uint32_t rc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) rc[i] = rc_ext(r, i);
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = sbox_rc(s[i], rc[i]);
    external_layer(s);
  }
#pragma unroll 1
  for (int r = 0; r < 13; r++) {
    s[0] = sbox_rc(s[0], rc_int(r));
    internal_layer(s, diag);
  }
#pragma unroll
  for (int r = 4; r < 8; r++) {
    uint32_t rc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) rc[i] = rc_ext(r, i);
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = sbox_rc(s[i], rc[i]);
    external_layer(s);
  }
}

__device__ __forceinline__ void permute(uint32_t s[16]) {
  permute_impl(
      s, [](int r, int i) { return d_rc_ext[r][i]; }, [](int r) { return d_rc_int[r]; },
      [](int i) { return d_diag[i]; });
}

inline void permute_host(uint32_t s[16]) {
  permute_impl(
      s, [](int r, int i) { return ZKM_RC_16_30_MONTY[r < 4 ? r : r + 13][i] - kb::P; },
      [](int r) { return ZKM_RC_16_30_MONTY[4 + r][0] - kb::P; }, [](int i) { return ZKM_INTERNAL_DIAG_16_MONTY[i]; });
}

// upload tables into constant memory (once per process, any device)
inline hipError_t upload_tables() {
  uint32_t ext[8][16], in[13];
  for (int r = 0; r < 8; r++)
    for (int i = 0; i < 16; i++) ext[r][i] = ZKM_RC_16_30_MONTY[r < 4 ? r : r + 13][i] - kb::P;
  for (int r = 0; r < 13; r++) in[r] = ZKM_RC_16_30_MONTY[4 + r][0] - kb::P;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(d_rc_ext), ext, sizeof ext);
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(d_rc_int), in, sizeof in);
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(d_diag), ZKM_INTERNAL_DIAG_16_MONTY, sizeof(uint32_t) * 16);
}

}  // namespace p2
