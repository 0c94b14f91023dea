// Poseidon2 width-16 over KoalaBear (Montgomery form) for gfx950 kernels and the host
// transcript. Same permutation as zkm_primitives::poseidon2_init
// (crates/primitives/src/lib.rs:1107-1122; layers crates/recursion/core/include/poseidon2.hpp:21-71):
// initial external layer, 4 full rounds, 13 partial rounds on lane 0, 4 full rounds, S-box x^3.
//
// The 16-word state lives in VGPRs. The external layer is the add-only circ(2 M4, M4, M4, M4)
// form; the internal layer is s_i <- s_i * V_i + sum(s) with the diagonal
//   V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/8, 2^-24, -2^-8, -1/8, -1/16, -2^-24],
// computed for all 16 lanes as one 64-bit multiply-add and a signed Montgomery step, the lanes staying unreduced
// int32 words across the 13 partial rounds (see "partial rounds in a signed representation" below).
// Round constants come from the reference's table (poseidon2_constants.inc).
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

// (s + rc)^3 with the round-constant addition and both intermediate corrections folded away:
// `rcm` = rc - p (two's complement), so y = s + rcm lies in [-p, p) as a signed word; a signed Montgomery
// step maps x in (-p^2, p^2) to (x - t p) / 2^32 in (-p, p) with t = x * p^-1 taken as a signed word,
// so y^2 and then (y^2) * y stay signed and only the final value is brought back to [0, p).
// 11 instructions instead of 15 (add-reduce + two full multiplies).
KB_HD int32_t sbox_rc_signed(uint32_t s, uint32_t rcm) {
  const int32_t y = (int32_t)(s + rcm);
  const int64_t x1 = (int64_t)y * y;
  const int32_t t1 = (int32_t)((uint32_t)x1 * kb::MU);
  uint32_t hi1 = (uint32_t)((uint64_t)x1 >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(hi1));  // keep the next line a 32-bit subtract (the optimiser otherwise widens it to a borrow chain)
#endif
  const int32_t z = (int32_t)(hi1 - (uint32_t)kb::mulhi_s32(t1, (int32_t)kb::P));
  const int64_t x2 = (int64_t)z * y;
  const int32_t t2 = (int32_t)((uint32_t)x2 * kb::MU);
  return (int32_t)((uint32_t)((uint64_t)x2 >> 32) - (uint32_t)kb::mulhi_s32(t2, (int32_t)kb::P));  // in (-p, p)
}
KB_HD uint32_t sbox_rc(uint32_t s, uint32_t rcm) {
  const uint32_t r = (uint32_t)sbox_rc_signed(s, rcm);
  return kb::umin32(r, r + kb::P);
}

// ---- partial rounds in a signed representation ----------------------------------------------------------------
// Between the 13 partial rounds every lane only feeds (a) the lane sum and (b) its own s_i * V_i + sum, both of which
// a *signed* Montgomery step accepts: a lane is kept as an int32 congruent to its value, |s_i| < 2^31, and the final
// correction to [0, p) (two instructions per lane per round) is dropped. One round is then, per lane, a 64-bit
// multiply-add and a three-instruction reduction; only lane 0 is lifted (s or s + p) for its S-box.
// Bounds: a reduction returns |r| <= |x| / 2^32 + p / 2 with |x| <= |s| V + |sum| R < 2^31 p + 2^56, so
// |r| < p + 2^23.3 < 0.997 * 2^31 whatever the round: no int32 ever overflows, and a lifted lane lies in (-2^23.3, 0.997 * 2^31),
// for which the S-box's own signed arithmetic (|s + rc - p| < 2^31, squares below 2^31 p) still holds.

// acc += x for a signed 32-bit x: one v_mad_i64_i32 on the device
KB_HD void acc_add_signed(int64_t& acc, int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_i64_i32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(x) : "vcc");
#else
  acc += x;
#endif
}
// s in (-2^31, 2^31) congruent to a field value  ->  s or s + p: what sbox_rc_signed and the first S-boxes of the next
// full round accept (see the bounds above)
KB_HD uint32_t lift_signed(int32_t s) { return (uint32_t)s + ((uint32_t)(s >> 31) & kb::P); }

template <class DiagFn>
KB_HD void internal_layer_signed(int32_t s[16], DiagFn diag) {
  // S = sum of the 16 lanes, |S| < 2^35. With q = round(S / 2^31): S - q p = (S - q 2^31) + q (2^24 - 1) lies in
  // (-1.25 * 2^30, 1.25 * 2^30), so it is exact in 32-bit wrap-around arithmetic.
  int64_t acc = (int64_t)s[0] + (1 << 30);
#pragma unroll
  for (int i = 1; i < 16; i++) acc_add_signed(acc, s[i]);
  const uint32_t q = (uint32_t)(acc >> 31);
  const int32_t sum = (int32_t)((uint32_t)acc - q * kb::P - (1u << 30));
  const int64_t sum_r = (int64_t)sum * (int64_t)kb::ONE;  // sum * R: rides inside the reduction below
#pragma unroll
  for (int i = 0; i < 16; i++) {
    s[i] = kb::monty_reduce_signed(kb::mad_i64_i32_uniform(s[i], diag(i), sum_r));  // diag(i): wave-uniform table entry
#if defined(__HIP_DEVICE_COMPILE__)
    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four 64-bit products in flight, not sixteen: 24 fewer VGPRs
#endif
  }
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
  {
    int32_t t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = (int32_t)s[i];
#pragma unroll 1
    for (int r = 0; r < 13; r++) {
      t[0] = sbox_rc_signed(lift_signed(t[0]), rc_int(r));  // stays signed: the layer below takes it as it is
      internal_layer_signed(t, diag);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = lift_signed(t[i]);
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
