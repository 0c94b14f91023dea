// KoalaBear (p = 2^31 - 2^24 + 1) in Montgomery form, R = 2^32, and its quartic extension
// F[X]/(X^4 - 3), for gfx950 device code and the host-side transcript.
//
// Representation is the one Plonky3's KoalaBear (MontyField31) keeps in memory and the one
// the reference's kb31_t uses (crates/core/machine/include/kb31_t.hpp:458-503), so trace
// words cross the ABI untouched. Reduction is written for the CDNA4 integer pipe: one
// 32x32->64 product (v_mad_u64_u32 / v_mul_hi_u32), the Montgomery quotient as shift-adds
// (MU = 2^31 + 2^24 + 1), one v_mul_hi_u32 by p and a min-based final correction — no branches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KB_HD __host__ __device__ __forceinline__

namespace kb {

constexpr uint32_t P = 0x7f000001u;
constexpr uint32_t MU = 0x81000001u;    // p^-1 mod 2^32
constexpr uint32_t ONE = 0x01fffffeu;   // R mod p
constexpr uint32_t R2 = 0x17f7efe4u;    // R^2 mod p
constexpr uint32_t GEN = 0x05fffffau;   // 3 * R mod p  (multiplicative generator 3)
constexpr int TWO_ADICITY = 24;

KB_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

KB_HD uint32_t add(uint32_t a, uint32_t b) {
  uint32_t s = a + b;
  return umin32(s, s - P);
}
KB_HD uint32_t sub(uint32_t a, uint32_t b) {
  uint32_t d = a - b;
  return umin32(d, d + P);
}
// a - b + p in (0, 2p): not reduced; valid only as the `a` operand of mul (a < 2^32, b < p suffices)
KB_HD uint32_t sub_lazy(uint32_t a, uint32_t b) { return a - b + P; }
KB_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }
KB_HD uint32_t dbl(uint32_t a) { return add(a, a); }

// x < 2^32 * p  ->  x * R^-1 mod p, in [0, p)
KB_HD uint32_t monty_reduce(uint64_t x) {
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  uint32_t t = lo + (lo << 24) + (lo << 31);  // lo * MU mod 2^32
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t uhi = __umulhi(t, P);
#else
  uint32_t uhi = (uint32_t)(((uint64_t)t * P) >> 32);
#endif
  uint32_t r = hi - uhi;  // low words of x and t*p coincide, so no borrow from them
  return umin32(r, r + P);
}
KB_HD uint32_t mul(uint32_t a, uint32_t b) { return monty_reduce((uint64_t)a * b); }
KB_HD uint32_t sqr(uint32_t a) { return mul(a, a); }
// d * b for a signed d in (-p, p) (e.g. a difference of two reduced values, unreduced) and b in [0, p):
// the Montgomery quotient is taken as a signed word, so (x - t p) / 2^32 lies in (-p, p); one correction.
KB_HD uint32_t mul_signed(uint32_t d_twos_complement, uint32_t b) {
  const int64_t x = (int64_t)(int32_t)d_twos_complement * (int64_t)b;
  const int32_t t = (int32_t)((uint32_t)x * MU);
#if defined(__HIP_DEVICE_COMPILE__)
  const int32_t uhi = __mulhi(t, (int32_t)P);
#else
  const int32_t uhi = (int32_t)(((int64_t)t * (int64_t)P) >> 32);
#endif
  const uint32_t r = (uint32_t)((int32_t)(x >> 32) - uhi);
  return umin32(r, r + P);
}
// ---- signed (unreduced) arithmetic: values kept as int32 words congruent to the field element, |v| < 2^31 -----------
KB_HD int32_t mulhi_s32(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __mulhi(a, b);
#else
  return (int32_t)(((int64_t)a * b) >> 32);
#endif
}
// a * b + c, signed 32 x 32 + 64: one v_mad_i64_i32 on the device. Pinned with inline asm: left to itself the compiler
// sometimes expands the 64-bit product into an unsigned multiply plus sign fix-ups (three to four instructions).
KB_HD int64_t mad_i64_i32(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t r;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
  return r;
#else
  return (int64_t)a * (int64_t)b + c;
#endif
}
// the same with a wave-uniform b in [0, 2^31) (a constant or a table entry with a uniform index): b stays in an SGPR
KB_HD int64_t mad_i64_i32_uniform(int32_t a, uint32_t b_uniform, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t r;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c) : "vcc");
  return r;
#else
  return (int64_t)a * (int64_t)(int32_t)b_uniform + c;
#endif
}
// x in (-2^31 p, 2^31 p)  ->  (x - t p) / 2^32 with t = x p^-1 mod 2^32 taken as a signed word: congruent to x / R,
// |result| <= |x| / 2^32 + p / 2, no correction to [0, p)
KB_HD int32_t monty_reduce_signed(int64_t x) {
  const int32_t t = (int32_t)((uint32_t)x * MU);
  uint32_t hi = (uint32_t)((uint64_t)x >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(hi));  // keep the next line a 32-bit subtract (the optimiser otherwise widens it to a borrow chain)
#endif
  return (int32_t)(hi - (uint32_t)mulhi_s32(t, (int32_t)P));
}
KB_HD uint32_t to_monty(uint32_t canonical) { return mul(canonical, R2); }
KB_HD uint32_t from_monty(uint32_t m) { return monty_reduce((uint64_t)m); }

KB_HD uint32_t pow(uint32_t a, uint64_t e) {
  uint32_t r = ONE;
  while (e) {
    if (e & 1) r = mul(r, a);
    a = sqr(a);
    e >>= 1;
  }
  return r;
}
// a b / R for signed words |a|, |b| < p, as a signed word in (-p, p): |a b| / 2^32 + p / 2 < 0.9961 p, so a chain of products needs no
// correction between them (four instructions a product instead of six)
KB_HD int32_t mul_ss(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t x;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(x) : "v"(a), "v"(b) : "vcc");
  return monty_reduce_signed(x);
#else
  return monty_reduce_signed((int64_t)a * (int64_t)b);
#endif
}
// a^(p-2); p - 2 = 0x7effffff. The chain runs on signed unreduced words (mul_ss) and is corrected to [0, p) once at the end.
KB_HD uint32_t inv(uint32_t a_canonical) {
  // addition chain: a^(2^24 - 1), then a^(127 * 2^24 - 1) = a^(p-2)
  const int32_t a = (int32_t)a_canonical;
  int32_t x2 = mul_ss(mul_ss(a, a), a);      // 2^2-1
  int32_t x3 = mul_ss(mul_ss(x2, x2), a);    // 2^3-1
  int32_t x6 = x3;
  for (int i = 0; i < 3; i++) x6 = mul_ss(x6, x6);
  x6 = mul_ss(x6, x3);                       // 2^6-1
  int32_t x12 = x6;
  for (int i = 0; i < 6; i++) x12 = mul_ss(x12, x12);
  x12 = mul_ss(x12, x6);                     // 2^12-1
  int32_t x24 = x12;
  for (int i = 0; i < 12; i++) x24 = mul_ss(x24, x24);
  x24 = mul_ss(x24, x12);                    // 2^24-1
  // p - 2 = (2^7 - 2) * 2^24 + (2^24 - 1) = 0b1111110 followed by 24 ones
  int32_t r = mul_ss(x6, x6);                // a^(2^7 - 2)  (= (2^6-1)*2)
  for (int i = 0; i < 24; i++) r = mul_ss(r, r);
  const uint32_t u = (uint32_t)mul_ss(r, x24);
  return umin32(u, u + P);
}

KB_HD uint32_t two_adic_generator(int bits) { return pow(GEN, (uint64_t)(P - 1) >> bits); }

KB_HD uint32_t bitrev(uint32_t x, int bits) {
  if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x) >> (32 - bits);
#else
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
#endif
}

// (hi 2^64 + lo64) / R mod p in [0, p) for a sum known to be below 127 * 2^63 — a linear form of up to 126 products of a word below
// 2^32 by a reduced word, which is every linear form and folded constraint sum the generated kernels build (ziren_amd/codegen.py
// computes the bound of each and falls back to acc96_reduce beyond it). One Montgomery step on the low word leaves
// x1 = (x - t p) / 2^32 = (hi : mid) - mulhi(t, p), a signed value in (-p, 127 * 2^31); its quotient by 2^31, q in [-1, 126], is
// also a good enough quotient by p: x1 - q p = (x1 mod 2^31) + q (2^24 - 1) lies in [0, 2p) (p = 2^31 - 2^24 + 1), one conditional
// subtraction from the answer. Ten instructions against acc96_reduce's twenty-two (round 5).
KB_HD uint32_t reduce96_bounded(uint32_t hi, uint64_t lo64) {
  const uint32_t lo = (uint32_t)lo64, mid = (uint32_t)(lo64 >> 32);
  const uint32_t t = lo + (lo << 24) + (lo << 31);  // lo * MU mod 2^32
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t uhi = __umulhi(t, P);
#else
  const uint32_t uhi = (uint32_t)(((uint64_t)t * P) >> 32);
#endif
  const uint64_t x1 = ((((uint64_t)hi) << 32) | mid) - uhi;  // two's complement; the low words of x and t p coincide: no borrow from them
  const uint32_t q = (uint32_t)(x1 >> 31);                   // floor(x1 / 2^31): 0xffffffff for a negative x1
  const uint32_t low = (uint32_t)x1 & 0x7fffffffu;
  const uint32_t x2 = low + (q << 24) - q;                   // x1 - q p
  return umin32(x2, x2 - P);
}

// ---- long dot products: 96-bit accumulation, one reduction at the end ------------------------------------------------
// sum_i a_i b_i with a_i, b_i < 2^32: each term is one v_mad_u64_u32 into the low 64 bits plus the carry into a third
// word (two instructions per product instead of 3.5 with a Montgomery reduction every second product). Holds 2^32 terms.
#if defined(__HIPCC__)
struct Acc96 {
  uint64_t lo;
  uint32_t hi;
};
__device__ __forceinline__ Acc96 acc96_zero() { return Acc96{0, 0}; }
__device__ __forceinline__ void acc96_fma(Acc96& acc, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc.lo), "+v"(acc.hi) : "v"(a), "v"(b) : "vcc");
}
// the same with a wave-uniform a (a table entry with a uniform index): it stays in an SGPR
__device__ __forceinline__ void acc96_fma_uniform(Acc96& acc, uint32_t a_uniform, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc.lo), "+v"(acc.hi) : "s"(a_uniform), "v"(b) : "vcc");
}
// (hi 2^64 + lo) / R mod p, in [0, p): the middle word is reduced below p so the low 64 bits satisfy monty_reduce's
// precondition, and hi 2^64 / R = hi R is one more Montgomery product (2^32 < 2.02 p: two conditional subtractions)
__device__ __forceinline__ uint32_t acc96_reduce(const Acc96& acc) {
  uint32_t mid = (uint32_t)(acc.lo >> 32);
  mid = umin32(mid, mid - P);
  mid = umin32(mid, mid - P);
  const uint32_t r_lo = monty_reduce(((uint64_t)mid << 32) | (uint32_t)acc.lo);
  uint32_t h = acc.hi;            // number of carries: small, but reduce it anyway
  h = umin32(h, h - P);
  h = umin32(h, h - P);
  return add(r_lo, mul(h, R2));   // h * R^2 / R = h * R
}
#endif

// ---- quartic extension, X^4 = 3 (crates/stark/src/air/extension.rs:55-74) ------------------
struct alignas(16) E4 {
  uint32_t c[4];
};

KB_HD E4 ezero() { return E4{{0, 0, 0, 0}}; }
KB_HD E4 eone() { return E4{{ONE, 0, 0, 0}}; }
KB_HD E4 efrom(uint32_t a) { return E4{{a, 0, 0, 0}}; }
KB_HD bool eq(const E4& a, const E4& b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3]; }
KB_HD E4 eadd(const E4& a, const E4& b) { return E4{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}}; }
KB_HD E4 esub(const E4& a, const E4& b) { return E4{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}}; }
KB_HD E4 eneg(const E4& a) { return E4{{neg(a.c[0]), neg(a.c[1]), neg(a.c[2]), neg(a.c[3])}}; }
KB_HD E4 escale(const E4& a, uint32_t s) { return E4{{mul(a.c[0], s), mul(a.c[1], s), mul(a.c[2], s), mul(a.c[3], s)}}; }
KB_HD E4 eadd_base(const E4& a, uint32_t b) { return E4{{add(a.c[0], b), a.c[1], a.c[2], a.c[3]}}; }
KB_HD E4 esub_base(const E4& a, uint32_t b) { return E4{{sub(a.c[0], b), a.c[1], a.c[2], a.c[3]}}; }
KB_HD uint32_t mul3(uint32_t a) { return add(dbl(a), a); }

// Products are accumulated in 64 bits two at a time (2 p^2 < 2^32 p) before one reduction.
KB_HD uint32_t dot2(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  return monty_reduce((uint64_t)a0 * b0 + (uint64_t)a1 * b1);
}
// x < 4 p^2 (a sum of up to four products of reduced words: four v_mad_u64_u32 into one 64-bit accumulator) -> x / R mod p in [0, p).
// (x - t p) / 2^32 lies in (-p, 2p): the borrow of the high words tells which side of zero, then one conditional subtraction.
KB_HD uint32_t monty_reduce_wide(uint64_t x) {
  uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  uint32_t t = lo + (lo << 24) + (lo << 31);  // lo * MU mod 2^32
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t uhi = __umulhi(t, P);
#else
  uint32_t uhi = (uint32_t)(((uint64_t)t * P) >> 32);
#endif
  uint32_t r = hi - uhi;
  r += hi < uhi ? P : 0u;
  return umin32(r, r - P);
}
// Every coefficient is four products summed in 64 bits and reduced once (round 3: 77 instructions instead of 103 with a reduction
// every second product); the X^4 = 3 wrap-around is taken on a's side first (three modular triplings).
KB_HD E4 emul(const E4& a, const E4& b) {
  const uint32_t t1 = mul3(a.c[1]), t2 = mul3(a.c[2]), t3 = mul3(a.c[3]);
  E4 r;
  // c0 = a0 b0 + 3 (a1 b3 + a2 b2 + a3 b1)
  r.c[0] = monty_reduce_wide((uint64_t)a.c[0] * b.c[0] + (uint64_t)t1 * b.c[3] + (uint64_t)t2 * b.c[2] + (uint64_t)t3 * b.c[1]);
  // c1 = a0 b1 + a1 b0 + 3 (a2 b3 + a3 b2)
  r.c[1] = monty_reduce_wide((uint64_t)a.c[0] * b.c[1] + (uint64_t)a.c[1] * b.c[0] + (uint64_t)t2 * b.c[3] + (uint64_t)t3 * b.c[2]);
  // c2 = a0 b2 + a1 b1 + a2 b0 + 3 a3 b3
  r.c[2] = monty_reduce_wide((uint64_t)a.c[0] * b.c[2] + (uint64_t)a.c[1] * b.c[1] + (uint64_t)a.c[2] * b.c[0] + (uint64_t)t3 * b.c[3]);
  // c3 = a0 b3 + a1 b2 + a2 b1 + a3 b0
  r.c[3] = monty_reduce_wide((uint64_t)a.c[0] * b.c[3] + (uint64_t)a.c[1] * b.c[2] + (uint64_t)a.c[2] * b.c[1] + (uint64_t)a.c[3] * b.c[0]);
  return r;
}
KB_HD E4 esqr(const E4& a) { return emul(a, a); }
// Inverse through the tower F < F[Y]/(Y^2-3) < EF with Y = X^2:
// a = A + X B, A = a0 + a2 Y, B = a1 + a3 Y;  1/a = (A - X B) / (A^2 - Y B^2).
// The extension inverse in two halves around its one base-field inversion (54 of its 74 multiplications), so that a kernel with several
// inverses per thread can take all the base inversions in one (Montgomery's trick: inv_batch below): einv_norm gives the element's norm
// down to the base field and the two intermediate words, einv_finish takes the norm's inverse.
KB_HD uint32_t einv_norm(const E4& a, uint32_t& d0, uint32_t& d1) {
  uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
  // A^2 = (a0^2 + 3 a2^2) + (2 a0 a2) Y ;  B^2 = (a1^2 + 3 a3^2) + (2 a1 a3) Y
  // Y B^2 = 3 (2 a1 a3) + (a1^2 + 3 a3^2) Y
  d0 = sub(add(sqr(a0), mul3(sqr(a2))), mul3(dbl(mul(a1, a3))));
  d1 = sub(dbl(mul(a0, a2)), add(sqr(a1), mul3(sqr(a3))));
  // 1/(d0 + d1 Y) = (d0 - d1 Y) / (d0^2 - 3 d1^2)
  return sub(sqr(d0), mul3(sqr(d1)));
}
KB_HD E4 einv_finish(const E4& a, uint32_t d0, uint32_t d1, uint32_t nrm_inv) {
  uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
  uint32_t e0 = mul(d0, nrm_inv), e1 = neg(mul(d1, nrm_inv));
  E4 r;
  r.c[0] = add(mul(a0, e0), mul3(mul(a2, e1)));
  r.c[2] = add(mul(a0, e1), mul(a2, e0));
  r.c[1] = neg(add(mul(a1, e0), mul3(mul(a3, e1))));
  r.c[3] = neg(add(mul(a1, e1), mul(a3, e0)));
  return r;
}
// x[i] <- 1 / x[i] for N base-field words with ONE inversion (prefix products, invert, peel back: 3 (N - 1) multiplications); a zero
// stays zero, as inv(0) = 0 does
template <int N>
KB_HD void inv_batch(uint32_t (&x)[N]) {
  uint32_t nz[N], pre[N];
#pragma unroll
  for (int i = 0; i < N; i++) { nz[i] = x[i] != 0; x[i] = nz[i] ? x[i] : ONE; }
  pre[0] = x[0];
#pragma unroll
  for (int i = 1; i < N; i++) pre[i] = mul(pre[i - 1], x[i]);
  uint32_t iv = inv(pre[N - 1]);
#pragma unroll
  for (int i = N - 1; i > 0; i--) {
    const uint32_t xi = x[i];
    x[i] = mul(iv, pre[i - 1]);
    iv = mul(iv, xi);
  }
  x[0] = iv;
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = nz[i] ? x[i] : 0u;
}
KB_HD E4 einv(const E4& a) {
  uint32_t a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
  // A^2 = (a0^2 + 3 a2^2) + (2 a0 a2) Y ;  B^2 = (a1^2 + 3 a3^2) + (2 a1 a3) Y
  // Y B^2 = 3 (2 a1 a3) + (a1^2 + 3 a3^2) Y
  uint32_t d0 = sub(add(sqr(a0), mul3(sqr(a2))), mul3(dbl(mul(a1, a3))));
  uint32_t d1 = sub(dbl(mul(a0, a2)), add(sqr(a1), mul3(sqr(a3))));
  // 1/(d0 + d1 Y) = (d0 - d1 Y) / (d0^2 - 3 d1^2)
  uint32_t nrm = inv(sub(sqr(d0), mul3(sqr(d1))));
  uint32_t e0 = mul(d0, nrm), e1 = neg(mul(d1, nrm));
  // (A - X B) * (e0 + e1 Y):
  //   A (e0 + e1 Y) = (a0 e0 + 3 a2 e1) + (a0 e1 + a2 e0) Y
  //   B (e0 + e1 Y) = (a1 e0 + 3 a3 e1) + (a1 e1 + a3 e0) Y
  E4 r;
  r.c[0] = add(mul(a0, e0), mul3(mul(a2, e1)));
  r.c[2] = add(mul(a0, e1), mul(a2, e0));
  r.c[1] = neg(add(mul(a1, e0), mul3(mul(a3, e1))));
  r.c[3] = neg(add(mul(a1, e1), mul(a3, e0)));
  return r;
}
#if defined(__HIPCC__)
// The folded constraint sum of a quotient kernel, acc = sum_k alpha^(C-1-k) * constraint_k (folder.rs:79-102), kept as four 96-bit
// integer accumulators (one per extension coefficient) of unreduced 64-bit products and reduced once at the end: a base-field
// constraint costs four v_mad_u64_u32 + carry instead of four Montgomery products and four modular additions (8 instructions
// instead of 44), an extension-field one sixteen. alpha's powers are wave-uniform (SGPR operands). Room: a product is below 2^62,
// a chip has at most a few thousand constraints.
struct FoldAcc { Acc96 c[4]; };
__device__ __forceinline__ FoldAcc fold_zero() { return FoldAcc{{acc96_zero(), acc96_zero(), acc96_zero(), acc96_zero()}}; }
__device__ __forceinline__ void fold_base(FoldAcc& f, const E4& alpha_pow_uniform, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) acc96_fma_uniform(f.c[i], alpha_pow_uniform.c[i], v);
}
__device__ __forceinline__ void fold_ext(FoldAcc& f, const E4& a /* uniform */, const E4& e) {
  // (a0 + a1 X + a2 X^2 + a3 X^3)(e0 + ...) with X^4 = 3
  const uint32_t t1 = mul3(a.c[1]), t2 = mul3(a.c[2]), t3 = mul3(a.c[3]);
  acc96_fma_uniform(f.c[0], a.c[0], e.c[0]); acc96_fma_uniform(f.c[0], t1, e.c[3]); acc96_fma_uniform(f.c[0], t2, e.c[2]); acc96_fma_uniform(f.c[0], t3, e.c[1]);
  acc96_fma_uniform(f.c[1], a.c[0], e.c[1]); acc96_fma_uniform(f.c[1], a.c[1], e.c[0]); acc96_fma_uniform(f.c[1], t2, e.c[3]); acc96_fma_uniform(f.c[1], t3, e.c[2]);
  acc96_fma_uniform(f.c[2], a.c[0], e.c[2]); acc96_fma_uniform(f.c[2], a.c[1], e.c[1]); acc96_fma_uniform(f.c[2], a.c[2], e.c[0]); acc96_fma_uniform(f.c[2], t3, e.c[3]);
  acc96_fma_uniform(f.c[3], a.c[0], e.c[3]); acc96_fma_uniform(f.c[3], a.c[1], e.c[2]); acc96_fma_uniform(f.c[3], a.c[2], e.c[1]); acc96_fma_uniform(f.c[3], a.c[3], e.c[0]);
}
__device__ __forceinline__ E4 fold_finish(const FoldAcc& f) {
  return E4{{acc96_reduce(f.c[0]), acc96_reduce(f.c[1]), acc96_reduce(f.c[2]), acc96_reduce(f.c[3])}};
}
// What the generated kernels use for a linear form whose bound they know (codegen.emit_form): the accumulator starts at a constant
// (c R sits in the middle word: no addition afterwards), takes row values as one product by R mod p, products of two row values
// (the multiplicity-weighted denominators of a LogUp batch), and is reduced by reduce96_bounded.
__device__ __forceinline__ FoldAcc fold_from(const E4& c) {
  return FoldAcc{{Acc96{(uint64_t)c.c[0] << 32, 0}, Acc96{(uint64_t)c.c[1] << 32, 0}, Acc96{(uint64_t)c.c[2] << 32, 0}, Acc96{(uint64_t)c.c[3] << 32, 0}}};
}
__device__ __forceinline__ void fold_add(FoldAcc& f, const E4& e) {
#pragma unroll
  for (int i = 0; i < 4; i++) acc96_fma_uniform(f.c[i], ONE, e.c[i]);
}
__device__ __forceinline__ void fold_scaled(FoldAcc& f, const E4& e, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) acc96_fma(f.c[i], e.c[i], v);
}
__device__ __forceinline__ E4 fold_finish_bounded(const FoldAcc& f) {
  return E4{{reduce96_bounded(f.c[0].hi, f.c[0].lo), reduce96_bounded(f.c[1].hi, f.c[1].lo), reduce96_bounded(f.c[2].hi, f.c[2].lo),
             reduce96_bounded(f.c[3].hi, f.c[3].lo)}};
}
// The same in 64 bits, for a form whose bound is below 2^64 (no carry word: one v_mad_u64_u32 per product); `narrow`: the bound is
// below 2^32 p as well, so the reduction is an ordinary Montgomery reduction.
struct FoldAcc64 { uint64_t c[4]; };
__device__ __forceinline__ FoldAcc64 fold64_zero() { return FoldAcc64{{0, 0, 0, 0}}; }
__device__ __forceinline__ FoldAcc64 fold64_from(const E4& c) {
  return FoldAcc64{{(uint64_t)c.c[0] << 32, (uint64_t)c.c[1] << 32, (uint64_t)c.c[2] << 32, (uint64_t)c.c[3] << 32}};
}
__device__ __forceinline__ void fold64_scaled(FoldAcc64& f, const E4& e, uint32_t v) {
#pragma unroll
  for (int i = 0; i < 4; i++) f.c[i] += (uint64_t)e.c[i] * v;
}
__device__ __forceinline__ void fold64_add(FoldAcc64& f, const E4& e) {
#pragma unroll
  for (int i = 0; i < 4; i++) f.c[i] += (uint64_t)e.c[i] * ONE;
}
template <bool NARROW>
__device__ __forceinline__ E4 fold64_finish(const FoldAcc64& f) {
  if (NARROW) return E4{{monty_reduce(f.c[0]), monty_reduce(f.c[1]), monty_reduce(f.c[2]), monty_reduce(f.c[3])}};
  return E4{{reduce96_bounded(0, f.c[0]), reduce96_bounded(0, f.c[1]), reduce96_bounded(0, f.c[2]), reduce96_bounded(0, f.c[3])}};
}
#endif
KB_HD E4 epow(E4 a, uint64_t e) {
  E4 r = eone();
  while (e) {
    if (e & 1) r = emul(r, a);
    a = esqr(a);
    e >>= 1;
  }
  return r;
}
KB_HD E4 epow2k(E4 a, int k) {
  for (int i = 0; i < k; i++) a = esqr(a);
  return a;
}

}  // namespace kb
