// Big-field arithmetic for the precompile chips whose rows hold 256-bit field elements as byte limbs (crates/core/machine/src/operations/field/:
// the reference computes them with num::BigUint on the host). Numbers are little-endian 32-bit limbs; a modulus comes with its Barrett constant
// mu = floor(2^(64 NL) / p); a product is reduced with two multiplications and at most two subtractions, an inverse is a^(p - 2). Every loop over limbs is
// unrolled and no limb array is indexed by a run-time value, so the numbers live in registers (the 12-limb kernels spilled 336 bytes per lane before).
// The polynomial side of a gadget — the witness of op(x) - result(x) - carry(x) p(x) = (x - 256) w(x) over byte limbs — is in tracegen.cuh.
#pragma once
#include <cstdint>

namespace bigfield {

template <int NL> struct Modulus { uint32_t p[NL]; uint32_t mu[NL + 1]; };

template <int NA, int NB> __device__ __forceinline__ void mul(const uint32_t* a, const uint32_t* b, uint32_t* out /* NA + NB */) {
#pragma unroll
  for (int i = 0; i < NA + NB; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = t >> 32;
    }
    out[i + NB] = (uint32_t)carry;
  }
}
template <int N> __device__ __forceinline__ int cmp(const uint32_t* a, const uint32_t* b) {
  int order = 0;      // branch-free over the limbs, so that they stay in registers
#pragma unroll
  for (int i = 0; i < N; i++) order = a[i] != b[i] ? (a[i] < b[i] ? -1 : 1) : order;
  return order;
}
template <int N> __device__ __forceinline__ uint32_t add(uint32_t* a, const uint32_t* b) {     // a += b, returns the carry out
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < N; i++) { const uint64_t t = (uint64_t)a[i] + b[i] + carry; a[i] = (uint32_t)t; carry = t >> 32; }
  return (uint32_t)carry;
}
template <int N> __device__ __forceinline__ uint32_t sub(uint32_t* a, const uint32_t* b) {     // a -= b, returns the borrow out
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < N; i++) { const uint64_t t = (uint64_t)a[i] - b[i] - borrow; a[i] = (uint32_t)t; borrow = (t >> 32) & 1; }
  return (uint32_t)borrow;
}
// x (2 NL limbs, below 2^(64 NL)) = q * p + r with 0 <= r < p: Barrett (HAC 14.42) in base 2^32
template <int NL> __device__ __forceinline__ void divmod(const uint32_t* x, const Modulus<NL>& m, uint32_t* q /* NL + 1 */, uint32_t* r /* NL */) {
  uint32_t q2[2 * NL + 2];
  mul<NL + 1, NL + 1>(x + NL - 1, m.mu, q2);            // floor(x / b^(NL-1)) * mu
  for (int i = 0; i <= NL; i++) q[i] = q2[NL + 1 + i];   // / b^(NL+1): the estimate, at most 2 below the quotient
  uint32_t qp[2 * NL + 1], rem[2 * NL + 1];
  mul<NL + 1, NL>(q, m.p, qp);
  for (int i = 0; i < 2 * NL; i++) rem[i] = x[i];
  rem[2 * NL] = 0;
  sub<2 * NL + 1>(rem, qp);
  uint32_t pw[NL + 1];
  for (int i = 0; i < NL; i++) pw[i] = m.p[i];
  pw[NL] = 0;
  const uint32_t one[NL + 1] = {1};
  for (int k = 0; k < 3 && cmp<NL + 1>(rem, pw) >= 0; k++) { sub<NL + 1>(rem, pw); add<NL + 1>(q, one); }
  for (int i = 0; i < NL; i++) r[i] = rem[i];
}
template <int NL> __device__ __forceinline__ void mulmod(const uint32_t* a, const uint32_t* b, const Modulus<NL>& m, uint32_t* out) {
  uint32_t t[2 * NL], q[NL + 1];
  mul<NL, NL>(a, b, t);
  divmod<NL>(t, m, q, out);
}
// a^e mod p, a in [0, p), e > 0: left to right over two-bit digits of the exponent with a, a^2, a^3 at hand — the exponents here (p - 2,
// (p + 1) / 4, (p + 3) / 8 of pseudo-Mersenne primes) are mostly ones, where the plain binary method multiplies at every bit
template <int NL> __device__ __forceinline__ void pow(const uint32_t* a, const uint32_t* e, const Modulus<NL>& m, uint32_t* out) {
  uint32_t a2[NL], a3[NL], acc[NL];
  mulmod<NL>(a, a, m, a2);
  mulmod<NL>(a2, a, m, a3);
  bool started = false;
  for (int digit = 16 * NL - 1; digit >= 0; digit--) {
    uint32_t word = 0;      // e[digit / 16] without indexing the array dynamically (that would put it in scratch memory)
#pragma unroll
    for (int l = 0; l < NL; l++) word = l == digit / 16 ? e[l] : word;
    const uint32_t d = (word >> (2 * (digit % 16))) & 3;
    if (started) { mulmod<NL>(acc, acc, m, acc); mulmod<NL>(acc, acc, m, acc); }
    if (d) {
      uint32_t f[NL];
#pragma unroll
      for (int i = 0; i < NL; i++) f[i] = d == 1 ? a[i] : d == 2 ? a2[i] : a3[i];
      if (started) mulmod<NL>(acc, f, m, acc);
      else {
#pragma unroll
        for (int i = 0; i < NL; i++) acc[i] = f[i];
        started = true;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NL; i++) out[i] = acc[i];
}
// a^(p - 2) mod p, a in [1, p) (p odd, p > 2)
template <int NL> __device__ __forceinline__ void inverse(const uint32_t* a, const Modulus<NL>& m, uint32_t* out) {
  uint32_t e[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) e[i] = m.p[i];
  const uint32_t two[NL] = {2};
  sub<NL>(e, two);
  pow<NL>(a, e, m, out);
}
// The same two out of line: for the twelve-limb fields, whose kernels are at the register limit, the long dependent chain of an inverse or
// a root is better off with a register allocation of its own than inlined into every field operation of its caller
template <int NL> __device__ __noinline__ void inverse_call(const uint32_t* a, const Modulus<NL>* m, uint32_t* out) {
  uint32_t x[NL], y[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) x[i] = a[i];
  inverse<NL>(x, *m, y);
#pragma unroll
  for (int i = 0; i < NL; i++) out[i] = y[i];
}
template <int NL> __device__ __noinline__ void pow_call(const uint32_t* a, const uint32_t* e, const Modulus<NL>* m, uint32_t* out) {
  uint32_t x[NL], ex[NL], y[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) { x[i] = a[i]; ex[i] = e[i]; }
  pow<NL>(x, ex, *m, y);
#pragma unroll
  for (int i = 0; i < NL; i++) out[i] = y[i];
}
__device__ __forceinline__ uint32_t byte_of(const uint32_t* a, int i) { return (a[i / 4] >> (8 * (i % 4))) & 0xff; }

}  // namespace bigfield
