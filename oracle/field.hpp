// TEST INFRASTRUCTURE — CPU oracle. Never linked, imported or executed by the product path.
//
// KoalaBear base field and its degree-4 binomial extension, restated in *canonical*
// integer arithmetic (plain `% p`), deliberately not Montgomery, so that agreement with
// the HIP path (which is Montgomery end to end) is agreement between two independent
// implementations.
//
// Follows:
//   p = 2^31 - 2^24 + 1, Montgomery R = 2^32      crates/core/machine/include/kb31_t.hpp:458-503
//   generator 3, two-adicity 24                    crates/recursion/compiler/src/ir/utils.rs:10-12
//   EF = F[X]/(X^4 - 3), coefficients low→high     crates/stark/src/air/extension.rs:55-74
// Pinned by tests/golden/kb31_field_kat.json (generated from the reference's kb31_t.hpp).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

static const uint32_t P = 0x7f000001u;
static const uint32_t GENERATOR = 3;
static const int TWO_ADICITY = 24;

typedef uint32_t F;  // canonical representative in [0, P)

static inline F fadd(F a, F b) { uint32_t s = a + b; return s >= P ? s - P : s; }
static inline F fsub(F a, F b) { return a >= b ? a - b : a + P - b; }
static inline F fneg(F a) { return a ? P - a : 0; }
static inline F fmul(F a, F b) { return (F)(((uint64_t)a * b) % P); }
static inline F fpow(F a, uint64_t e) {
  F r = 1;
  while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; }
  return r;
}
static inline F finv(F a) { assert(a != 0); return fpow(a, P - 2); }

// Montgomery boundary conversions (the ABI carries Montgomery words, kb31_t.hpp:487-503).
static inline F from_monty(uint32_t m) {
  static const F RINV = finv((F)((1ull << 32) % P));
  return fmul(m % P, RINV);
}
static inline uint32_t to_monty(F a) { return (uint32_t)((((uint64_t)a) << 32) % P); }

// omega_k = 3^((p-1)/2^k): generator of the order-2^k subgroup.
static inline F two_adic_generator(int k) {
  assert(k <= TWO_ADICITY);
  return fpow(GENERATOR, (uint64_t)(P - 1) >> k);
}

struct E {
  F c[4];
  bool operator==(const E& o) const { return c[0]==o.c[0] && c[1]==o.c[1] && c[2]==o.c[2] && c[3]==o.c[3]; }
  bool operator!=(const E& o) const { return !(*this == o); }
};
static const uint32_t W = 3;  // X^4 = W

static inline E ezero() { return E{{0,0,0,0}}; }
static inline E eone() { return E{{1,0,0,0}}; }
static inline E efrom(F a) { return E{{a,0,0,0}}; }
static inline bool eis_zero(const E& a) { return !(a.c[0] | a.c[1] | a.c[2] | a.c[3]); }
static inline E eadd(const E& a, const E& b) { E r; for (int i=0;i<4;i++) r.c[i]=fadd(a.c[i],b.c[i]); return r; }
static inline E esub(const E& a, const E& b) { E r; for (int i=0;i<4;i++) r.c[i]=fsub(a.c[i],b.c[i]); return r; }
static inline E eneg(const E& a) { E r; for (int i=0;i<4;i++) r.c[i]=fneg(a.c[i]); return r; }
static inline E escale(const E& a, F s) { E r; for (int i=0;i<4;i++) r.c[i]=fmul(a.c[i],s); return r; }
// schoolbook product reduced by X^4 = 3 (extension.rs:58-74)
static inline E emul(const E& a, const E& b) {
  E r = ezero();
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      F t = fmul(a.c[i], b.c[j]);
      if (i + j >= 4) r.c[i + j - 4] = fadd(r.c[i + j - 4], fmul(W, t));
      else r.c[i + j] = fadd(r.c[i + j], t);
    }
  return r;
}
// Frobenius x -> x^p acts on the basis as X^i -> z^i X^i with z = W^((p-1)/4).
static inline E efrob(const E& a) {
  static const F z = fpow(W, (P - 1) / 4);
  E r; F zi = 1;
  for (int i = 0; i < 4; i++) { r.c[i] = fmul(a.c[i], zi); zi = fmul(zi, z); }
  return r;
}
// a^{-1} = (a^p a^{p^2} a^{p^3}) / Norm(a), Norm(a) = a * that product, lies in F.
static inline E einv(const E& a) {
  assert(!eis_zero(a));
  E a1 = efrob(a), a2 = efrob(a1), a3 = efrob(a2);
  E prod = emul(emul(a1, a2), a3);
  E norm = emul(a, prod);
  assert(norm.c[1] == 0 && norm.c[2] == 0 && norm.c[3] == 0);
  return escale(prod, finv(norm.c[0]));
}
static inline E ediv(const E& a, const E& b) { return emul(a, einv(b)); }
static inline E epow(E a, uint64_t e) {
  E r = eone();
  while (e) { if (e & 1) r = emul(r, a); a = emul(a, a); e >>= 1; }
  return r;
}
static inline E epow2k(E a, int k) { for (int i = 0; i < k; i++) a = emul(a, a); return a; }

static inline uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
static inline int log2_strict(size_t n) {
  int k = 0; while (((size_t)1 << k) < n) k++;
  assert(((size_t)1 << k) == n);
  return k;
}

}  // namespace orc
