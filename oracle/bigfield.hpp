// TEST INFRASTRUCTURE (see oracle/README or DESIGN.md section 1): big-field arithmetic for the restated field gadgets of
// crates/core/machine/src/operations/field/ (field_op.rs, field_inner_product.rs, field_den.rs), which the reference does with num::BigUint.
// Deliberately plain and different from the device code (csrc/bigfield.cuh: 32-bit limbs, Barrett reduction, Fermat inversion): numbers are
// little-endian byte vectors, division is binary long division, inversion is the binary extended Euclidean algorithm.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace bigfield {

using Big = std::vector<uint32_t>;   // little-endian bytes, each 0..255; trailing zeros allowed

static inline void trim(Big& a) { while (!a.empty() && a.back() == 0) a.pop_back(); }
static inline Big from_u64(uint64_t v) { Big a; while (v) { a.push_back(v & 0xff); v >>= 8; } return a; }
static inline Big from_words(const uint32_t* w, int n_words) {
  Big a(4 * n_words);
  for (int i = 0; i < 4 * n_words; i++) a[i] = (w[i / 4] >> (8 * (i % 4))) & 0xff;
  return a;
}
static inline Big from_bytes(const uint8_t* b, int n) { return Big(b, b + n); }
static inline int cmp(Big a, Big b) {
  trim(a); trim(b);
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  for (size_t i = a.size(); i-- > 0;)
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return 0;
}
static inline bool is_zero(Big a) { trim(a); return a.empty(); }
static inline Big add(const Big& a, const Big& b) {
  Big out(std::max(a.size(), b.size()) + 1, 0);
  uint32_t carry = 0;
  for (size_t i = 0; i < out.size(); i++) {
    const uint32_t s = (i < a.size() ? a[i] : 0) + (i < b.size() ? b[i] : 0) + carry;
    out[i] = s & 0xff; carry = s >> 8;
  }
  trim(out);
  return out;
}
static inline Big sub(const Big& a, const Big& b) {      // a >= b
  if (cmp(a, b) < 0) throw std::runtime_error("bigfield: negative difference");
  Big out(a.size(), 0);
  int borrow = 0;
  for (size_t i = 0; i < a.size(); i++) {
    int d = (int)a[i] - (int)(i < b.size() ? b[i] : 0) - borrow;
    borrow = d < 0;
    out[i] = (uint32_t)(d + 256 * borrow);
  }
  trim(out);
  return out;
}
static inline Big mul(const Big& a, const Big& b) {
  std::vector<uint64_t> acc(a.size() + b.size() + 1, 0);
  for (size_t i = 0; i < a.size(); i++)
    for (size_t j = 0; j < b.size(); j++) acc[i + j] += (uint64_t)a[i] * b[j];
  Big out(acc.size() + 8, 0);
  uint64_t carry = 0;
  for (size_t i = 0; i < out.size(); i++) {
    const uint64_t s = (i < acc.size() ? acc[i] : 0) + carry;
    out[i] = s & 0xff; carry = s >> 8;
  }
  trim(out);
  return out;
}
static inline bool bit(const Big& a, size_t k) { return k / 8 < a.size() && ((a[k / 8] >> (k % 8)) & 1); }
static inline void halve(Big& a) {
  uint32_t carry = 0;
  for (size_t i = a.size(); i-- > 0;) { const uint32_t v = a[i] | (carry << 8); a[i] = v >> 1; carry = v & 1; }
  trim(a);
}
// num = q * den + r, 0 <= r < den: one bit of the quotient at a time; the running remainder lives in eight 64-bit words (den below 2^448)
static inline void divmod(const Big& num, const Big& den_in, Big& q, Big& r) {
  Big den = den_in;
  trim(den);
  if (den.empty()) throw std::runtime_error("bigfield: division by zero");
  if (den.size() > 56) throw std::runtime_error("bigfield: divisor too wide");
  uint64_t d[8] = {0}, rem[8] = {0};
  for (size_t i = 0; i < den.size(); i++) d[i / 8] |= (uint64_t)den[i] << (8 * (i % 8));
  q.assign(num.size() + 1, 0);
  for (size_t k = 8 * num.size(); k-- > 0;) {
    for (int w = 7; w > 0; w--) rem[w] = (rem[w] << 1) | (rem[w - 1] >> 63);
    rem[0] = (rem[0] << 1) | (bit(num, k) ? 1u : 0u);
    bool ge = true;
    for (int w = 7; w >= 0; w--)
      if (rem[w] != d[w]) { ge = rem[w] > d[w]; break; }
    if (ge) {
      uint64_t borrow = 0;
      for (int w = 0; w < 8; w++) {
        const uint64_t x = rem[w], y = d[w];
        rem[w] = x - y - borrow;
        borrow = (x < y) || (x == y && borrow);
      }
      q[k / 8] |= 1u << (k % 8);
    }
  }
  r.assign(64, 0);
  for (size_t i = 0; i < 64; i++) r[i] = (rem[i / 8] >> (8 * (i % 8))) & 0xff;
  trim(r);
  trim(q);
}
static inline Big mod(const Big& a, const Big& p) { Big q, r; divmod(a, p, q, r); return r; }
// a^-1 mod p for an odd prime p and a not a multiple of p: binary extended Euclid, every intermediate kept in [0, p)
static inline Big inv_mod(const Big& a_in, const Big& p) {
  Big u = mod(a_in, p), v = p, x1 = from_u64(1), x2;
  if (is_zero(u)) throw std::runtime_error("bigfield: zero has no inverse");
  auto half_mod = [&](Big& x) { if (bit(x, 0)) x = add(x, p); halve(x); };
  auto sub_mod = [&](const Big& x, const Big& y) { return cmp(x, y) >= 0 ? sub(x, y) : sub(add(x, p), y); };
  const Big one = from_u64(1);
  while (cmp(u, one) != 0 && cmp(v, one) != 0) {
    while (!bit(u, 0)) { halve(u); half_mod(x1); }
    while (!bit(v, 0)) { halve(v); half_mod(x2); }
    if (cmp(u, v) >= 0) { u = sub(u, v); x1 = sub_mod(x1, x2); } else { v = sub(v, u); x2 = sub_mod(x2, x1); }
  }
  return cmp(u, one) == 0 ? x1 : x2;
}
static inline Big pow_mod(const Big& a, const Big& e, const Big& p) {      // right-to-left square and multiply
  Big acc = from_u64(1), base = mod(a, p);
  for (size_t k = 0; k < 8 * e.size(); k++) {
    if (bit(e, k)) acc = mod(mul(acc, base), p);
    base = mod(mul(base, base), p);
  }
  return acc;
}
static inline uint32_t limb(const Big& a, size_t i) { return i < a.size() ? a[i] : 0; }

// coefficients of a byte polynomial product / sum, as signed 64-bit integers (|.| < 2^24 for every gadget here)
using Poly = std::vector<int64_t>;
static inline Poly poly(const Big& a, size_t n) { Poly p(n, 0); for (size_t i = 0; i < n; i++) p[i] = limb(a, i); return p; }
static inline Poly pmul(const Poly& a, const Poly& b) {
  Poly out(a.size() + b.size() - 1, 0);
  for (size_t i = 0; i < a.size(); i++)
    for (size_t j = 0; j < b.size(); j++) out[i + j] += a[i] * b[j];
  return out;
}
static inline Poly padd(Poly a, const Poly& b, int sign = 1) {
  if (a.size() < b.size()) a.resize(b.size(), 0);
  for (size_t i = 0; i < b.size(); i++) a[i] += sign * b[i];
  return a;
}

}  // namespace bigfield
