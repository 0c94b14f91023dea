// TEST INFRASTRUCTURE ONLY (nothing shipped links or calls this): CPU restatement of the septic extension and curve behind the
// reference's global lookups — F_{p^7} = F_p[z] / (z^7 + 2z - 8) (crates/stark/src/septic_extension.rs:1-16) and the curve
// y^2 = x^3 + 3z x - 3 over it (crates/stark/src/septic_curve.rs:1-7, curve_formula :100-122, lift_x :126-154, add_incomplete :53-58,
// sum_checker_x / _y :159-176, dummy point :18-38); is_receive / is_send / is_exception: septic_extension.rs:683-698. The algorithms are
// generic (schoolbook product, x^p by exponentiation, Tonelli-Shanks in F_p) rather than the reference's table-driven Frobenius.
#pragma once
#include <array>
#include <stdexcept>
#include "field.hpp"

namespace septic {
using namespace orc;

struct S7 { F c[7]; };
static inline S7 s_zero() { S7 r{}; return r; }
static inline S7 s_base(F a) { S7 r{}; r.c[0] = a; return r; }
static inline bool s_eq(const S7& a, const S7& b) { for (int i = 0; i < 7; i++) if (a.c[i] != b.c[i]) return false; return true; }
static inline S7 s_add(const S7& a, const S7& b) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = fadd(a.c[i], b.c[i]); return r; }
static inline S7 s_sub(const S7& a, const S7& b) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = fsub(a.c[i], b.c[i]); return r; }
static inline S7 s_neg(const S7& a) { return s_sub(s_zero(), a); }
static inline S7 s_scale(const S7& a, F k) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = fmul(a.c[i], k); return r; }
static inline S7 s_mul(const S7& a, const S7& b) {
  F t[13] = {0};
  for (int i = 0; i < 7; i++)
    for (int j = 0; j < 7; j++) t[i + j] = fadd(t[i + j], fmul(a.c[i], b.c[j]));
  for (int k = 12; k >= 7; k--) {   // z^k = z^(k-7) * (8 - 2z)
    t[k - 7] = fadd(t[k - 7], fmul(t[k], 8));
    t[k - 6] = fsub(t[k - 6], fmul(t[k], 2));
  }
  S7 r;
  for (int i = 0; i < 7; i++) r.c[i] = t[i];
  return r;
}
static inline S7 s_pow(S7 x, uint64_t e) {
  S7 r = s_base(1);
  while (e) { if (e & 1) r = s_mul(r, x); x = s_mul(x, x); e >>= 1; }
  return r;
}
static inline S7 s_frob(const S7& x) { return s_pow(x, P); }
// the norm to F_p and the product of the six conjugates (x^(p + ... + p^6)), from which the inverse follows
static inline void s_norm(const S7& x, S7* conj, F* norm) {
  S7 f = s_frob(x), acc = f;
  for (int i = 2; i <= 6; i++) { f = s_frob(f); acc = s_mul(acc, f); }
  S7 n = s_mul(acc, x);
  for (int i = 1; i < 7; i++) if (n.c[i] != 0) throw std::runtime_error("septic: norm is not in the base field");
  *conj = acc; *norm = n.c[0];
}
static inline S7 s_inv(const S7& x) { S7 cj; F n; s_norm(x, &cj, &n); return s_scale(cj, finv(n)); }
static inline F fpow(F a, uint64_t e) { F r = 1; while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; } return r; }
static inline bool f_sqrt(F a, F* out) {   // Tonelli-Shanks, p - 1 = 2^24 * 127
  if (a == 0) { *out = 0; return true; }
  if (fpow(a, (P - 1) / 2) != 1) return false;
  F z = 2;
  while (fpow(z, (P - 1) / 2) == 1) z++;
  uint32_t m = 24;
  F c = fpow(z, 127), t = fpow(a, 127), r = fpow(a, 64);
  while (t != 1) {
    uint32_t i = 0; F tt = t;
    while (tt != 1) { tt = fmul(tt, tt); i++; }
    F b = c;
    for (uint32_t k = 0; k + i + 1 < m; k++) b = fmul(b, b);
    m = i; c = fmul(b, b); t = fmul(t, c); r = fmul(r, b);
  }
  *out = r;
  return true;
}
static inline bool s_sqrt(const S7& n, S7* out) {
  if (s_eq(n, s_zero()) || s_eq(n, s_base(1))) { *out = n; return true; }
  S7 cj; F norm;
  s_norm(n, &cj, &norm);
  if (fpow(norm, (P - 1) / 2) != 1) return false;
  // n^((p+1)/2 * (p + p^3 + p^5) + 1) squared is n * Norm(n), so dividing by sqrt(Norm(n)) gives a root of n
  S7 t = s_pow(n, ((uint64_t)P + 1) / 2);
  S7 f1 = s_frob(t), f3 = s_frob(s_frob(f1)), f5 = s_frob(s_frob(f3));
  S7 d = s_mul(s_mul(s_mul(f1, f3), f5), n);
  F s;
  if (!f_sqrt(finv(norm), &s)) return false;
  S7 r = s_scale(d, s);
  if (!s_eq(s_mul(r, r), n)) throw std::runtime_error("septic: square root check failed");
  *out = r;
  return true;
}

struct Point { S7 x, y; };
static inline S7 curve_formula(const S7& x) {   // x^3 + 3z x - 3
  S7 three_z = s_zero(); three_z.c[1] = 3;
  return s_sub(s_add(s_mul(s_mul(x, x), x), s_mul(x, three_z)), s_base(3));
}
static inline Point add_incomplete(const Point& a, const Point& b) {
  S7 dx = s_sub(b.x, a.x);
  if (s_eq(dx, s_zero())) throw std::runtime_error("septic: addition of points with equal x");
  S7 slope = s_mul(s_sub(b.y, a.y), s_inv(dx));
  Point r;
  r.x = s_sub(s_sub(s_mul(slope, slope), a.x), b.x);
  r.y = s_sub(s_mul(slope, s_sub(a.x, r.x)), a.y);
  return r;
}
static inline S7 sum_checker_x(const Point& p1, const Point& p2, const Point& p3) {
  S7 dx = s_sub(p2.x, p1.x), dy = s_sub(p2.y, p1.y);
  return s_sub(s_mul(s_add(s_add(p1.x, p2.x), p3.x), s_mul(dx, dx)), s_mul(dy, dy));
}
// lift_x: the first offset in 0..255 for which x = (m0, .., m5, m6 * 256 + offset) is on the curve with y6 != 0; y normalised to the
// "receive" half (1 <= y6 <= (p - 1) / 2)
static inline Point lift_x(const S7& m, uint8_t* offset) {
  for (int off = 0; off < 256; off++) {
    S7 x = m;
    x.c[6] = fadd(fmul(m.c[6], 256), (F)off);
    S7 y;
    if (!s_sqrt(curve_formula(x), &y)) continue;
    if (y.c[6] == 0) continue;
    if (y.c[6] >= (P + 1) / 2) y = s_neg(y);
    *offset = (uint8_t)off;
    return Point{x, y};
  }
  throw std::runtime_error("septic: no curve point within 256 offsets");
}
static const uint32_t DUMMY_X[7] = {1706420302, 1319108093, 148224806, 26874985, 1766171812, 1645633948, 2028659224};
static const uint32_t DUMMY_Y[7] = {942390502, 1239997438, 458866455, 1843332012, 1309764648, 572807436, 74267719};

// SepticDigest's Sum (crates/stark/src/septic_digest.rs:60-76) and is_zero (:53-57): what Machine::verify evaluates over the shard
// proofs' global_cumulative_sums and the key's initial one (crates/stark/src/machine.rs:657-671). Each digest carries the offset
// `zero` (the cumulative-sum start point); the fold runs from a second fixed point so that no step adds equal x-coordinates.
static const uint32_t DIGEST_SUM_START_X[7] = {1656788302, 897965284, 874620737, 1581672598, 655804282, 1962911564, 80580607};
static const uint32_t DIGEST_SUM_START_Y[7] = {1024875409, 218609128, 1856341123, 583920580, 1274441611, 118766316, 81843042};
static inline Point p_neg(const Point& a) { return Point{a.x, s_neg(a.y)}; }
static inline Point digest_sum(const Point* digests, size_t n, const Point& zero) {
  Point start;
  for (int i = 0; i < 7; i++) { start.x.c[i] = DIGEST_SUM_START_X[i]; start.y.c[i] = DIGEST_SUM_START_Y[i]; }
  Point acc = start;
  for (size_t i = 0; i < n; i++) acc = add_incomplete(add_incomplete(acc, digests[i]), p_neg(zero));
  acc = add_incomplete(acc, zero);
  return add_incomplete(acc, p_neg(start));
}

}  // namespace septic
