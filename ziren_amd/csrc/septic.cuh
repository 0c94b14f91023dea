// The septic extension F_p[z] / (z^7 + 2z - 8) of KoalaBear (crates/stark/src/septic_extension.rs) and the curve
// y^2 = x^3 + 3z x - 3 over it (crates/stark/src/septic_curve.rs) that the Global chip accumulates its messages on; every value in
// Montgomery form. Frobenius is linear over F_p: x^p = sum_i x_i (z^i)^p with the six (z^i)^p in constant memory, computed once on
// the host by exponentiation (upload_tables) — no tables copied from the reference. The square root follows the structure of
// septic_extension.rs:632-660: n^((p+1)/2 (p + p^3 + p^5) + 1) squares to n Norm(n), divided by a root of Norm(n) found in F_p
// (Tonelli-Shanks, p - 1 = 2^24 * 127); the inverse is the product of the six conjugates over the norm.
#pragma once
#include <hip/hip_runtime.h>
#include "kb31.cuh"

namespace septic {

struct S7 { uint32_t c[7]; };
typedef uint32_t FrobTable[6][7];
__constant__ FrobTable d_frob;
static FrobTable h_frob;

KB_HD S7 s_zero() { S7 r; for (int i = 0; i < 7; i++) r.c[i] = 0; return r; }
KB_HD bool s_is_zero(const S7& a) { uint32_t v = 0; for (int i = 0; i < 7; i++) v |= a.c[i]; return v == 0; }
KB_HD S7 s_add(const S7& a, const S7& b) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::add(a.c[i], b.c[i]); return r; }
KB_HD S7 s_sub(const S7& a, const S7& b) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::sub(a.c[i], b.c[i]); return r; }
KB_HD S7 s_neg(const S7& a) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::neg(a.c[i]); return r; }
KB_HD S7 s_scale(const S7& a, uint32_t k) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::mul(a.c[i], k); return r; }
KB_HD S7 s_mul(const S7& a, const S7& b) {
  uint32_t t[13];
#pragma unroll
  for (int k = 0; k < 13; k++) t[k] = 0;
#pragma unroll
  for (int i = 0; i < 7; i++)
#pragma unroll
    for (int j = 0; j < 7; j++) t[i + j] = kb::add(t[i + j], kb::mul(a.c[i], b.c[j]));
#pragma unroll
  for (int k = 12; k >= 7; k--) {   // z^k = z^(k-7) (8 - 2z)
    const uint32_t two = kb::dbl(t[k]);
    t[k - 7] = kb::add(t[k - 7], kb::dbl(kb::dbl(two)));
    t[k - 6] = kb::sub(t[k - 6], two);
  }
  S7 r;
#pragma unroll
  for (int i = 0; i < 7; i++) r.c[i] = t[i];
  return r;
}
KB_HD S7 s_pow(S7 x, uint32_t e) {
  S7 r = s_zero();
  r.c[0] = kb::ONE;
  while (e) {
    if (e & 1) r = s_mul(r, x);
    x = s_mul(x, x);
    e >>= 1;
  }
  return r;
}
KB_HD S7 s_frob(const S7& x, const FrobTable& tab) {
  S7 r = s_zero();
  r.c[0] = x.c[0];
#pragma unroll
  for (int i = 1; i < 7; i++)
#pragma unroll
    for (int k = 0; k < 7; k++) r.c[k] = kb::add(r.c[k], kb::mul(x.c[i], tab[i - 1][k]));
  return r;
}
// product of the six conjugates x^(p + ... + p^6) and the norm x * that (an element of F_p)
KB_HD void s_norm(const S7& x, const FrobTable& tab, S7* conj, uint32_t* norm) {
  S7 f = s_frob(x, tab), acc = f;
  for (int i = 2; i <= 6; i++) {
    f = s_frob(f, tab);
    acc = s_mul(acc, f);
  }
  *conj = acc;
  *norm = s_mul(acc, x).c[0];
}
KB_HD S7 s_inv(const S7& x, const FrobTable& tab) {
  S7 cj;
  uint32_t n;
  s_norm(x, tab, &cj, &n);
  return s_scale(cj, kb::inv(n));
}
KB_HD uint32_t f_pow(uint32_t a, uint32_t e) {
  uint32_t r = kb::ONE;
  while (e) {
    if (e & 1) r = kb::mul(r, a);
    a = kb::sqr(a);
    e >>= 1;
  }
  return r;
}
KB_HD bool f_is_square(uint32_t a) { return f_pow(a, (kb::P - 1) / 2) == kb::ONE; }   // a != 0
KB_HD uint32_t f_sqrt(uint32_t a) {   // a is a nonzero square; Tonelli-Shanks with the non-residue 3
  uint32_t m = 24, c = f_pow(kb::GEN, 127), t = f_pow(a, 127), r = f_pow(a, 64);
  while (t != kb::ONE) {
    uint32_t i = 0, tt = t;
    while (tt != kb::ONE) { tt = kb::sqr(tt); i++; }
    uint32_t b = c;
    for (uint32_t k = 0; k + i + 1 < m; k++) b = kb::sqr(b);
    m = i;
    c = kb::sqr(b);
    t = kb::mul(t, c);
    r = kb::mul(r, b);
  }
  return r;
}
// n has a square root iff its norm is a nonzero square of F_p (n not in {0, 1}: lift_x skips those, their roots have y6 = 0)
KB_HD bool s_is_square(const S7& n, const FrobTable& tab, uint32_t* norm) {
  S7 cj;
  s_norm(n, tab, &cj, norm);
  return *norm != 0 && f_is_square(*norm);
}
// a square root of n, which has one and whose norm is `norm`
KB_HD S7 s_sqrt(const S7& n, uint32_t norm, const FrobTable& tab) {
  const S7 t = s_pow(n, (kb::P + 1) / 2);
  const S7 f1 = s_frob(t, tab), f3 = s_frob(s_frob(f1, tab), tab), f5 = s_frob(s_frob(f3, tab), tab);
  const S7 d = s_mul(s_mul(s_mul(f1, f3), f5), n);
  return s_scale(d, f_sqrt(kb::inv(norm)));
}

struct Point { S7 x, y; uint32_t inf; };
KB_HD S7 curve_formula(const S7& x) {   // x^3 + 3z x - 3
  S7 three_z = s_zero();
  three_z.c[1] = kb::to_monty(3);
  S7 r = s_add(s_mul(s_mul(x, x), x), s_mul(x, three_z));
  r.c[0] = kb::sub(r.c[0], kb::to_monty(3));
  return r;
}
// SepticCurveComplete's addition (septic_curve.rs:198-216): the identity, opposite points and doubling are handled, which a parallel
// scan needs (a partial sum may hold a message and its own removal)
KB_HD Point add_complete(const Point& a, const Point& b, const FrobTable& tab) {
  if (a.inf) return b;
  if (b.inf) return a;
  const S7 dx = s_sub(b.x, a.x);
  S7 slope;
  if (s_is_zero(dx)) {
    if (s_is_zero(s_add(a.y, b.y))) { Point r = a; r.inf = 1; return r; }
    S7 num = s_mul(a.x, a.x);
    num = s_add(s_add(num, num), num);
    num.c[1] = kb::add(num.c[1], kb::to_monty(3));
    slope = s_mul(num, s_inv(s_add(a.y, a.y), tab));
  } else {
    slope = s_mul(s_sub(b.y, a.y), s_inv(dx, tab));
  }
  Point r;
  r.x = s_sub(s_sub(s_mul(slope, slope), a.x), b.x);
  r.y = s_sub(s_mul(slope, s_sub(a.x, r.x)), a.y);
  r.inf = 0;
  return r;
}
KB_HD S7 sum_checker_x(const Point& p1, const Point& p2, const Point& p3) {   // septic_curve.rs:159-166
  const S7 dx = s_sub(p2.x, p1.x), dy = s_sub(p2.y, p1.y);
  return s_sub(s_mul(s_add(s_add(p1.x, p2.x), p3.x), s_mul(dx, dx)), s_mul(dy, dy));
}
// SepticCurve::lift_x (septic_curve.rs:126-154): the first offset in 0..255 whose x = (m0, .., m5, 256 m6 + offset) carries a point
// with y6 != 0; y in the half 1 <= y6 <= (p - 1) / 2 (the "receive" sign). false: no offset works (probability 2^-256).
// The search for the offset (a norm and an Euler criterion per candidate) runs first and the one square root after it: lanes of a
// wave need different numbers of candidates, and a root taken inside the search loop would be serialised once per distinct count.
KB_HD bool lift_x(const S7& m, const FrobTable& tab, Point* out, uint32_t* offset) {
  const uint32_t m6 = kb::mul(m.c[6], kb::to_monty(256));
  uint32_t off = 0;
  while (off < 256) {
    S7 x = m, n;
    uint32_t norm = 0;
    for (; off < 256; off++) {
      x.c[6] = kb::add(m6, kb::to_monty(off));
      n = curve_formula(x);
      if (s_is_square(n, tab, &norm)) break;
    }
    if (off == 256) return false;
    S7 y = s_sqrt(n, norm, tab);
    const uint32_t y6 = kb::from_monty(y.c[6]);
    if (y6 == 0) { off++; continue; }   // is_exception (septic_extension.rs:696): next candidate
    if (y6 >= (kb::P + 1) / 2) y = s_neg(y);
    out->x = x; out->y = y; out->inf = 0;
    *offset = off;
    return true;
  }
  return false;
}

inline hipError_t upload_tables() {
  S7 z = s_zero();
  z.c[1] = kb::ONE;
  const S7 zp = s_pow(z, kb::P);
  S7 cur = zp;
  for (int i = 0; i < 6; i++) {
    for (int k = 0; k < 7; k++) h_frob[i][k] = cur.c[k];
    cur = s_mul(cur, zp);
  }
  return hipMemcpyToSymbol(HIP_SYMBOL(d_frob), h_frob, sizeof h_frob);
}

}  // namespace septic
