// Poseidon2 width-16 over KoalaBear on the FP64 vector pipe of gfx950.
//
// Same permutation as poseidon2.cuh (zkm_primitives::poseidon2_init, crates/primitives/src/lib.rs:1107-1122;
// layers crates/recursion/core/include/poseidon2.hpp:21-71), bit for bit, but the state is sixteen *doubles* holding
// exact integers congruent to the canonical (non-Montgomery) field values. Why: on MI355X v_add/mul/fma/rndne_f64
// issue at the same rate as the 32-bit integer multiplies (4.3-4.4 cycles per wave64 instruction,
// profiles/r02_ubench_valu.txt), and a double has 53 bits where an int32 has 31:
//   * the external MDS layer (coefficient sum 35) is 64 plain additions with no reduction at all
//     (the integer version: ~70 modular additions of three instructions each);
//   * a modular product a b - q p is four instructions (mulmod_q below: the quotient by the magic-number rounding, q (p - 1) exact
//     because p - 1 = 127 * 2^24, the residue recovered next to 1.5 * 2^52), no carries; a cube is nine;
//   * the internal layer's diagonal (+-2^-k, small integers) is one fma per lane: the lanes with a 2^-k entry stay dyadic
//     rationals (exact in a double) between the rounds and are made integers again only every few rounds.
// Every operation below is exact integer arithmetic as long as the stated magnitude bounds hold; they are
// re-derived next to each step and hammered by tests/test_host_abi.py::test_fp64_poseidon2_* (host build of this very code, which
// uses the same IEEE operations: plain words, unreduced sponges and compress-with-injection chains on adversarial states, with the
// magnitudes below *measured* by the probes of the host build and held against the documented bounds) and by the GPU parity
// tests against the integer version and the oracle.
//
// Interface: load_monty (u32 Montgomery word -> canonical double in (-p, 0]), permute, store_monty (-> [0, p) Montgomery).
#pragma once
#include "poseidon2.cuh"

namespace p2f {

constexpr double P = 2130706433.0;
constexpr double PINV = 1.0 / 2130706433.0;  // RN(1/p): relative error <= 2^-53

KB_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
KB_HD double rne(double x) { return __builtin_rint(x); }  // v_rndne_f64 (round to nearest even, the default mode on the host too)
// A constant that is not one of the inline operands (+-0.5, 1, 2, 4), held in a scalar register pair. As a literal it can only be
// encoded in the two-address v_fmac_f64, whose addend is overwritten — and the addend of the internal layer's lanes is the lane sum,
// which every lane needs: the compiler copies it first (v_mov_b64, nine per partial round). An opaque scalar operand makes it the
// three-address v_fma_f64 instead.
KB_HD double sconst(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(c));
#endif
  return c;
}
// ... and one held in a vector register pair: an instruction of this family reads at most one scalar operand, so the second constant of
// fma(x, p - 1, -M p) sits in VGPRs (as a literal it would again be v_fmac_f64 with a v_mov_b64 of the addend in front).
KB_HD double vconst(double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(c));
#endif
  return c;
}

// Host build only: the largest magnitudes seen at the points the exactness argument rests on (a permutation's inputs, the lane sum
// of a partial round before its reduction, an S-box input, any lane after a partial round, the sum of the fractional lanes), and the
// number of fractional-lane operations that lost a bit (must be zero). The device build compiles them away.
struct Audit { double in = 0, lane_sum = 0, sbox_in = 0, lane = 0, frac_sum = 0, inexact = 0, sbox_fast_in = 0; };
inline Audit& audit() { static thread_local Audit a; return a; }
inline void probe(double& slot, double v) { v = v < 0 ? -v : v; if (v > slot) slot = v; }
#if !defined(__HIP_DEVICE_COMPILE__)
#define P2F_PROBE(slot, v) probe(audit().slot, (v))
#else
#define P2F_PROBE(slot, v) ((void)0)
#endif

// canonical tables as balanced doubles (filled by upload_tables / the host mirror below)
__constant__ double d_rc_ext[8][16];
__constant__ double d_rc_int[13];

// x integer, |x| < 2^52  ->  congruent integer with |r| <= p/2 + p*2^-20
KB_HD double reduce(double x) {
  const double q = rne(x * PINV);
  return fma_(-q, P, x);
}
// ---- the four-instruction modular product (round 4) ----
// a b - q p for the integer q nearest to a * bp, bp = RN(b / p), without ever forming the product's high part:
//   qm = fma(a, bp, M)         M = 1.5 * 2^52: the sum lands in [2^52, 2^53) where a double's ulp is 1, so qm = M + q exactly, q the
//                              integer nearest to a*bp (|q| < 2^51); |q - ab/p| <= 1/2 + |ab/p| 2^-52
//   tm = fma(qm, p - 1, -M p)  = (M + q)(p - 1) - M p = q (p - 1) - M exactly: p - 1 = 127 * 2^24, so q (p - 1) is 127 q shifted —
//                              representable while |127 q| < 2^53, i.e. |q| < 2^46 — and M is a multiple of 2^24 as well;
//                              M p = 3 p * 2^51 is a double (3 p has 33 bits)
//   rm = fma(a, b, -tm)        = (ab - q p) + q + M exactly: an integer in [2^52, 2^53) again (|ab - q p| <= p (1/2 + 2^-6), |q| < 2^46)
//   rm - qm                    = ab - q p exactly.
// Bounds: |ab| < 2^77 (so |q| < 2^46); result |r| <= p (1/2 + |ab/p| 2^-52) — for |ab| < 2^66.4 (an S-box's second product) that is
// p/2 + 2^15. The quotient's multiplier bp is shared by the two products of a cube (both multiply by y), so an S-box is
// 1 + 4 + 4 = nine instructions instead of twelve. The host build checks every such product against 128-bit integers.
constexpr double MAGIC = 6755399441055744.0;        // 1.5 * 2^52
constexpr double PM1 = 2130706432.0;                 // p - 1 = 127 * 2^24
constexpr double MAGIC_P = MAGIC * P;                // exact: 3 p * 2^51
KB_HD double mulmod_q(double a, double b, double bp) {
  const double qm = fma_(a, bp, sconst(MAGIC));
  const double tm = fma_(qm, sconst(PM1), vconst(-MAGIC_P));
  const double rm = fma_(a, b, -tm);
  const double r = rm - qm;
#if !defined(__HIP_DEVICE_COMPILE__)
  {
    const __int128 ab = (__int128)(long long)a * (__int128)(long long)b, ri = (__int128)(long long)r;
    const double lim = P * (0.5 + 1.0 / 32);
    if ((double)(long long)a != a || (double)(long long)b != b || (ab - ri) % (__int128)2130706433LL != 0 || r > lim || r < -lim ||
        (a < 0 ? -a : a) * (b < 0 ? -b : b) >= 0x1p77)
      audit().inexact += 1;
  }
#endif
  return r;
}
// (y)^3 for |y| < 2^38.5 (y^2 < 2^77): every S-box but the first sixteen of a permutation (steady state |y| <= 2^37.2).
// |z| <= p (1/2 + 2^-6) < 2^30.05, |z y| < 2^68.6; for |y| <= 2^36.3: |w| <= p/2 + 2^15.4.
KB_HD double sbox(double y) {
  P2F_PROBE(sbox_fast_in, y);
  const double yp = y * PINV;
  const double z = mulmod_q(y, y, yp);
  return mulmod_q(z, y, yp);
}

// The same for |y| < 2^40.6 — the first sixteen S-boxes of a permutation, which sit behind two linear layers in a row (inputs up to
// 2^35.3, first layer x 35). y^2 / p reaches 2^50.2, too many bits for 127 q; so the first product takes a quotient that is a multiple
// of 32 (the magic constant one binade up per factor of two: at 1.5 * 2^57 a double's ulp is 32) and leaves a residue below 16 p:
//   Qm = fma(y, yp, M32)    = M32 + Q, Q the multiple of 32 nearest to y yp; |Q - y^2 / p| <= 16 + 2^-2
//   Q  = Qm - M32           exact
//   tm = Q (p - 1)          exact: 127 Q = 32 * (127 Q / 32) with |127 Q / 32| < 2^52.2
//   rm = fma(y, y, -tm)     = (y^2 - Q p) + Q exactly: an integer below 2^51
//   z  = rm - Q             = y^2 - Q p, |z| <= p (16 + 2^-2) < 2^35.03
// five instructions with scalar constants only (the four-instruction form would want a second constant in vector registers, and
// compress_layer sits at 128 of them); then z y (|z y| < 2^75.7 < 2^77) is an ordinary four-instruction product: ten instead of twelve.
constexpr double MAGIC32 = 216172782113783808.0;                  // 1.5 * 2^57
KB_HD double sbox_wide(double y) {
  P2F_PROBE(sbox_in, y);
  const double yp = y * PINV;
  const double m32 = sconst(MAGIC32);
  const double Qm = fma_(y, yp, m32);
  const double Q = Qm - m32;
  const double tm = Q * sconst(PM1);
  const double rm = fma_(y, y, -tm);
  const double z = rm - Q;
#if !defined(__HIP_DEVICE_COMPILE__)
  {
    const __int128 yy = (__int128)(long long)y * (__int128)(long long)y, zi = (__int128)(long long)z;
    const double lim = P * 16.5;
    if ((double)(long long)y != y || (yy - zi) % (__int128)2130706433LL != 0 || z > lim || z < -lim || (y < 0 ? -y : y) >= 1.6668e12) audit().inexact += 1;
  }
#endif
  return mulmod_q(z, y, yp);
}

KB_HD void m4(double& s0, double& s1, double& s2, double& s3) {
  // [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
  const double t01 = s0 + s1, t23 = s2 + s3;
  const double t0123 = t01 + t23;
  const double t01123 = t0123 + s1, t01233 = t0123 + s3;
  const double n3 = fma_(2.0, s0, t01233);
  const double n1 = fma_(2.0, s2, t01123);
  const double n0 = t01123 + t01;
  const double n2 = t01233 + t23;
  s0 = n0; s1 = n1; s2 = n2; s3 = n3;
}
// every output is a combination of the inputs with non-negative coefficients summing to 35
KB_HD void external_layer(double s[16]) {
#pragma unroll
  for (int i = 0; i < 16; i += 4) m4(s[i], s[i + 1], s[i + 2], s[i + 3]);
  const double c0 = (s[0] + s[4]) + (s[8] + s[12]);
  const double c1 = (s[1] + s[5]) + (s[9] + s[13]);
  const double c2 = (s[2] + s[6]) + (s[10] + s[14]);
  const double c3 = (s[3] + s[7]) + (s[11] + s[15]);
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    s[i] += c0; s[i + 1] += c1; s[i + 2] += c2; s[i + 3] += c3;
  }
}

// The internal layer's diagonal has seven entries +-2^-K (K = 1, 1, 8, 3, 8, 3, 4 for lanes 3, 6, 9, 10, 12, 13, 14). Z[1/2] -> F_p is a ring
// homomorphism (2 is invertible), so the real number x / 2^K + sum — exact in a double while its fractional bits fit under the integer
// ones — already *is* the lane's new value: those seven lanes are left as dyadic rationals (one fma per lane and round), and only made
// integers again ("integerize") when their fractional bits run out or an S-box is about to read them (round 4: before, every such lane
// was made an integer in every round, four instructions instead of one). What a round needs from them is their sum, and that is
// integerized once per round instead.
// w = t + lo with t = rint(w), lo = j / 2^f, |lo| <= 1/2, f <= 24; 2^-f = -(p-1)/2^f (mod p) because 2^f (p-1)/2^f = -1, so lo stands for
// -j (p-1)/2^f = -lo (p-1), an integer of magnitude <= (p-1)/2 (2^24 divides p - 1). |result| <= |w| + 1/2 + 2^30.
KB_HD double integerize(double w) {
  constexpr double PM1 = 2130706432.0;
  const double t = rne(w);
  const double lo = w - t;
  return fma_(-lo, PM1, t);
}
// scale * x + sum with scale = +-2^-K, x a dyadic rational, sum an integer; the host build checks that no bit is lost
KB_HD double frac_lane(double scale, double x, double sum) {
  const double w = fma_(scale, x, sum);
#if !defined(__HIP_DEVICE_COMPILE__)
  if ((long double)w != (long double)scale * (long double)x + (long double)sum) audit().inexact += 1;   // 64-bit significands: the right side is exact
#endif
  return w;
}

// s_i <- V_i s_i + sum(s), V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/8, 2^-24, -2^-8, -1/8, -1/16, -2^-24]
// The two 2^-24 entries are the small integers -+127 modulo p (p - 1 = 127 * 2^24), one fma each; those two lanes then grow by 2^7 per
// round and are reduced every other round (permute_impl).
KB_HD void internal_layer(double s[16]) {
  // integer lanes and fractional lanes are summed apart: the integer part may reach 2^50.4, the fractional one carries up to 16 bits
  // below the point and stays under 2^36 (bounds below)
  double si = ((s[0] + s[1]) + (s[2] + s[4])) + ((s[5] + s[7]) + (s[8] + s[11])) + s[15];
  const double sf = ((s[3] + s[6]) + (s[9] + s[10])) + ((s[12] + s[13]) + s[14]);
#if !defined(__HIP_DEVICE_COMPILE__)
  {
    double a = 0, f = 0;
    for (int i = 0; i < 16; i++) a += s[i] < 0 ? -s[i] : s[i];
    P2F_PROBE(lane_sum, a);  // sum of magnitudes: no order of the additions can exceed it
    const int fl[7] = {3, 6, 9, 10, 12, 13, 14};
    long double e = 0;
    bool fractional = false;
    for (int i : fl) { f += s[i] < 0 ? -s[i] : s[i]; e += (long double)s[i]; fractional |= s[i] != rne(s[i]); }
    if (fractional) P2F_PROBE(frac_sum, f);   // (in the first partial round they are still integers, below 7 * 2^35.2)
    if ((long double)sf != e) audit().inexact += 1;
  }
#endif
  double sum = reduce(si + integerize(sf));
  s[0] = fma_(-2.0, s[0], sum);
  s[1] = s[1] + sum;
  s[2] = fma_(2.0, s[2], sum);
  s[3] = frac_lane(0.5, s[3], sum);
  s[4] = fma_(sconst(3.0), s[4], sum);
  s[5] = fma_(4.0, s[5], sum);
  s[6] = frac_lane(-0.5, s[6], sum);
  s[7] = fma_(sconst(-3.0), s[7], sum);
  s[8] = fma_(-4.0, s[8], sum);
  s[9] = frac_lane(sconst(1.0 / 256), s[9], sum);
  s[10] = frac_lane(sconst(0.125), s[10], sum);
  s[11] = fma_(sconst(-127.0), s[11], sum);  // 2^-24 = -127 (mod p): 127 * 2^24 = p - 1
  s[12] = frac_lane(sconst(-1.0 / 256), s[12], sum);
  s[13] = frac_lane(sconst(-0.125), s[13], sum);
  s[14] = frac_lane(sconst(-0.0625), s[14], sum);
  s[15] = fma_(sconst(127.0), s[15], sum);   // -2^-24 = 127
}

// Magnitudes (B = 2^30 + 2^15 bounds an S-box output; inputs of a permutation: |s_i| <= 2^35.3):
//  first layer: <= 35 * 2^35.3 = 2^40.5 -> first S-boxes see |y| < 2^40.6 (fine for sbox, see there), then every full
//  round: S-box outputs < B, layer outputs < 35 B < 2^35.2.
//  partial rounds: lane 0 is an S-box output (< B) before the layer and <= 2 B + |sum| after; lane 1 grows by |sum| <= p/2 + 2^11 per
//  round; the integer-diagonal lanes 2, 4, 5, 7, 8 grow by at most x4 + |sum| per round and are reduced after rounds 4 and 9, so they
//  stay below 2^35.2 * 4^5 + ... < 2^46, and they leave the last round below 2^30 * 4^3 + 2^33 < 2^37; lanes 11 and 15 (diagonal -+127)
//  grow by x127 + |sum| per round and are reduced after every odd round: 2^35.2 -> 2^42.2 -> 2^49.2 once at the start, afterwards
//  2^30 -> 2^37 -> 2^44, and they leave the last (even) round below 2^37.1. The lane sum stays below 2 * 2^49.2 + 5 * 2^46 + 9 * 2^36
//  < 2^50.4 (reduce() takes |x| < 2^52).
//  The fractional lanes (diagonal +-2^-K) contract: |s| <- |s| / 2^K + |sum|, from < 2^35.2 to < 2^34.3 (K = 1) / 2^32.5 (K = 3) / 2^31.8
//  (K = 4) / 2^30.1 (K = 8) after the first round and towards 2 |sum| < 2^31.1 after that; an integerization adds at most 2^30 + 1/2.
//  They gain K bits below the point per round: lanes 9 and 12 (K = 8) are integerized after every odd round (16 bits at that moment, 8
//  when a sum reads them), lanes 10, 13 (K = 3) and 14 (K = 4) after rounds 4 and 9 (15 / 20 bits at that moment — a lane below 2^32 has
//  21 to spare — and 12 / 16 when a sum reads them), lanes 3 and 6 (K = 1) only at the end (13 bits). All seven are integerized after
//  the last partial round, before the S-boxes of the full rounds read them. Their sum is below 2^35.8 in round 1 (8 bits below the
//  point) and below 2^35 from round 2 on (at most 16 bits): exact in a double. The host build checks every one of these operations
//  against 64-bit significands (Audit::inexact) and records the largest sum of their magnitudes (Audit::frac_sum).
template <class RcExt, class RcInt>
KB_HD void permute_impl(double s[16], RcExt rc_ext, RcInt rc_int) {
#if !defined(__HIP_DEVICE_COMPILE__)
  for (int i = 0; i < 16; i++) P2F_PROBE(in, s[i]);
#endif
  external_layer(s);
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = sbox_wide(s[i] + rc_ext(0, i));   // two linear layers in a row behind these: |y| < 2^40.6
  external_layer(s);
#pragma unroll
  for (int r = 1; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = sbox(s[i] + rc_ext(r, i));
    external_layer(s);
  }
#pragma unroll 1
  for (int r = 0; r < 13; r++) {
    s[0] = sbox(s[0] + rc_int(r));
    internal_layer(s);
    if (r == 4 || r == 9) {
      s[2] = reduce(s[2]); s[4] = reduce(s[4]); s[5] = reduce(s[5]); s[7] = reduce(s[7]); s[8] = reduce(s[8]);
      s[10] = integerize(s[10]); s[13] = integerize(s[13]); s[14] = integerize(s[14]);
    }
    if (r & 1) {
      s[11] = reduce(s[11]); s[15] = reduce(s[15]);
      s[9] = integerize(s[9]); s[12] = integerize(s[12]);
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    for (int i = 0; i < 16; i++) P2F_PROBE(lane, s[i]);
#endif
  }
  s[3] = integerize(s[3]); s[6] = integerize(s[6]); s[9] = integerize(s[9]); s[10] = integerize(s[10]);
  s[12] = integerize(s[12]); s[13] = integerize(s[13]); s[14] = integerize(s[14]);
#pragma unroll
  for (int r = 4; r < 8; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = sbox(s[i] + rc_ext(r, i));
    external_layer(s);
  }
}

// u32 Montgomery word m < 2^32 -> canonical value in (-p, 0]: m / R = -(m * p^-1 mod R) * p / R (mod p), the integer
// Montgomery step with a zero high word
KB_HD double load_monty(uint32_t m) {
  const uint32_t t = m * kb::MU;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t u = __umulhi(t, kb::P);
#else
  const uint32_t u = (uint32_t)(((uint64_t)t * kb::P) >> 32);
#endif
  // m = t p (mod 2^32), so (m - t p) / 2^32 = -u exactly when m != 0 ... and 0 - u + (m != 0 ? 0 : 0): hi(m) = 0, borrow-free
  return -(double)u;
}
// integer |x| < 2^51 -> x * R mod p as a Montgomery word in [0, p)
KB_HD uint32_t store_monty(double x) {
  const double h = x * 4294967296.0;  // exact
  const double q = rne(h * PINV);
  const double r = fma_(-q, P, h);  // integer, |r| <= p/2 + p 2^-20
  const int32_t v = (int32_t)r;
  return (uint32_t)v + ((uint32_t)(v >> 31) & kb::P);
}

// The 128 + 13 round constants are 282 SGPRs' worth: inside a loop (sponge over a row's columns) the compiler would hoist
// their scalar loads out of the loop and park them in VGPR lanes (v_writelane / v_readlane: +6 % VALU instructions per
// permutation). Laundering the table address through an empty asm per call keeps the loads inside the permutation,
// where they are s_load_dwordx16 running under the VALU work.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) * const_table;  // constant address space: uniform loads stay scalar
__device__ __forceinline__ void permute(double s[16]) {
  uint64_t a_ext = (uint64_t)&d_rc_ext[0][0], a_in = (uint64_t)&d_rc_int[0];
  asm volatile("" : "+s"(a_ext), "+s"(a_in));
  const const_table ext = (const_table)a_ext, in = (const_table)a_in;
  permute_impl(
      s, [ext](int r, int i) { return ext[16 * r + i]; }, [in](int r) { return in[r]; });
}
#else
void permute(double s[16]);  // device only
#endif

// canonical balanced value of a Montgomery constant
inline double canonical_balanced(uint32_t monty) {
  const uint32_t c = kb::from_monty(monty);
  return c > kb::P / 2 ? (double)c - P : (double)c;
}
inline void permute_host(double s[16]) {
  permute_impl(
      s, [](int r, int i) { return canonical_balanced(p2::ZKM_RC_16_30_MONTY[r < 4 ? r : r + 13][i]); },
      [](int r) { return canonical_balanced(p2::ZKM_RC_16_30_MONTY[4 + r][0]); });
}
// the integer permutation's interface on top of the FP64 one (host): Montgomery words in, Montgomery words out
inline void permute_host_words(uint32_t w[16]) {
  double s[16];
  for (int i = 0; i < 16; i++) s[i] = load_monty(w[i]);
  permute_host(s);
  for (int i = 0; i < 16; i++) w[i] = store_monty(s[i]);
}

// Host mirrors of the hashing kernels' data flow (merkle.cuh), state kept in doubles exactly as there:
// absorb_row / hash_leaves: overwrite the rate with the next <= 8 words, permute, capacity carried on unreduced
inline void sponge_host(const uint32_t* words, size_t n, uint32_t digest[8]) {
  double s[16];
  for (int i = 0; i < 16; i++) s[i] = 0.0;
  for (size_t g0 = 0; g0 < n; g0 += 8) {
    for (size_t i = 0; i < 8 && g0 + i < n; i++) s[i] = load_monty(words[g0 + i]);
    permute_host(s);
  }
  for (int i = 0; i < 8; i++) digest[i] = store_monty(s[i]);
}
// compress_layer: node = compress(left, right); with an injected row: node = compress(node, hash(row)), both halves handed over
// as unreduced doubles
inline void compress_inject_host(const uint32_t left[8], const uint32_t right[8], const uint32_t* row, size_t n, uint32_t out[8]) {
  double s[16], node[8];
  for (int i = 0; i < 8; i++) { s[i] = load_monty(left[i]); s[8 + i] = load_monty(right[i]); }
  permute_host(s);
  if (n > 0) {
    for (int i = 0; i < 8; i++) node[i] = s[i];
    for (int i = 0; i < 16; i++) s[i] = 0.0;
    for (size_t g0 = 0; g0 < n; g0 += 8) {
      for (size_t i = 0; i < 8 && g0 + i < n; i++) s[i] = load_monty(row[g0 + i]);
      permute_host(s);
    }
    for (int i = 0; i < 8; i++) { s[8 + i] = s[i]; s[i] = node[i]; }
    permute_host(s);
  }
  for (int i = 0; i < 8; i++) out[i] = store_monty(s[i]);
}

inline hipError_t upload_tables() {
  double ext[8][16], in[13];
  for (int r = 0; r < 8; r++)
    for (int i = 0; i < 16; i++) ext[r][i] = canonical_balanced(p2::ZKM_RC_16_30_MONTY[r < 4 ? r : r + 13][i]);
  for (int r = 0; r < 13; r++) in[r] = canonical_balanced(p2::ZKM_RC_16_30_MONTY[4 + r][0]);
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(d_rc_ext), ext, sizeof ext);
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(d_rc_int), in, sizeof in);
}

}  // namespace p2f
