// TEST INFRASTRUCTURE — CPU oracle. Never linked, imported or executed by the product path.
//
// Poseidon2 width-16 permutation over KoalaBear, sponge hash, 2-to-1 compression and the
// duplex challenger, restated from the reference in canonical arithmetic.
//
// Follows:
//   schedule (initial ext layer; 4 full; 13 partial on lane 0; 4 full; S-box x^3):
//       crates/recursion/core/include/poseidon2_wide.hpp:11-146,
//       crates/primitives/src/lib.rs:1107-1122 (RC rows 0-3 | 4-16 elt 0 | 17-20)
//   external layer circ(2*M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]:
//       crates/recursion/core/include/poseidon2.hpp:21-52
//   internal layer s_i <- s_i * diag_i + sum(s):  poseidon2.hpp:54-71
//   PaddingFreeSponge (rate 8, overwrite mode, no padding), TruncatedPermutation compress:
//       crates/recursion/circuit/src/hash.rs:40-49,75-80
//   DuplexChallenger<KoalaBear, Perm, 16, 8>:     crates/recursion/circuit/src/challenger.rs:90-114,201-233
//
// Pinning: the constant tables are generated from the reference header (gen_fixtures.py);
// the layer/schedule logic cannot be compiled from the reference here (its prelude needs a
// cbindgen-generated header), so the permutation as a whole is checked against the three
// known-answer vectors recorded in SURVEY.md §8(c) (tests/golden/poseidon2_kat.json) and is
// otherwise "parity unpinned" at the bit level.
#pragma once
#include "field.hpp"
#include <cstring>

namespace orc {

#include "poseidon2_constants.inc"

static const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};

static inline void external_layer(F s[16]) {
  F t[16];
  for (int blk = 0; blk < 4; blk++)
    for (int r = 0; r < 4; r++) {
      F acc = 0;
      for (int c = 0; c < 4; c++) acc = fadd(acc, fmul((F)M4[r][c], s[4 * blk + c]));
      t[4 * blk + r] = acc;
    }
  // circ(2*M4, M4, M4, M4): every block additionally receives the column sums.
  F sums[4] = {0, 0, 0, 0};
  for (int k = 0; k < 4; k++)
    for (int blk = 0; blk < 4; blk++) sums[k] = fadd(sums[k], t[4 * blk + k]);
  for (int i = 0; i < 16; i++) s[i] = fadd(t[i], sums[i % 4]);
}

static inline void internal_layer(F s[16]) {
  F sum = 0;
  for (int i = 0; i < 16; i++) sum = fadd(sum, s[i]);
  for (int i = 0; i < 16; i++) s[i] = fadd(fmul(s[i], ORC_INTERNAL_DIAG_16[i]), sum);
}

static inline F sbox(F x) { return fmul(fmul(x, x), x); }

static inline void poseidon2_permute(F s[16]) {
  external_layer(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 16; i++) s[i] = sbox(fadd(s[i], ORC_RC_16_30[r][i]));
    external_layer(s);
  }
  for (int r = 0; r < 13; r++) {
    s[0] = sbox(fadd(s[0], ORC_RC_16_30[4 + r][0]));
    internal_layer(s);
  }
  for (int r = 4; r < 8; r++) {
    for (int i = 0; i < 16; i++) s[i] = sbox(fadd(s[i], ORC_RC_16_30[13 + r][i]));
    external_layer(s);
  }
}

struct Digest {
  F d[8];
  bool operator==(const Digest& o) const { return !memcmp(d, o.d, sizeof d); }
  bool operator!=(const Digest& o) const { return !(*this == o); }
};

// hash.rs:40-49
static inline Digest hash_slice(const F* in, size_t len) {
  F st[16] = {0};
  for (size_t off = 0; off < len; off += 8) {
    size_t m = len - off < 8 ? len - off : 8;
    for (size_t i = 0; i < m; i++) st[i] = in[off + i];
    poseidon2_permute(st);
  }
  Digest d; memcpy(d.d, st, sizeof d.d);
  return d;
}
// hash.rs:75-80
static inline Digest compress(const Digest& l, const Digest& r) {
  F st[16];
  memcpy(st, l.d, 32); memcpy(st + 8, r.d, 32);
  poseidon2_permute(st);
  Digest d; memcpy(d.d, st, sizeof d.d);
  return d;
}

// challenger.rs:62-233
struct Challenger {
  F state[16];
  std::vector<F> in, out;
  Challenger() { memset(state, 0, sizeof state); }
  void duplexing() {
    assert(in.size() <= 8);
    for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
    in.clear();
    poseidon2_permute(state);
    out.assign(state, state + 8);
  }
  void observe(F v) {
    out.clear();
    in.push_back(v);
    if (in.size() == 8) duplexing();
  }
  void observe_slice(const F* v, size_t n) { for (size_t i = 0; i < n; i++) observe(v[i]); }
  void observe_digest(const Digest& d) { observe_slice(d.d, 8); }
  void observe_ext(const E& e) { observe_slice(e.c, 4); }
  F sample() {
    if (!in.empty() || out.empty()) duplexing();
    F v = out.back(); out.pop_back();
    return v;
  }
  E sample_ext() { E e; for (int i = 0; i < 4; i++) e.c[i] = sample(); return e; }
  uint32_t sample_bits(int bits) { F v = sample(); return v & ((1u << bits) - 1); }
  bool check_witness(int bits, F w) { observe(w); return sample_bits(bits) == 0; }
  // grind: smallest witness w.r.t. canonical value (any valid witness verifies).
  F grind(int bits) {
    for (F w = 0; w < P; w++) {
      Challenger c = *this;
      if (c.check_witness(bits, w)) { check_witness(bits, w); return w; }
    }
    assert(false); return 0;
  }
};

}  // namespace orc
