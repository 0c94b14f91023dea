// TEST INFRASTRUCTURE — builds into oracle/_ref/kb31_ref (git-ignored).
//
// Driver (our own code) around two reference headers compiled *where they lie*
// under /root/reference, with no stand-ins:
//   crates/core/machine/include/kb31_t.hpp          (KoalaBear Montgomery class, host branch :458-621)
//   crates/recursion/core/include/poseidon2_constants.hpp (RC_16_30_U32 :534, internal diag :1083)
// The reference's poseidon2.hpp / poseidon2_wide.hpp layer functions are NOT buildable
// here: they include prelude.hpp -> a cbindgen-generated header that only exists after
// the Rust build. We therefore pin only (a) field arithmetic and (b) the constant tables
// against the real reference; the permutation schedule is restated (oracle/poseidon2.hpp).
//
// Usage:
//   kb31_ref constants            -> JSON: rc (30x16 raw u32), diag_monty (16 u32), diag_canonical
//   kb31_ref fieldkat SEED N      -> JSON: N seeded (a,b) pairs with a+b, a-b, a*b, 1/a, a<<3, a>>5, a^5 (canonical)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kb31_t.hpp"
#include "poseidon2_constants.hpp"

using namespace zkm_recursion_core_sys::constants;

static uint64_t sm_state;
static uint64_t splitmix64() {
  uint64_t z = (sm_state += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "constants")) {
    printf("{\"rc\": [");
    for (int r = 0; r < 30; r++) {
      printf("%s[", r ? ", " : "");
      for (int i = 0; i < 16; i++) printf("%s%u", i ? ", " : "", RC_16_30_U32[r][i]);
      printf("]");
    }
    printf("],\n \"diag_monty\": [");
    for (int i = 0; i < 16; i++)
      printf("%s%u", i ? ", " : "", POSEIDON2_INTERNAL_MATRIX_DIAG_16_KOALABEAR_MONTY[i].val);
    printf("],\n \"diag_canonical\": [");
    for (int i = 0; i < 16; i++)
      printf("%s%u", i ? ", " : "",
             kb31_t::from_monty(POSEIDON2_INTERNAL_MATRIX_DIAG_16_KOALABEAR_MONTY[i].val));
    printf("],\n \"mod\": %u, \"monty_one\": %u}\n", kb31_t::MOD, kb31_t::one().val);
    return 0;
  }
  if (argc >= 4 && !strcmp(argv[1], "fieldkat")) {
    sm_state = strtoull(argv[2], nullptr, 0);
    int n = atoi(argv[3]);
    printf("{\"seed\": %llu, \"cases\": [", (unsigned long long)sm_state);
    for (int k = 0; k < n; k++) {
      uint32_t a = (uint32_t)(splitmix64() % kb31_t::MOD);
      uint32_t b = (uint32_t)(splitmix64() % kb31_t::MOD);
      if (k == 0) { a = 1; b = kb31_t::MOD - 1; }
      if (k == 1) { a = kb31_t::MOD - 1; b = kb31_t::MOD - 1; }
      if (k == 2) { a = 2; b = 0; }
      kb31_t A = kb31_t::from_canonical_u32(a), B = kb31_t::from_canonical_u32(b);
      kb31_t shl = A << 3;
      kb31_t shr = A; shr >>= 5;
      kb31_t p5 = A ^ 5u;
      printf("%s{\"a\": %u, \"b\": %u, \"a_monty\": %u, \"add\": %u, \"sub\": %u, \"mul\": %u, \"inv\": %u, \"shl3\": %u, \"shr5\": %u, \"pow5\": %u}",
             k ? ",\n " : "", a, b, A.val, (A + B).as_canonical_u32(), (A - B).as_canonical_u32(),
             (A * B).as_canonical_u32(), A.reciprocal().as_canonical_u32(), shl.as_canonical_u32(),
             shr.as_canonical_u32(), p5.as_canonical_u32());
    }
    printf("]}\n");
    return 0;
  }
  fprintf(stderr, "usage: kb31_ref constants | fieldkat SEED N\n");
  return 2;
}
