/* A plain C consumer of include/zkm_hip.h — what a cgo / JNI / Rust-FFI binding sees: no Python, no torch. Compiled by
 * tests/test_host_abi.py with gcc against the header and linked to libzkm_hip.so.
 *
 *   consumer            no GPU expected: zkm_ctx_create must fail loudly (no CPU fallback) -> exit 0 and print the message
 *   consumer gpu        on a GPU: an AddSub event (5 + 7 = 12) -> device trace -> one-matrix commitment; prints "ok <root word>", then two
 *                       GlobalLookupEvents -> the Global chip's trace; prints "global <the fourteen words of the shard's digest>"
 *   consumer prove      on a GPU, the whole hot path as the Rust shim would drive it: zkm_chip_desc of the AddSub chip (constraint bytecode and
 *                       lookups from a header the test generates with the recorder: tests/c_abi/addsub_desc.h), forty AluEvents -> device trace,
 *                       zkm_pk_setup (no preprocessed trace) -> zkm_challenger_init + zkm_pk_observe_into -> zkm_commit -> zkm_open with the
 *                       too-small-buffer retry -> prints "proof <words> <fnv1a of the stream> <main commitment word 0> <transcript word>"
 *   consumer fail       on a GPU, three failures driven through the ABI — a freed trace handle handed to zkm_commit, zkm_open with a chip list
 *                       that disagrees with the commit, an allocation refused by a capped pool (zkm_ctx_set_memory_limit) — each must return
 *                       non-zero with its message in zkm_last_error() and leave the transcript alone; the SAME context then proves the shard:
 *                       the line printed at the end equals `consumer prove`'s (crates/stark/src/prover.rs:206-208: the reference's callers
 *                       unwrap(); across a C ABI a failure has to be a status, never an unwind, and must not poison the context) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkm_hip.h"
#ifdef ZKM_HAVE_ADDSUB_DESC
#include "addsub_desc.h"   /* ADDSUB_PROGRAM[], ADDSUB_LOOKUPS[], ADDSUB_NUM_CONSTRAINTS, ADDSUB_LQD, N_PUBLIC_VALUES, NUM_PV_ELTS */

static int prove(zkm_ctx* ctx, int with_failures) {
  enum { N = 40 };
  zkm_alu_event ev[N];
  memset(ev, 0, sizeof ev);
  for (int i = 0; i < N; i++) {
    ev[i].pc = 0x1000 + 4 * i; ev[i].next_pc = ev[i].pc + 4;
    ev[i].opcode = i & 1;                                   /* ADD, SUB */
    ev[i].b = 0x01020304u * (uint32_t)(i + 1); ev[i].c = 0xfffefdfcu - 77u * (uint32_t)i;
    ev[i].a = ev[i].opcode ? ev[i].b - ev[i].c : ev[i].b + ev[i].c;
  }
  zkm_matrix* trace = NULL;
  if (zkm_tracegen_alu(ctx, ZKM_CHIP_ADD_SUB, ev, N, -1, NULL, &trace) != 0) { printf("tracegen: %s\n", zkm_last_error()); return 20; }
  if (with_failures) {
    /* (1) a handle that was freed: refused by name, nothing dereferenced */
    zkm_matrix* gone = NULL;
    if (zkm_tracegen_alu(ctx, ZKM_CHIP_ADD_SUB, ev, N, -1, NULL, &gone) != 0) return 40;
    zkm_matrix_free(ctx, gone);
    zkm_matrix_free(ctx, gone);                               /* a double free is ignored */
    const char* nm[1] = {"AddSub"};
    const zkm_matrix* dead[1] = {gone};
    uint32_t c8[ZKM_DIGEST_ELEMS], ord[1];
    static uint32_t pv0[N_PUBLIC_VALUES];
    zkm_main_data* none = NULL;
    if (zkm_commit(ctx, 1, nm, dead, pv0, N_PUBLIC_VALUES, 1, c8, ord, &none) == 0 || none != NULL) return 41;
    if (!strstr(zkm_last_error(), "not a live handle")) { printf("freed handle: %s\n", zkm_last_error()); return 42; }
    printf("refused: %s\n", zkm_last_error());
    /* (3) a capped pool: the traces of 2^20 events do not fit 8 MiB; the failing call gives back what it took */
    const size_t held = zkm_ctx_memory_held(ctx);
    if (zkm_ctx_set_memory_limit(ctx, held + ((size_t)8 << 20)) != 0) return 43;
    enum { BIG = 1 << 20 };
    zkm_alu_event* many = (zkm_alu_event*)calloc(BIG, sizeof *many);
    for (int i = 0; i < BIG; i++) { many[i].opcode = 0; many[i].b = (uint32_t)i; many[i].c = 1; many[i].a = (uint32_t)i + 1; many[i].pc = 4 * (uint32_t)i; many[i].next_pc = many[i].pc + 4; }
    zkm_matrix* big = NULL;
    if (zkm_tracegen_alu(ctx, ZKM_CHIP_ADD_SUB, many, BIG, -1, NULL, &big) == 0 || big != NULL) return 44;
    if (!strstr(zkm_last_error(), "out of device memory")) { printf("capped pool: %s\n", zkm_last_error()); return 45;}
    printf("refused: %s\n", zkm_last_error());
    if (zkm_ctx_memory_held(ctx) > held + ((size_t)8 << 20)) return 46;
    if (zkm_ctx_set_memory_limit(ctx, 0) != 0) return 47;
    if (zkm_tracegen_alu(ctx, ZKM_CHIP_ADD_SUB, many, BIG, -1, NULL, &big) != 0) { printf("after the cap: %s\n", zkm_last_error()); return 48; }
    zkm_matrix_free(ctx, big);
    free(many);
  }
  zkm_chip_desc chip;
  memset(&chip, 0, sizeof chip);
  chip.name = "AddSub";
  chip.main_width = (uint32_t)zkm_tracegen_alu_width(ZKM_CHIP_ADD_SUB);
  chip.prep_index = -1;
  chip.log_quotient_degree = ADDSUB_LQD;
  chip.local_only = 1;
  chip.num_constraints = ADDSUB_NUM_CONSTRAINTS;
  chip.lookups = ADDSUB_LOOKUPS; chip.lookups_len = sizeof ADDSUB_LOOKUPS / 4;
  chip.program = ADDSUB_PROGRAM; chip.program_len = sizeof ADDSUB_PROGRAM / 4;
  const zkm_fri_config fri = {1, 84, 16};                   /* core configuration, kb31_poseidon2.rs:203-213 */
  uint32_t igcs[14] = {0};
  zkm_pk* pk = NULL;
  if (zkm_pk_setup(ctx, 0, NULL, NULL, 0, igcs, fri.log_blowup, &pk) != 0) { printf("pk: %s\n", zkm_last_error()); return 21; }
  zkm_challenger ch;
  zkm_challenger_init(&ch);
  if (zkm_pk_observe_into(pk, &ch) != 0) return 22;
  static uint32_t pv[N_PUBLIC_VALUES];
  const char* names[1] = {"AddSub"};
  const zkm_matrix* traces[1] = {trace};
  uint32_t main_commit[ZKM_DIGEST_ELEMS], order[1];
  zkm_main_data* data = NULL;
  if (zkm_commit(ctx, 1, names, traces, pv, N_PUBLIC_VALUES, fri.log_blowup, main_commit, order, &data) != 0) { printf("commit: %s\n", zkm_last_error()); return 23; }
  /* first with a buffer that is too small: the call fails, says how many words it takes, and leaves the transcript untouched */
  uint32_t tiny[8];
  size_t need = 0;
  const zkm_challenger before = ch;
  if (zkm_prove_shard(ctx, pk, 1, &chip, traces, pv, N_PUBLIC_VALUES, &fri, NUM_PV_ELTS, &ch, tiny, 8, &need) == 0) return 24;
  if (need <= 8 || memcmp(&before, &ch, sizeof ch) != 0 || !strstr(zkm_last_error(), "too small")) return 25;
  uint32_t* proof = (uint32_t*)malloc(need * 4);
  size_t len = 0;
  if (with_failures) {
    /* (2) a chip list that disagrees with what was committed: refused, the commitment and the transcript untouched */
    zkm_chip_desc wrong = chip;
    wrong.main_width = chip.main_width + 1;
    if (zkm_open(ctx, pk, data, &wrong, &fri, NUM_PV_ELTS, &ch, proof, need, &len) == 0) return 50;
    if (!strstr(zkm_last_error(), "does not match") || memcmp(&before, &ch, sizeof ch) != 0) { printf("chip list: %s\n", zkm_last_error()); return 51; }
    printf("refused: %s\n", zkm_last_error());
    /* and a proving key that was freed */
    zkm_pk* pk2 = NULL;
    if (zkm_pk_setup(ctx, 0, NULL, NULL, 0, igcs, fri.log_blowup, &pk2) != 0) return 52;
    zkm_pk_free(ctx, pk2);
    if (zkm_open(ctx, pk2, data, &chip, &fri, NUM_PV_ELTS, &ch, proof, need, &len) == 0 || !strstr(zkm_last_error(), "not a live handle")) return 53;
  }
  if (zkm_open(ctx, pk, data, &chip, &fri, NUM_PV_ELTS, &ch, proof, need, &len) != 0) { printf("open: %s\n", zkm_last_error()); return 26; }
  if (len != need || memcmp(proof, main_commit, 32) != 0) return 27;
  uint32_t h = 2166136261u;
  for (size_t i = 0; i < len; i++) { h ^= proof[i]; h *= 16777619u; }
  printf("proof %zu %u %u %u\n", len, h, main_commit[0], zkm_challenger_sample(&ch));
  free(proof);
  zkm_pk_free(ctx, pk);
  zkm_matrix_free(ctx, trace);
  return 0;
}
#endif

int main(int argc, char** argv) {
  zkm_ctx* ctx = NULL;
  int rc = zkm_ctx_create(0, &ctx);
  if (argc < 2) {
    if (rc == 0) { printf("unexpected: a context without a GPU\n"); return 1; }
    printf("refused: %s\n", zkm_last_error());
    return strstr(zkm_last_error(), "no CPU fallback") ? 0 : 2;
  }
  if (rc != 0) { printf("ctx: %s\n", zkm_last_error()); return 3; }
#ifdef ZKM_HAVE_ADDSUB_DESC
  if (!strcmp(argv[1], "prove")) { rc = prove(ctx, 0); zkm_ctx_destroy(ctx); return rc; }
  if (!strcmp(argv[1], "fail")) { rc = prove(ctx, 1); zkm_ctx_destroy(ctx); return rc; }
#endif
  zkm_alu_event ev;
  memset(&ev, 0, sizeof ev);
  ev.pc = 0x1000; ev.next_pc = 0x1004; ev.opcode = 0; ev.a = 12; ev.b = 5; ev.c = 7;   /* ADD */
  zkm_matrix* trace = NULL;
  if (zkm_tracegen_alu(ctx, ZKM_CHIP_ADD_SUB, &ev, 1, -1, NULL, &trace) != 0) { printf("tracegen: %s\n", zkm_last_error()); return 4; }
  if (zkm_matrix_height(trace) != 16 || zkm_matrix_width(trace) != zkm_tracegen_alu_width(ZKM_CHIP_ADD_SUB)) return 5;
  uint32_t rows[16 * 19];
  if (zkm_matrix_download(ctx, trace, rows) != 0) return 6;
  /* column 2 is the low byte of the sum, stored in Montgomery form: 12 * 2^32 mod p */
  const uint32_t p = 0x7f000001u;
  const uint32_t want = (uint32_t)((((unsigned long long)12) << 32) % p);
  if (rows[2] != want || rows[19 + 2] != 0) { printf("row mismatch: %u vs %u\n", rows[2], want); return 7; }
  const zkm_matrix* mats[1] = {trace};
  uint32_t root[ZKM_DIGEST_ELEMS];
  zkm_pcs_data* data = NULL;
  if (zkm_pcs_commit(ctx, 1, mats, NULL, 1, root, &data) != 0) { printf("commit: %s\n", zkm_last_error()); return 8; }
  printf("ok %u\n", root[0]);
  zkm_pcs_data_free(ctx, data);
  zkm_matrix_free(ctx, trace);
  /* a register's access chain through one shard: received as the previous shard left it, sent as this one leaves it */
  zkm_global_lookup_event ge[2];
  memset(ge, 0, sizeof ge);
  ge[0].message[0] = 0; ge[0].message[1] = 0; ge[0].message[2] = 8; ge[0].is_receive = 1; ge[0].kind = 1;
  ge[1].message[0] = 1; ge[1].message[1] = 4003; ge[1].message[2] = 8; ge[1].message[3] = 12; ge[1].is_receive = 0; ge[1].kind = 1;
  zkm_byte_lookups* blu = NULL;
  if (zkm_byte_lookups_create(ctx, &blu) != 0) { printf("blu: %s\n", zkm_last_error()); return 9; }
  zkm_matrix* global = NULL;
  if (zkm_tracegen_global(ctx, ge, 2, -1, blu, &global) != 0) { printf("global: %s\n", zkm_last_error()); return 10; }
  if (zkm_matrix_height(global) != 16 || zkm_matrix_width(global) != ZKM_GLOBAL_WIDTH) return 11;
  static uint32_t grows[16 * ZKM_GLOBAL_WIDTH];
  if (zkm_matrix_download(ctx, global, grows) != 0) return 12;
  printf("global");
  for (int k = 0; k < 14; k++) printf(" %u", grows[15 * ZKM_GLOBAL_WIDTH + 85 + k]);
  printf("\n");
  ge[1].message[0] = 1u << 16;   /* not a u16: refused, with a message */
  zkm_matrix* none = NULL;
  if (zkm_tracegen_global(ctx, ge, 2, -1, blu, &none) == 0 || !strstr(zkm_last_error(), "not a u16")) return 13;
  zkm_matrix_free(ctx, global);
  zkm_byte_lookups_free(ctx, blu);
  zkm_ctx_destroy(ctx);
  return 0;
}
