/* A plain C consumer of include/zkm_hip.h — what a cgo / JNI / Rust-FFI binding sees: no Python, no torch. Compiled by
 * tests/test_host_abi.py with gcc against the header and linked to libzkm_hip.so.
 *
 *   consumer            no GPU expected: zkm_ctx_create must fail loudly (no CPU fallback) -> exit 0 and print the message
 *   consumer gpu        on a GPU: an AddSub event (5 + 7 = 12) -> device trace -> one-matrix commitment; prints "ok <root word>", then two
 *                       GlobalLookupEvents -> the Global chip's trace; prints "global <the fourteen words of the shard's digest>" */
#include <stdio.h>
#include <string.h>
#include "zkm_hip.h"

int main(int argc, char** argv) {
  zkm_ctx* ctx = NULL;
  int rc = zkm_ctx_create(0, &ctx);
  if (argc < 2) {
    if (rc == 0) { printf("unexpected: a context without a GPU\n"); return 1; }
    printf("refused: %s\n", zkm_last_error());
    return strstr(zkm_last_error(), "no CPU fallback") ? 0 : 2;
  }
  if (rc != 0) { printf("ctx: %s\n", zkm_last_error()); return 3; }
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
