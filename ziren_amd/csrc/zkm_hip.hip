// libzkm_hip.so — host side of the MI355X shard prover: context, device memory, the
// Fiat-Shamir transcript, and the commit/open orchestration behind the C ABI of
// include/zkm_hip.h. The reference's host side is Rust (CpuProver, crates/stark/src/prover.rs);
// no Rust toolchain exists in this environment, so this layer is C++ and mirrors
// CpuProver::commit (:258-292) and CpuProver::open (:298-653) step for step.
//
// There is no CPU fallback anywhere in this file: every entry point that computes needs the
// GPU and fails loudly (non-zero status + zkm_last_error) if HIP is unavailable.
#include "../../include/zkm_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <exception>
#include <vector>

#include <sched.h>
#include <time.h>
#include <sys/prctl.h>

#include "kb31.cuh"
#include "poseidon2.cuh"
#include "lde.cuh"
#include "merkle.cuh"
#include "stark.cuh"
#include "open.cuh"
#include "tracegen.cuh"

using kb::E4;

#include "host_ctx.hpp"
#include "host_pcs.hpp"
#include "host_open.hpp"

// ---- C ABI ---------------------------------------------------------------------------------------
template <int CHIP>
static void launch_alu_rows(zkm_ctx* ctx, const uint32_t* d_events, size_t n_events, size_t height, uint32_t* out, uint32_t* counts) {
  KLAUNCH(ctx, "tracegen_alu", 4.0 * tracegen::event_words(CHIP) * n_events + 4.0 * height * tracegen::chip_width(CHIP), tracegen::alu_rows<CHIP>,
          dim3(div_up(height, (counts ? tracegen::TILES_PER_BLOCK : 1) * tracegen::THREADS)), dim3(tracegen::THREADS),
          counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0, d_events, n_events, height, out, counts,
          counts ? tracegen::TILES_PER_BLOCK : 1);
}

// Nothing unwinds across the C ABI (SURVEY 8b "Errors"): every entry point that can fail returns a status and leaves the message in
// zkm_last_error(); whatever is thrown — a std::exception or anything else — ends here.
#define API_BEGIN try {
#define API_END                              \
  }                                          \
  catch (const std::exception& e) {          \
    g_err = e.what();                        \
    return -1;                               \
  }                                          \
  catch (...) {                              \
    g_err = "unknown exception (not derived from std::exception)"; \
    return -1;                               \
  }                                          \
  return 0;

// The outermost API call on a context that ends in an exception gives back every pool buffer it took and still holds: a call that fails
// returns no handle, so whatever it allocated is garbage (objects it already destroyed while unwinding released theirs: a second release
// is ignored). The streams are drained first — the pool reuses buffers in stream order, and the failing call's kernels may still be
// running. After that the context proves again as if the call had not been made (tests/c_abi/consumer.c `fail`).
struct CallScope {
  zkm_ctx* c;
  int exceptions;
  explicit CallScope(zkm_ctx* ctx) : c(ctx), exceptions(std::uncaught_exceptions()) {
    if (c->call_depth++ == 0) c->call_allocs.clear();
  }
  ~CallScope() {
    if (--c->call_depth > 0) return;
    if (std::uncaught_exceptions() > exceptions) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipStreamSynchronize(c->stream2);
      if (c->ev_dma) (void)hipStreamSynchronize(c->ev_dma);
      (void)hipGetLastError();
      c->cur = c->stream;
      c->side_pending = false;
      for (void* p : c->side_deferred) c->release(p);
      c->side_deferred.clear();
      for (void* p : c->call_allocs) c->release(p);
    }
    c->call_allocs.clear();
  }
};

extern "C" {

const char* zkm_last_error(void) { return g_err.c_str(); }

#ifndef ZKM_SOURCES_DIGEST
#define ZKM_SOURCES_DIGEST "unrecorded"
#endif
#ifndef ZKM_HIPCC_VERSION
#define ZKM_HIPCC_VERSION "unknown"
#endif
// also found by reading the file's bytes (ziren_amd/build.py recorded_digest): the build script checks it without loading the library
static const char build_info_string[] = "ZKM_SOURCES_DIGEST=" ZKM_SOURCES_DIGEST ";hipcc=" ZKM_HIPCC_VERSION ";arch=gfx950";
const char* zkm_build_info(void) { return build_info_string; }

int zkm_ctx_create(int device, zkm_ctx** out) {
  API_BEGIN
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    throw std::runtime_error("no HIP device available: libzkm_hip has no CPU fallback (hipGetDeviceCount: " +
                             std::string(hipGetErrorString(e)) + ")");
  if (device < 0 || device >= count) throw std::runtime_error("device index out of range");
  HIP_CHECK(hipSetDevice(device));
  zkm_ctx* c = new zkm_ctx();
  c->device = device;
  // non-blocking streams: nothing in this library runs on the null stream, and a stream that synchronises with it pays for every launch
  // as soon as the process has another null-stream user (torch.cuda initialised beside us — every rank of an N > 1 run: +0.5-0.8 ms a shard)
  HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  c->cur = c->stream;
  HIP_CHECK(p2::upload_tables());
  HIP_CHECK(p2f::upload_tables());
  HIP_CHECK(tracegen::upload_tables());
  HIP_CHECK(septic::upload_tables());
  {
    // the Global chip's dummy point (crates/stark/src/septic_curve.rs:18-38) next to the start digest
    static const uint32_t DUMMY_X[7] = {1706420302, 1319108093, 148224806, 26874985, 1766171812, 1645633948, 2028659224};
    static const uint32_t DUMMY_Y[7] = {942390502, 1239997438, 458866455, 1843332012, 1309764648, 572807436, 74267719};
    uint32_t consts[28];
    for (int k = 0; k < 7; k++) {
      consts[k] = kb::to_monty(SEPTIC_X[k]); consts[7 + k] = kb::to_monty(SEPTIC_Y[k]);
      consts[14 + k] = kb::to_monty(DUMMY_X[k]); consts[21 + k] = kb::to_monty(DUMMY_Y[k]);
    }
    HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(tracegen::d_global_consts), consts, sizeof consts));
  }
  HIP_CHECK(hipFuncSetAttribute((const void*)lde::lde_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)lde::lde_rows_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::ADD_SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::BITWISE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::LT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::SHIFT_LEFT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::SHIFT_RIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::CLO_CLZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::JUMP>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::MOV_COND>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::MUL>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::DIVREM>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::cpu_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::MEMORY_INSTRS>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::MISC_INSTRS>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)tracegen::alu_rows<tracegen::BRANCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  {      // the big-field precompile kernels keep their byte-limb polynomials next to the lookup table
    const void* big_field_kernels[] = {
        (const void*)tracegen::u8_pair_histogram, (const void*)tracegen::ed_add_rows, (const void*)tracegen::ed_decompress_rows,
        (const void*)tracegen::weierstrass_rows<8, false>, (const void*)tracegen::weierstrass_rows<8, true>,
        (const void*)tracegen::weierstrass_rows<12, false>, (const void*)tracegen::weierstrass_rows<12, true>,
        (const void*)tracegen::uint256_mul_rows, (const void*)tracegen::u256x2048_mul_rows, (const void*)tracegen::weierstrass_decompress_rows<8, false>, (const void*)tracegen::weierstrass_decompress_rows<12, true>,
        (const void*)tracegen::fp_tower_rows<8, 0>, (const void*)tracegen::fp_tower_rows<8, 1>, (const void*)tracegen::fp_tower_rows<8, 2>,
        (const void*)tracegen::fp_tower_rows<12, 0>, (const void*)tracegen::fp_tower_rows<12, 1>, (const void*)tracegen::fp_tower_rows<12, 2>};
    for (const void* k : big_field_kernels) HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  HIP_CHECK(hipFuncSetAttribute((const void*)lde::lde_cols<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)lde::lde_cols<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_CHECK(hipFuncSetAttribute((const void*)stark::quotient_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + quotient_emit's 4.6 KiB static stage
  *out = c;
  API_END
}

void zkm_ctx_destroy(zkm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->prefetched) { (void)hipEventSynchronize(kv.second.landed); (void)hipEventDestroy(kv.second.landed); }   // their buffers are in `live`
  for (auto& kv : ctx->free_list) (void)hipFree(kv.second);
  for (auto& kv : ctx->live) (void)hipFree(kv.first);
  for (auto& kv : ctx->tw_fwd) (void)hipFree(kv.second);
  for (auto& kv : ctx->tw_inv) (void)hipFree(kv.second);
  for (auto& kv : ctx->pow_tabs) { (void)hipFree(kv.second.first); (void)hipFree(kv.second.second); }
  ctx->drop_coset_tables();
  for (auto& kv : ctx->row_tabs) (void)hipFree(kv.second);
  for (auto& kv : ctx->ct_fwd) (void)hipFree(kv.second);
  for (auto& kv : ctx->ct_inv) (void)hipFree(kv.second);
  for (auto& kv : ctx->selector_tabs) (void)hipFree(kv.second);
  for (auto& m : ctx->marks) (void)hipEventDestroy(m.second);
  for (auto& e : ctx->event_pool) (void)hipEventDestroy(e);
  for (auto& m : ctx->modules) (void)hipModuleUnload(m);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->ev_dma) (void)hipStreamDestroy(ctx->ev_dma);
  (void)hipStreamDestroy(ctx->stream);
  (void)hipStreamDestroy(ctx->stream2);
  if (ctx->up_dma) {
    (void)hipStreamDestroy(ctx->up_dma);
    (void)hipStreamDestroy(ctx->up_tr);
    for (int k = 0; k < 2; k++) { (void)hipFree(ctx->up_stage[k]); (void)hipEventDestroy(ctx->up_freed[k]); (void)hipEventDestroy(ctx->up_landed[k]); }
  }
  delete ctx;
}

// Return the cached (idle) device buffers of the context's pool to the driver.
int zkm_ctx_trim(zkm_ctx* ctx) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  ctx->sync(ctx->stream);
  for (auto& kv : ctx->free_list) { HIP_CHECK(hipFree(kv.second)); ctx->pool_bytes -= kv.first; }
  ctx->free_list.clear();
  ctx->drop_coset_tables();
  // per-height tables a long-lived prover accumulates (12 B per quotient-domain row per (height, degree); n words per height): rebuilt on
  // the next use of that height
  for (auto& kv : ctx->selector_tabs) HIP_CHECK(hipFree(kv.second));
  ctx->selector_tabs.clear();
  for (auto& kv : ctx->row_tabs) HIP_CHECK(hipFree(kv.second));
  ctx->row_tabs.clear();
  API_END
}

int zkm_ctx_set_memory_limit(zkm_ctx* ctx, size_t bytes) {
  API_BEGIN
  if (!ctx) throw std::runtime_error("zkm_ctx_set_memory_limit: null context");
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->pool_limit = bytes;
  if (bytes && ctx->pool_bytes > bytes) {      // a cap below what the pool holds: the cached (idle) buffers go back to the driver now
    HIP_CHECK(hipSetDevice(ctx->device));
    ctx->drop_cached();
  }
  API_END
}
size_t zkm_ctx_memory_held(zkm_ctx* ctx) {
  if (!ctx) return 0;
  std::lock_guard<std::mutex> lk(ctx->mu);
  return ctx->pool_bytes;
}

int zkm_ctx_synchronize(zkm_ctx* ctx) {
  API_BEGIN
  ctx->sync(ctx->stream);
  if (ctx->ev_dma) ctx->sync(ctx->ev_dma);   // and every event prefetch queued so far has landed
  API_END
}

int zkm_ctx_last_timings(zkm_ctx* ctx, const char** names, float* ms, int cap) {
  int n = (int)ctx->timing_names.size();
  for (int i = 0; i < n && i < cap; i++) { names[i] = ctx->timing_names[i].c_str(); ms[i] = ctx->timing_ms[i]; }
  return n;
}

int zkm_ctx_kernel_timings(zkm_ctx* ctx, const char** names, float* ms, uint32_t* calls, double* bytes, int cap) {
  int i = 0;
  for (auto& kv : ctx->kstats) {
    if (i < cap) { names[i] = kv.first.c_str(); ms[i] = (float)kv.second.ms; calls[i] = kv.second.calls; bytes[i] = kv.second.bytes; }
    i++;
  }
  return i;
}
void zkm_ctx_set_kernel_timing(zkm_ctx* ctx, int mode) { ctx->kernel_timing = mode; }
void zkm_ctx_set_kernel_timing_only(zkm_ctx* ctx, const char* name) { ctx->timing_only = name ? name : ""; ctx->kernel_timing = 3; }
void zkm_ctx_set_host_wait(zkm_ctx* ctx, int blocking) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->host_wait_blocking = blocking != 0; }
void zkm_ctx_set_lde_overlap(zkm_ctx* ctx, int on) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->lde_overlap = on != 0; }
void zkm_ctx_set_rows_up_front(zkm_ctx* ctx, int on) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->rows_up_front = on != 0; }

int zkm_ctx_register_quotient_kernel(zkm_ctx* ctx, const uint32_t* program, uint32_t program_len, const void* code_object,
                                     size_t code_object_len) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (!program_len || !code_object_len) throw std::runtime_error("empty program or code object");
  // one gfx950 code object, or — for a program long enough to be cut into several kernels (ziren_amd/codegen.py) — a container of them:
  // "ZKMQPART", u32 count, u32 zero, count x u64 lengths, then the code objects back to back. The first kernel stores its share of the
  // quotient, the others add theirs; they are launched in this order.
  std::vector<std::pair<const char*, size_t>> parts;
  std::vector<std::vector<char>> copies;
  const char* bytes = (const char*)code_object;
  if (code_object_len >= 16 && memcmp(bytes, "ZKMQPART", 8) == 0) {
    uint32_t count;
    memcpy(&count, bytes + 8, 4);
    size_t at = 16 + 8 * (size_t)count;
    if (count == 0 || count > 4096 || at > code_object_len) throw std::runtime_error("malformed quotient kernel container");
    for (uint32_t i = 0; i < count; i++) {
      uint64_t len;
      memcpy(&len, bytes + 16 + 8 * (size_t)i, 8);
      if (len == 0 || len > code_object_len - at) throw std::runtime_error("malformed quotient kernel container");
      copies.emplace_back(bytes + at, bytes + at + len);     // its own (aligned) allocation: hipModuleLoadData parses an ELF image
      at += len;
    }
    for (auto& c : copies) parts.push_back({c.data(), c.size()});
  } else {
    parts.push_back({bytes, code_object_len});
  }
  std::vector<hipFunction_t> fns;
  std::vector<hipModule_t> mods;
  try {
    for (auto& part : parts) {
      hipModule_t mod;
      HIP_CHECK(hipModuleLoadData(&mod, part.first));
      mods.push_back(mod);
      hipFunction_t fn;
      if (hipModuleGetFunction(&fn, mod, "zkm_quotient_specialized") != hipSuccess) throw std::runtime_error("code object lacks zkm_quotient_specialized");
      fns.push_back(fn);
    }
  } catch (...) {
    for (hipModule_t m : mods) (void)hipModuleUnload(m);
    throw;
  }
  for (hipModule_t m : mods) ctx->modules.push_back(m);
  const uint64_t key = fnv1a(program, program_len);
  ctx->quotient_fns[key] = fns;
  ctx->quotient_uniform_fns.erase(key);
  hipFunction_t uni;
  if (hipModuleGetFunction(&uni, mods[0], "zkm_quotient_uniforms") == hipSuccess) ctx->quotient_uniform_fns[key] = uni;   // of a program cut into parts: in the first
  else (void)hipGetLastError();      // a code object without the table kernel (no uniform arithmetic, an older generator): not an error
  API_END
}

int zkm_ctx_register_perm_kernel(zkm_ctx* ctx, const uint32_t* lookups, uint32_t lookups_len, uint32_t log_quotient_degree,
                                 const void* code_object, size_t code_object_len) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (lookups_len < 2 || !code_object_len) throw std::runtime_error("empty lookups blob or code object");
  std::vector<char> copy((const char*)code_object, (const char*)code_object + code_object_len);   // an aligned allocation of its own
  hipModule_t mod;
  HIP_CHECK(hipModuleLoadData(&mod, copy.data()));
  hipFunction_t fn;
  if (hipModuleGetFunction(&fn, mod, "zkm_perm_rows_specialized") != hipSuccess) {
    (void)hipModuleUnload(mod);
    throw std::runtime_error("code object lacks zkm_perm_rows_specialized");
  }
  ctx->modules.push_back(mod);
  ctx->perm_fns[perm_key(lookups, lookups_len, log_quotient_degree)] = fn;
  API_END
}

// Pinned host memory for trace buffers: uploads from it are plain DMA (no staging copy on the CPU).
void* zkm_host_alloc(zkm_ctx* ctx, size_t bytes) {
  std::lock_guard<std::mutex> lk(ctx->mu);
  void* p = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void zkm_host_free(zkm_ctx* ctx, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}

// Row-major host rows are copied in slabs (DMA stream) and transposed into the column-major matrix as they land
// (transpose stream); two staging slabs keep the DMA engine and the transpose kernel busy at the same time. Nothing
// here touches the compute stream: a consumer waits for m->ready in-stream right before its first use of the matrix, so
// the uploads of the later (shorter) traces of a shard overlap the LDE and hashing of the first ones.
static zkm_matrix* upload_async(zkm_ctx* ctx, const uint32_t* host, size_t height, size_t width) {
  log2_strict(height);
  if (!ctx->up_dma) {
    HIP_CHECK(hipStreamCreateWithFlags(&ctx->up_dma, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&ctx->up_tr, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      HIP_CHECK(hipMalloc((void**)&ctx->up_stage[k], zkm_ctx::UP_SLAB_BYTES));
      HIP_CHECK(hipEventCreateWithFlags(&ctx->up_freed[k], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&ctx->up_landed[k], hipEventDisableTiming));
    }
  }
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = width;
  try {
    m->d = ctx->alloc_n<uint32_t>(std::max<size_t>(height * width, 1));
    HIP_CHECK(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
    if (height * width) {
      // the pool hands buffers back in compute-stream order (release()); this one is written from the transpose stream, so that stream
      // first waits for whatever the compute stream still has queued — nothing, when every entry point has synchronised, but the
      // ordering no longer rests on that
      hipEvent_t reuse = ctx->get_event();
      HIP_CHECK(hipEventRecord(reuse, ctx->stream));
      HIP_CHECK(hipStreamWaitEvent(ctx->up_tr, reuse, 0));
      ctx->event_pool.push_back(reuse);
      if (width * 4 * 32 > zkm_ctx::UP_SLAB_BYTES) throw std::runtime_error("zkm_matrix_upload: matrix too wide for the staging slab");
      const size_t slab_rows = std::max<size_t>(32, std::min<size_t>(height, zkm_ctx::UP_SLAB_BYTES / (width * 4)) & ~(size_t)31);
      for (size_t r0 = 0; r0 < height; r0 += slab_rows) {
        const int k = ctx->up_next;
        ctx->up_next ^= 1;
        const size_t rows = std::min(slab_rows, height - r0);
        if (ctx->up_freed_set[k]) HIP_CHECK(hipStreamWaitEvent(ctx->up_dma, ctx->up_freed[k], 0));
        HIP_CHECK(hipMemcpyAsync(ctx->up_stage[k], host + r0 * width, rows * width * 4, hipMemcpyHostToDevice, ctx->up_dma));
        HIP_CHECK(hipEventRecord(ctx->up_landed[k], ctx->up_dma));
        HIP_CHECK(hipStreamWaitEvent(ctx->up_tr, ctx->up_landed[k], 0));
        hipLaunchKernelGGL(open::transpose_slab, dim3(div_up(width, 32), div_up(rows, 32)), dim3(32, 8), 0, ctx->up_tr,
                           (const uint32_t*)ctx->up_stage[k], m->d, rows, width, r0, height);
        LAUNCH_CHECK();
        HIP_CHECK(hipEventRecord(ctx->up_freed[k], ctx->up_tr));
        ctx->up_freed_set[k] = true;
      }
    }
    HIP_CHECK(hipEventRecord(m->ready, ctx->up_tr));
  } catch (...) {
    if (m->ready) (void)hipEventDestroy(m->ready);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  return m;
}

int zkm_matrix_upload_async(zkm_ctx* ctx, const uint32_t* host, size_t height, size_t width, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  *out = upload_async(ctx, host, height, width);
  API_END
}
int zkm_matrix_wait(zkm_ctx* ctx, const zkm_matrix* m) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (m->ready) HIP_CHECK(hipEventSynchronize(m->ready));
  API_END
}
int zkm_matrix_upload(zkm_ctx* ctx, const uint32_t* host, size_t height, size_t width, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  zkm_matrix* m = upload_async(ctx, host, height, width);
  HIP_CHECK(hipEventSynchronize(m->ready));  // the host buffer is free again when this returns
  *out = m;
  API_END
}

static void download_colmajor(zkm_ctx* ctx, const uint32_t* d, size_t h, size_t w, uint32_t* host) {
  if (h * w == 0) return;
  uint32_t* stage = ctx->alloc_n<uint32_t>(h * w);
  hipLaunchKernelGGL(open::transpose, dim3(div_up(h, 32), div_up(w, 32)), dim3(32, 8), 0, ctx->stream, d, stage, w, h);
  LAUNCH_CHECK();
  HIP_CHECK(hipMemcpyAsync(host, stage, h * w * 4, hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync(ctx->stream);
  ctx->release(stage);
}

int zkm_matrix_download(zkm_ctx* ctx, const zkm_matrix* m, uint32_t* host) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  wait_ready(ctx->stream, *m);
  download_colmajor(ctx, m->d, m->h, m->w, host);
  API_END
}
size_t zkm_matrix_height(const zkm_matrix* m) { return m->h; }
size_t zkm_matrix_width(const zkm_matrix* m) { return m->w; }
void zkm_matrix_free(zkm_ctx* ctx, zkm_matrix* m) {
  if (!m || !HandleTable::get().is_live(m, H_MATRIX)) return;      // a second free of the same handle is ignored, not a double delete
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (m->ready) {
    (void)hipEventSynchronize(m->ready);  // never hand a buffer back to the pool while its upload is in flight
    (void)hipEventDestroy(m->ready);
  }
  if (m->owned) ctx->release(m->d);
  delete m;
}

// Executor events ahead of time: the copy is queued on the DMA stream and the call returns; a later zkm_tracegen_* call that is given the
// returned address instead of a host pointer finds the events in HBM. Issued right before zkm_prove_shard of shard i for the events of
// shard i + 1, the PCIe transfer runs under shard i's kernels (one context, one host thread; the copy engine and the compute queue do
// not contend). `host` should be page-locked (zkm_host_alloc) — a pageable source makes the runtime stage the copy synchronously.
int zkm_events_upload_async(zkm_ctx* ctx, const void* host, size_t bytes, void** device_out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (bytes && !host) throw std::runtime_error("zkm_events_upload_async: null events");
  if (!ctx->ev_dma) HIP_CHECK(hipStreamCreateWithFlags(&ctx->ev_dma, hipStreamNonBlocking));
  void* d = ctx->alloc(std::max<size_t>(bytes, 4));
  hipEvent_t landed = nullptr;
  try {
    HIP_CHECK(hipEventCreateWithFlags(&landed, hipEventDisableTiming));
    // pool buffers come back in compute-stream order: the DMA stream first waits for what the compute stream has queued so far
    hipEvent_t reuse = ctx->get_event();
    HIP_CHECK(hipEventRecord(reuse, ctx->stream));
    HIP_CHECK(hipStreamWaitEvent(ctx->ev_dma, reuse, 0));
    ctx->event_pool.push_back(reuse);
    if (bytes) HIP_CHECK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->ev_dma));
    HIP_CHECK(hipEventRecord(landed, ctx->ev_dma));
  } catch (...) {
    if (landed) (void)hipEventDestroy(landed);
    ctx->release(d);
    throw;
  }
  ctx->prefetched[d] = {bytes, landed};
  *device_out = d;
  API_END
}
void zkm_events_free(zkm_ctx* ctx, void* device_events) {
  if (!ctx || !device_events) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->prefetched.find(device_events);
  if (it == ctx->prefetched.end()) return;
  (void)hipEventSynchronize(it->second.landed);   // never hand a buffer back to the pool while its upload is in flight
  (void)hipEventDestroy(it->second.landed);
  ctx->prefetched.erase(it);
  ctx->release(device_events);
}

int zkm_pcs_commit(zkm_ctx* ctx, size_t n_mats, const zkm_matrix* const* mats, const uint32_t* domain_shifts, uint32_t log_blowup,
                   uint32_t root_out[8], zkm_pcs_data** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_mats == 0) throw std::runtime_error("zkm_pcs_commit: empty batch");
  std::vector<zkm_matrix> ms;
  std::vector<uint32_t> sh;
  for (size_t i = 0; i < n_mats; i++) { ms.push_back(*mats[i]); if (domain_shifts) sh.push_back(domain_shifts[i]); }
  ctx->begin_timing();
  zkm_pcs_data* d = pcs_commit(ctx, ms, sh, (int)log_blowup);
  ctx->mark("pcs commit");
  ctx->end_timing(false);
  memcpy(root_out, d->root, 32);
  *out = d;
  API_END
}
void zkm_pcs_data_free(zkm_ctx* ctx, zkm_pcs_data* d) {
  if (!d) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  free_pcs_data(ctx, d);
}
int zkm_pcs_data_get_lde(zkm_ctx* ctx, const zkm_pcs_data* d, size_t idx, uint32_t* host) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (idx >= d->ldes.size()) throw std::runtime_error("matrix index out of range");
  download_colmajor(ctx, d->ldes[idx].d, d->ldes[idx].h, d->ldes[idx].w, host);
  API_END
}
int zkm_pcs_open_batch(zkm_ctx* ctx, const zkm_pcs_data* d, size_t index, uint32_t* values_out, uint32_t* proof_out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  const Tree& t = d->tree;
  if (index >= t.max_height) throw std::runtime_error("open_batch index out of range");
  std::vector<const uint32_t*> src;
  size_t nvals = 0;
  for (auto& m : d->ldes) {
    size_t row = index >> (t.log_max - log2_strict(m.h));
    for (size_t c = 0; c < m.w; c++) src.push_back(m.d + c * m.h + row);
    nvals += m.w;
  }
  for (int l = 0; l < t.log_max; l++)
    for (int k = 0; k < 8; k++) src.push_back(t.node(l, (index >> l) ^ 1) + k);
  const uint32_t** d_src = (const uint32_t**)ctx->alloc(src.size() * sizeof(void*));
  uint32_t* d_dst = ctx->alloc_n<uint32_t>(src.size());
  HIP_CHECK(hipMemcpyAsync(d_src, src.data(), src.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(open::gather_words, dim3(div_up(src.size(), open::THREADS)), dim3(open::THREADS), 0, ctx->stream,
                     (const uint32_t* const*)d_src, src.size(), d_dst);
  LAUNCH_CHECK();
  std::vector<uint32_t> host(src.size());
  HIP_CHECK(hipMemcpyAsync(host.data(), d_dst, src.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync(ctx->stream);
  memcpy(values_out, host.data(), nvals * 4);
  memcpy(proof_out, host.data() + nvals, (size_t)t.log_max * 32);
  ctx->release((void*)d_src);
  ctx->release(d_dst);
  API_END
}

int zkm_pk_setup(zkm_ctx* ctx, size_t n_prep, const zkm_matrix* const* prep_traces, const uint32_t* prep_local_only,
                 uint32_t pc_start, const uint32_t igcs[14], uint32_t log_blowup, zkm_pk** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  for (size_t i = 0; i < n_prep; i++) check_handle(prep_traces ? prep_traces[i] : nullptr, H_MATRIX, "zkm_pk_setup", "preprocessed trace");
  if (!igcs || !out) throw std::runtime_error("zkm_pk_setup: null argument");
  ctx->begin_call();
  zkm_pk* pk = new zkm_pk();
  pk->pc_start = pc_start;
  memcpy(pk->igcs, igcs, sizeof pk->igcs);
  memset(pk->commit, 0, sizeof pk->commit);
  for (size_t i = 0; i < n_prep; i++) { pk->prep.push_back(*prep_traces[i]); pk->local_only.push_back(prep_local_only[i]); }
  if (n_prep) {
    try { pk->data = pcs_commit(ctx, pk->prep, {}, (int)log_blowup); } catch (...) { delete pk; throw; }
    memcpy(pk->commit, pk->data->root, 32);
  }
  *out = pk;
  API_END
}
int zkm_pk_commitment(const zkm_pk* pk, uint32_t root_out[8]) { memcpy(root_out, pk->commit, 32); return 0; }
int zkm_pk_observe_into(const zkm_pk* pk, zkm_challenger* c) {
  chal::observe_slice(c, pk->commit, 8);
  chal::observe(c, pk->pc_start);
  chal::observe_slice(c, pk->igcs, 14);
  chal::observe(c, 0);
  return 0;
}
void zkm_pk_free(zkm_ctx* ctx, zkm_pk* pk) {
  if (!pk || !HandleTable::get().is_live(pk, H_PK)) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  free_pcs_data(ctx, pk->data);
  delete pk;
}

static zkm_main_data* commit_impl(zkm_ctx* ctx, size_t n_chips, const char* const* names, const zkm_matrix* const* traces,
                                  const uint32_t* pv, size_t n_pv, uint32_t log_blowup) {
  if (n_chips == 0) throw std::runtime_error("zkm_commit: no chips");
  zkm_main_data* md = new zkm_main_data();
  md->order.resize(n_chips);
  std::iota(md->order.begin(), md->order.end(), 0);
  // (Reverse(height), name) — prover.rs:264
  std::sort(md->order.begin(), md->order.end(), [&](size_t a, size_t b) {
    if (traces[a]->h != traces[b]->h) return traces[a]->h > traces[b]->h;
    return strcmp(names[a], names[b]) < 0;
  });
  for (size_t i : md->order) md->traces.push_back(*traces[i]);
  md->public_values.assign(pv, pv + n_pv);
  try { md->data = pcs_commit(ctx, md->traces, {}, (int)log_blowup); } catch (...) { delete md; throw; }
  return md;
}

int zkm_commit(zkm_ctx* ctx, size_t n_chips, const char* const* names, const zkm_matrix* const* main_traces,
               const uint32_t* public_values, size_t n_pv, uint32_t log_blowup, uint32_t main_commit_out[8], uint32_t* order_out,
               zkm_main_data** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (!names || !main_traces || !out || !main_commit_out) throw std::runtime_error("zkm_commit: null argument");
  for (size_t i = 0; i < n_chips; i++) check_handle(main_traces[i], H_MATRIX, "zkm_commit", "main trace");
  ctx->begin_timing();
  zkm_main_data* md = commit_impl(ctx, n_chips, names, main_traces, public_values, n_pv, log_blowup);
  ctx->mark("commit main");
  ctx->end_timing(false);
  memcpy(main_commit_out, md->data->root, 32);
  if (order_out) for (size_t i = 0; i < n_chips; i++) order_out[i] = (uint32_t)md->order[i];
  *out = md;
  API_END
}
void zkm_main_data_free(zkm_ctx* ctx, zkm_main_data* d) {
  if (!d || !HandleTable::get().is_live(d, H_MAIN_DATA)) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  free_pcs_data(ctx, d->data);
  delete d;
}

int zkm_open(zkm_ctx* ctx, const zkm_pk* pk, zkm_main_data* data, const zkm_chip_desc* chips, const zkm_fri_config* fri,
             uint32_t num_pv_elts, zkm_challenger* challenger, uint32_t* proof_out, size_t proof_cap, size_t* proof_len) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  check_handle(pk, H_PK, "zkm_open", "proving key");
  check_handle(data, H_MAIN_DATA, "zkm_open", "main data");
  if (!chips || !fri || !challenger || !proof_len) throw std::runtime_error("zkm_open: null argument");
  if (num_pv_elts > data->public_values.size()) throw std::runtime_error("num_pv_elts exceeds public_values length");
  Writer w(proof_out, proof_cap);
  ctx->begin_timing();
  // the transcript runs on a copy: the caller's challenger only advances once the proof has been handed over, so a call that
  // fails (a proof buffer that is too small: *proof_len says how many words it takes) leaves it where it was
  zkm_challenger ch = *challenger;
  open_impl(ctx, pk, data, chips, fri, num_pv_elts, &ch, w);
  ctx->end_timing(true);
  *proof_len = w.size();
  if (w.size() > proof_cap) throw std::runtime_error("proof buffer too small");
  *challenger = ch;
  API_END
}

int zkm_prove_shard(zkm_ctx* ctx, const zkm_pk* pk, size_t n_chips, const zkm_chip_desc* chips, const zkm_matrix* const* main_traces,
                    const uint32_t* public_values, size_t n_pv, const zkm_fri_config* fri, uint32_t num_pv_elts,
                    zkm_challenger* challenger, uint32_t* proof_out, size_t proof_cap, size_t* proof_len) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  check_handle(pk, H_PK, "zkm_prove_shard", "proving key");
  if (!chips || !main_traces || !fri || !challenger || !proof_len) throw std::runtime_error("zkm_prove_shard: null argument");
  for (size_t i = 0; i < n_chips; i++) check_handle(main_traces[i], H_MATRIX, "zkm_prove_shard", "main trace");
  if (num_pv_elts > n_pv) throw std::runtime_error("num_pv_elts exceeds public_values length");
  std::vector<const char*> names;
  for (size_t i = 0; i < n_chips; i++) names.push_back(chips[i].name);
  ctx->begin_timing();
  zkm_main_data* md = commit_impl(ctx, n_chips, names.data(), main_traces, public_values, n_pv, fri->log_blowup);
  ctx->mark("commit main");
  Writer w(proof_out, proof_cap);
  zkm_challenger ch = *challenger;   // as in zkm_open: the caller's transcript advances only with a delivered proof
  try {
    open_impl(ctx, pk, md, chips, fri, num_pv_elts, &ch, w);
  } catch (...) {
    free_pcs_data(ctx, md->data);
    delete md;
    throw;
  }
  free_pcs_data(ctx, md->data);
  delete md;
  ctx->end_timing(false);
  *proof_len = w.size();
  if (w.size() > proof_cap) throw std::runtime_error("proof buffer too small");
  *challenger = ch;
  API_END
}

int zkm_poseidon2_permute_batch(zkm_ctx* ctx, uint32_t* states, size_t n) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n == 0) return 0;
  uint32_t* d = ctx->alloc_n<uint32_t>(n * 16);
  HIP_CHECK(hipMemcpyAsync(d, states, n * 64, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(merkle::permute_batch, dim3(div_up(n, merkle::THREADS)), dim3(merkle::THREADS), 0, ctx->stream, d, n);
  LAUNCH_CHECK();
  HIP_CHECK(hipMemcpyAsync(states, d, n * 64, hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync(ctx->stream);
  ctx->release(d);
  API_END
}

int zkm_poseidon2_permute_batch_int(zkm_ctx* ctx, uint32_t* states, size_t n) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n == 0) return 0;
  uint32_t* d = ctx->alloc_n<uint32_t>(n * 16);
  HIP_CHECK(hipMemcpyAsync(d, states, n * 64, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(merkle::permute_batch_int, dim3(div_up(n, merkle::THREADS)), dim3(merkle::THREADS), 0, ctx->stream, d, n);
  LAUNCH_CHECK();
  HIP_CHECK(hipMemcpyAsync(states, d, n * 64, hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync(ctx->stream);
  ctx->release(d);
  API_END
}

int zkm_coset_lde_batch(zkm_ctx* ctx, const uint32_t* host, size_t height, size_t width, uint32_t log_blowup, uint32_t lde_shift,
                        uint32_t* out) {
  API_BEGIN
  zkm_matrix* m = nullptr;
  if (zkm_matrix_upload(ctx, host, height, width, &m) != 0) throw std::runtime_error(g_err);
  {
    std::lock_guard<std::mutex> lk(ctx->mu);
    size_t N = height << log_blowup;
    uint32_t* l = ctx->alloc_n<uint32_t>(std::max<size_t>(N * width, 1));
    lde_columns(ctx, m->d, height, width, (int)log_blowup, lde_shift, l);
    download_colmajor(ctx, l, N, width, out);
    ctx->release(l);
  }
  zkm_matrix_free(ctx, m);
  API_END
}

// One chip's permutation trace on its own (the step `open` runs between the main and the permutation commitments): a fine-grained entry point
// for parity tests, like zkm_coset_lde_batch.
int zkm_permutation_trace(zkm_ctx* ctx, const zkm_chip_desc* chip, const zkm_matrix* main, const zkm_matrix* prep, const uint32_t challenges[8], zkm_matrix** out,
                          uint32_t local_sum[4]) {
  API_BEGIN
  if (!chip || !main || !challenges || !out || !local_sum) throw std::runtime_error("zkm_permutation_trace: null argument");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (main->w != chip->main_width) throw std::runtime_error("zkm_permutation_trace: chip main_width does not match its trace");
  if (chip->prep_width && (!prep || prep->w != chip->prep_width || prep->h != main->h)) throw std::runtime_error("zkm_permutation_trace: the chip's preprocessed trace is missing or of another shape");
  const ChipMeta c = chip_meta(chip, main->h, (size_t)-1);
  zkm_matrix* pt = new zkm_matrix();
  pt->h = c.n; pt->w = (size_t)c.perm_ext_w * 4;
  std::vector<void*> scratch;
  try {
    pt->d = ctx->alloc_n<uint32_t>(std::max<size_t>(pt->h * pt->w, 1));
    for (int e = 0; e < 4; e++) local_sum[e] = 0;
    if (c.perm_ext_w > 0) {
      E4 alpha, beta;
      std::vector<stark::ScanJob> scans;
      memcpy(alpha.c, challenges, 16); memcpy(beta.c, challenges + 4, 16);
      std::vector<E4> bp(c.max_values + 2);
      bp[0] = kb::eone();
      for (size_t i = 1; i < bp.size(); i++) bp[i] = kb::emul(bp[i - 1], beta);
      const E4* d_bp = (const E4*)ctx->upload(bp.data(), bp.size() * sizeof(E4), &scratch);
      const uint32_t* d_blob = (const uint32_t*)ctx->upload(chip->lookups, chip->lookups_len * 4, &scratch);
      launch_permutation_trace(ctx, c, d_blob, (const uint32_t*)main->d, chip->prep_width ? (const uint32_t*)prep->d : nullptr, alpha, d_bp, *pt,
                               [&](size_t bytes) { void* p = ctx->alloc(bytes); scratch.push_back(p); return p; }, scans);
      launch_scans(ctx, scans, &scratch);
      const uint32_t* last = pt->d + (size_t)(c.perm_ext_w - 1) * 4 * c.n;
      for (int e = 0; e < 4; e++) HIP_CHECK(hipMemcpyAsync(local_sum + e, last + (size_t)e * c.n + (c.n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    ctx->sync(ctx->stream);
  } catch (...) {
    for (void* p : scratch) ctx->release(p);
    if (pt->d) ctx->release(pt->d);
    delete pt;
    throw;
  }
  for (void* p : scratch) ctx->release(p);
  *out = pt;
  API_END
}

#include "api_tracegen.hpp"

void zkm_challenger_init(zkm_challenger* c) { memset(c, 0, sizeof *c); }
void zkm_challenger_observe(zkm_challenger* c, const uint32_t* values, size_t n) { chal::observe_slice(c, values, n); }
uint32_t zkm_challenger_sample(zkm_challenger* c) { return chal::sample(c); }
uint32_t zkm_challenger_sample_bits(zkm_challenger* c, uint32_t bits) { return chal::sample_bits(c, bits); }

// host-side arithmetic of the transcript layer, exposed for the CPU-only parity tests
void zkm_host_poseidon2_permute(uint32_t state[16]) { p2::permute_host(state); }
void zkm_host_poseidon2_permute_f64(uint32_t state[16]) { p2f::permute_host_words(state); }
void zkm_host_poseidon2_f64_sponge(const uint32_t* words, size_t n, uint32_t digest[8]) { p2f::sponge_host(words, n, digest); }
void zkm_host_poseidon2_f64_compress_inject(const uint32_t left[8], const uint32_t right[8], const uint32_t* row, size_t n, uint32_t out[8]) {
  p2f::compress_inject_host(left, right, row, n, out);
}
void zkm_host_poseidon2_f64_audit(double out[7], int reset) {
  p2f::Audit& a = p2f::audit();
  out[0] = a.in; out[1] = a.lane_sum; out[2] = a.sbox_in; out[3] = a.lane; out[4] = a.frac_sum; out[5] = a.inexact; out[6] = a.sbox_fast_in;
  if (reset) a = p2f::Audit();
}
void zkm_host_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]) {
  E4 r = kb::emul(E4{{a[0], a[1], a[2], a[3]}}, E4{{b[0], b[1], b[2], b[3]}});
  memcpy(out, r.c, 16);
}
void zkm_host_ext_inv(const uint32_t a[4], uint32_t out[4]) {
  E4 r = kb::einv(E4{{a[0], a[1], a[2], a[3]}});
  memcpy(out, r.c, 16);
}
uint32_t zkm_host_field_mul(uint32_t a, uint32_t b) { return kb::mul(a, b); }
uint32_t zkm_host_field_inv(uint32_t a) { return kb::inv(a); }
uint32_t zkm_host_reduce96_bounded(uint32_t hi, uint64_t lo) { return kb::reduce96_bounded(hi, lo); }
uint32_t zkm_host_two_adic_generator(uint32_t bits) { return kb::two_adic_generator((int)bits); }

}  // extern "C"
