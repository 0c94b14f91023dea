// Part of libzkm_hip.so's host side (one translation unit: csrc/zkm_hip.hip includes this file). The context: error reporting, launch / timing macros, the host-side duplex challenger, the device memory pool and pinned staging ring, and the handle types of the C ABI.
#pragma once
static thread_local std::string g_err;

#define HIP_CHECK(expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                               std::to_string(__LINE__) + ")");                                  \
  } while (0)
#define LAUNCH_CHECK() HIP_CHECK(hipGetLastError())
// launch on the context's stream, bracketed by HIP events; `bytes` = compulsory HBM bytes of this
// launch (each input and output array counted once) for the roofline report.
#define KLAUNCH(ctx, name, bytes, kernel, grid, block, lds, ...)                                        \
  do {                                                                                                   \
    (ctx)->flush_staged();                                                                               \
    if ((ctx)->kbegin(name, (double)(bytes))) {                                                          \
      /* start/stop timestamps ride on the dispatch's own completion signal: no extra barrier packets */ \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, (ctx)->cur, (ctx)->krecs.back().start,             \
                            (ctx)->krecs.back().stop, 0, __VA_ARGS__);                                   \
    } else {                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->cur, __VA_ARGS__);                             \
    }                                                                                                    \
    LAUNCH_CHECK();                                                                                      \
  } while (0)

static inline int log2_strict(size_t n) {
  int k = 0;
  while (((size_t)1 << k) < n) k++;
  if (((size_t)1 << k) != n) throw std::runtime_error("height is not a power of two");
  return k;
}
static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
static inline uint64_t fnv1a(const uint32_t* w, size_t n);
// registry key of a chip's permutation-trace kernel: its lookups blob and the batch size (2^log_quotient_degree lookups per column)
static inline uint64_t perm_key(const uint32_t* lookups, size_t len, uint32_t log_quotient_degree) {
  return fnv1a(lookups, len) * 1099511628211ull + log_quotient_degree + 1;
}
static inline uint64_t fnv1a(const uint32_t* w, size_t n) {
  uint64_t h = 1469598103934665603ull;
  const unsigned char* p = (const unsigned char*)w;
  for (size_t i = 0; i < n * 4; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}

// ---- host transcript: DuplexChallenger<KoalaBear, Poseidon2, 16, 8> -----------------------------
// crates/recursion/circuit/src/challenger.rs:90-114,201-233
namespace chal {
static void duplexing(zkm_challenger* c) {
  for (uint32_t i = 0; i < c->num_inputs; i++) c->sponge_state[i] = c->input_buffer[i];
  c->num_inputs = 0;
  p2::permute_host(c->sponge_state);
  for (int i = 0; i < 8; i++) c->output_buffer[i] = c->sponge_state[i];
  c->num_outputs = 8;
}
static void observe(zkm_challenger* c, uint32_t v) {
  c->num_outputs = 0;
  c->input_buffer[c->num_inputs++] = v;
  if (c->num_inputs == 8) duplexing(c);
}
static void observe_slice(zkm_challenger* c, const uint32_t* v, size_t n) {
  for (size_t i = 0; i < n; i++) observe(c, v[i]);
}
static void observe_ext(zkm_challenger* c, const E4& e) { observe_slice(c, e.c, 4); }
static uint32_t sample(zkm_challenger* c) {
  if (c->num_inputs != 0 || c->num_outputs == 0) duplexing(c);
  return c->output_buffer[--c->num_outputs];
}
static E4 sample_ext(zkm_challenger* c) {
  E4 e;
  for (int i = 0; i < 4; i++) e.c[i] = sample(c);
  return e;
}
static uint32_t sample_bits(zkm_challenger* c, uint32_t bits) {
  return kb::from_monty(sample(c)) & ((1u << bits) - 1);
}
}  // namespace chal

// ---- live handles --------------------------------------------------------------------------------
// Every object the C ABI hands out as an opaque pointer (zkm_matrix, zkm_pcs_data, zkm_pk, zkm_main_data) enters this table when it is
// made and leaves it when it is destroyed, so that an entry point can tell a live handle from one that was freed (or never was one)
// WITHOUT dereferencing it: a freed handle reused is an error message, not undefined behaviour (SURVEY 8b "Errors").
enum HandleKind { H_MATRIX = 1, H_PCS_DATA = 2, H_PK = 3, H_MAIN_DATA = 4 };
struct HandleTable {
  std::mutex mu;
  std::unordered_map<const void*, int> live;
  static HandleTable& get() { static HandleTable t; return t; }
  void add(const void* p, int kind) { std::lock_guard<std::mutex> lk(mu); live[p] = kind; }
  void drop(const void* p) { std::lock_guard<std::mutex> lk(mu); live.erase(p); }
  bool is_live(const void* p, int kind) { std::lock_guard<std::mutex> lk(mu); auto it = live.find(p); return it != live.end() && it->second == kind; }
};
template <int KIND>
struct Handle {
  Handle() { HandleTable::get().add(this, KIND); }
  Handle(const Handle&) { HandleTable::get().add(this, KIND); }
  Handle& operator=(const Handle&) { return *this; }
  ~Handle() { HandleTable::get().drop(this); }
};
static inline void check_handle(const void* p, int kind, const char* fn, const char* what) {
  if (!p) throw std::runtime_error(std::string(fn) + ": null " + what);
  if (!HandleTable::get().is_live(p, kind)) throw std::runtime_error(std::string(fn) + ": " + what + " is not a live handle (freed, or never returned by this library)");
}

// ---- context -----------------------------------------------------------------------------------
struct zkm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;   // main stream: phases, transcript round trips
  hipStream_t stream2 = nullptr;  // side stream (kept for experiments: hashing a tree beside the LDEs did not pay)
  // asynchronous uploads: DMA stream, transpose stream, two persistent staging slabs and their "free again" events
  hipStream_t up_dma = nullptr, up_tr = nullptr;
  uint32_t* up_stage[2] = {nullptr, nullptr};
  hipEvent_t up_freed[2] = {nullptr, nullptr}, up_landed[2] = {nullptr, nullptr};
  bool up_freed_set[2] = {false, false};
  int up_next = 0;
  static constexpr size_t UP_SLAB_BYTES = (size_t)32 << 20;
  hipStream_t cur = nullptr;      // where KLAUNCH / upload / kernel-timing events go right now
  // executor events copied ahead of their trace generation (zkm_events_upload_async): device address -> (bytes, "landed" event)
  hipStream_t ev_dma = nullptr;   // the DMA stream of zkm_events_upload_async (its own: the matrix uploads' streams come and go with their staging slabs)
  struct Prefetched { size_t bytes; hipEvent_t landed; };
  std::map<const void*, Prefetched> prefetched;
  // The device copy of a chip's events for a zkm_tracegen_* call. `events` is either a host pointer — copied now, on the compute
  // stream, into a pool buffer the caller releases (`owned` = true) — or an address zkm_events_upload_async returned: the compute
  // stream then only waits for that copy to have landed (the upload ran on the DMA stream, typically under the previous shard's proof).
  const uint32_t* events_on_device(const void* events, size_t bytes, uint32_t** owned) {
    *owned = nullptr;
    auto it = prefetched.find(events);
    if (it != prefetched.end()) {
      if (bytes > it->second.bytes) throw std::runtime_error("zkm_tracegen: more events asked for than zkm_events_upload_async copied");
      HIP_CHECK(hipStreamWaitEvent(stream, it->second.landed, 0));
      return (const uint32_t*)events;
    }
    uint32_t* d = (uint32_t*)alloc(std::max<size_t>(bytes, 4));
    *owned = d;
    if (bytes) HIP_CHECK(hipMemcpyAsync(d, events, bytes, hipMemcpyHostToDevice, stream));
    return d;
  }
  std::mutex mu;
  std::multimap<size_t, void*> free_list;  // caching allocator: exact-size reuse
  std::map<void*, size_t> live;
  std::map<int, uint32_t*> tw_fwd, tw_inv;  // stage-major twiddle tables (lde.cuh), per transform log-size
  std::vector<std::pair<std::string, hipEvent_t>> marks;
  std::vector<hipEvent_t> event_pool;
  std::vector<std::string> timing_names;
  std::vector<float> timing_ms;
  // per-kernel HIP-event timing on this stream (bench.py's roofline leg reads it)
  struct KRec { const char* name; double bytes; hipEvent_t start, stop; };
  struct KStat { double ms = 0, bytes = 0; uint32_t calls = 0; };
  std::vector<KRec> krecs;
  std::map<std::string, KStat> kstats;
  // 0: off; 1: every launch; 2 (default): only launches moving >= 256 KiB (the ~300 tiny launches of a proof
  // are left untimed)
  int kernel_timing = 2;
  // pcs_commit extends a commit's shorter matrices on the side stream under the leaf hashing of the tallest (zkm_ctx_set_lde_overlap;
  // ZKM_LDE_OVERLAP=0 turns it off for every context of the process). Off, every kernel of a proof runs alone on the main stream and
  // the per-kernel HIP-event durations add up to the busy time.
  bool lde_overlap = !(getenv("ZKM_LDE_OVERLAP") && atoi(getenv("ZKM_LDE_OVERLAP")) == 0);
  // build_tree: the rows of a commit's shorter heights hashed by one launch in front of the tree levels (merkle::hash_rows) instead of
  // inside compress_layer; ZKM_ROWS_UP_FRONT=0 for the A/B
  bool rows_up_front = !(getenv("ZKM_ROWS_UP_FRONT") && atoi(getenv("ZKM_ROWS_UP_FRONT")) == 0);
  bool root_poll = !(getenv("ZKM_ROOT_POLL") && atoi(getenv("ZKM_ROOT_POLL")) == 0);   // wait_root; cleared only when a FINISHED launch's root is still not visible
  int root_spin_before_yield = getenv("ZKM_ROOT_SPIN") ? atoi(getenv("ZKM_ROOT_SPIN")) : 4096;   // wait_root: spins before it starts yielding the core
  int root_sleep_ns = getenv("ZKM_ROOT_SLEEP_NS") ? atoi(getenv("ZKM_ROOT_SLEEP_NS")) : 0;        // > 0: past the spins it sleeps this long between looks instead of yielding
  // How this context's host thread waits for its stream (zkm_ctx_set_host_wait; ZKM_HOST_WAIT=blocking sets the default). Spinning
  // (hipStreamSynchronize, and wait_root watching the root arrive) is the lowest latency and costs a core per lane for the whole proof;
  // blocking sleeps on an interrupt-backed event: ~1 ms more per proof for one lane alone, nothing measurable with two lanes per GPU (the
  // other lane's kernels fill the wake-up latency), a tenth of the CPU time. A farm rank runs its lanes blocking.
  bool host_wait_blocking = getenv("ZKM_HOST_WAIT") && !strcmp(getenv("ZKM_HOST_WAIT"), "blocking");
  int wait_sleep_ns = getenv("ZKM_WAIT_SLEEP_NS") ? atoi(getenv("ZKM_WAIT_SLEEP_NS")) : 20000;
  // (an event created with hipEventBlockingSync does not make hipEventSynchronize sleep on this runtime — measured: the lane thread still
  // burns a core; only the process-wide hipDeviceScheduleBlockingSync does, and that would bind every context of the process. So the
  // blocking wait is a query of the stream between short sleeps: a few per cent of a core, ~40 us of wake-up latency per wait.)
  // The sleeping waits need a timer slack of ~1 us (the default 50 us would triple a 20 us sleep). The slack belongs to the CALLER's
  // thread — a Rust prover thread, a Python lane — so it is tightened only while one of this library's waits is actually sleeping and put
  // back when that wait ends: an embedding application's own timers are never left changed.
  struct TimerSlack {
    long saved = -1;
    void tighten() {
      if (saved >= 0) return;
      saved = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
      if (saved >= 0) prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    }
    ~TimerSlack() { if (saved >= 0) prctl(PR_SET_TIMERSLACK, (unsigned long)saved, 0, 0, 0); }
  };
  static void sleep_ns(long ns, TimerSlack& slack) {
    slack.tighten();
    timespec ts{0, ns};
    nanosleep(&ts, nullptr);
  }
  // host_wait_blocking: not a kernel-level blocking wait (see above: this runtime has none per stream) but a stream query between short sleeps
  void sync(hipStream_t s) {
    if (!host_wait_blocking) { HIP_CHECK(hipStreamSynchronize(s)); return; }
    TimerSlack slack;
    for (int i = 0;; i++) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) return;
      if (e != hipErrorNotReady) HIP_CHECK(e);
      if (i >= 4) sleep_ns(wait_sleep_ns, slack);
    }
  }
  // per-chip specialised quotient kernels (ziren_amd/codegen.py), keyed by a hash of the program words
  std::map<uint64_t, std::vector<hipFunction_t>> quotient_fns;   // a program's specialised kernel(s), in launch order
  std::map<uint64_t, hipFunction_t> quotient_uniform_fns;         // ... and, where its code object has one, the kernel that fills QuotientArgs::uniforms
  std::map<uint64_t, hipFunction_t> perm_fns;                    // a lookups blob's specialised permutation-trace kernel (key: perm_key)
  std::vector<hipModule_t> modules;
  hipEvent_t get_event() {
    hipEvent_t e;
    if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); }
    else HIP_CHECK(hipEventCreate(&e));
    return e;
  }
  // returns true when this launch is to be timed; the record then holds the two events to pass to the launch
  std::string timing_only;      // kernel_timing == 3: the one kernel name that is timed
  bool kbegin(const char* name, double bytes) {
    if (!(kernel_timing == 1 || (kernel_timing == 2 && bytes >= 262144.0) || (kernel_timing == 3 && timing_only == name))) return false;
    KRec r{name, bytes, get_event(), get_event()};
    krecs.push_back(r);
    return true;
  }

  // pinned host ring: short-lived host data goes H2D (and small results come D2H) through it without
  // a stream synchronisation; it is recycled at the start of every top-level call, when the stream is idle.
  char* pin = nullptr;
  size_t pin_cap = (size_t)32 << 20, pin_off = 0;
  void* pin_alloc(size_t bytes) {
    // coherent (fine-grained) whatever HIP_HOST_COHERENT says: wait_root watches words a running kernel stores into this ring
    if (!pin) HIP_CHECK(hipHostMalloc((void**)&pin, pin_cap, hipHostMallocCoherent));
    size_t off = (pin_off + 63) & ~(size_t)63;
    if (off + bytes > pin_cap) return nullptr;
    pin_off = off + bytes;
    return pin + off;
  }
  void begin_call() {
    sync(stream);
    sync(stream2);
    cur = stream;
    side_join();               // both streams are idle here: only the deferred scratch is left to hand back
    pin_off = 0;
    dirty_lo = SIZE_MAX; dirty_hi = 0;
  }
  // Small phase-local host tables (challenge powers, per-chip constants, descriptor and pointer arrays: ~100 per proof) are staged:
  // written into the pinned ring, handed out as addresses in a device mirror of the ring, and copied over in ONE transfer per run of
  // uploads, right before the next kernel launch (KLAUNCH flushes; the few direct launch sites call flush_staged themselves). One copy
  // dispatch per phase instead of one per table. The address stays valid until the next top-level call (begin_call rewinds the ring);
  // release() ignores it.
  char* arena = nullptr;
  size_t dirty_lo = SIZE_MAX, dirty_hi = 0;
  void* upload_staged(const void* src, size_t bytes, std::vector<void*>* scratch = nullptr) {
    const size_t room = bytes ? bytes : 4;             // an empty table still gets an address of its own
    void* h = room <= ((size_t)1 << 20) ? pin_alloc(room) : nullptr;
    if (!h) return upload(src, bytes, scratch);        // too large for the ring: its own buffer and copy
    if (!arena) HIP_CHECK(hipMalloc((void**)&arena, pin_cap));
    if (bytes) memcpy(h, src, bytes);
    bytes = room;
    const size_t off = (char*)h - pin;
    dirty_lo = std::min(dirty_lo, off);
    dirty_hi = std::max(dirty_hi, off + bytes);
    return arena + off;
  }
  void flush_staged() {
    if (dirty_hi <= dirty_lo) return;
    // the span [dirty_lo, dirty_hi) may also cover ring blocks that are not staged tables (download_async destinations, upload() staging):
    // copying them to the arena is harmless only while everything that touches the ring runs on the one main stream — so the copy is
    // always made there. When the launch that needs the tables goes to the side stream (pcs_commit extends a commit's shorter matrices
    // there), that stream then waits for the main stream up to this point: the copy, the first-use table fills and whatever produced
    // the kernel's inputs are all in front of it.
    HIP_CHECK(hipMemcpyAsync(arena + dirty_lo, pin + dirty_lo, dirty_hi - dirty_lo, hipMemcpyHostToDevice, stream));
    dirty_lo = SIZE_MAX; dirty_hi = 0;
    if (cur != stream) {
      hipEvent_t e = get_event();
      HIP_CHECK(hipEventRecord(e, stream));
      HIP_CHECK(hipStreamWaitEvent(cur, e, 0));
      event_pool.push_back(e);
    }
  }
  // Work queued on the side stream and not yet joined: scratch it uses goes back to the pool (which hands buffers out in main-stream
  // order) only after the main stream has waited for it.
  std::vector<void*> side_deferred;
  hipEvent_t side_done = nullptr;
  bool side_pending = false;
  void release_here(void* p) {            // release() for scratch of a launch sequence that may run on the side stream
    if (cur != stream) side_deferred.push_back(p);
    else release(p);
  }
  // Everything queued on the main stream so far is in front of the side stream's work: the producers of its inputs (cflags memset,
  // perm_rows / scan kernels, first-use table fills), pool buffers released in main-stream order, and any staged table copy.
  void side_begin() {
    hipEvent_t e = get_event();
    HIP_CHECK(hipEventRecord(e, stream));
    HIP_CHECK(hipStreamWaitEvent(stream2, e, 0));
    event_pool.push_back(e);
    cur = stream2;
  }
  void side_end() {
    if (!side_done) HIP_CHECK(hipEventCreateWithFlags(&side_done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(side_done, stream2));
    cur = stream;
    side_pending = true;
  }
  void side_join() {
    if (!side_pending) return;
    HIP_CHECK(hipStreamWaitEvent(stream, side_done, 0));
    for (void* p : side_deferred) release(p);
    side_deferred.clear();
    side_pending = false;
  }
  // copy `bytes` of host data to a fresh device buffer; the source may die as soon as this returns
  void* upload(const void* src, size_t bytes, std::vector<void*>* scratch) {
    void* d = alloc(bytes);
    if (scratch) scratch->push_back(d);
    if (bytes == 0) return d;
    void* h = pin_alloc(bytes);
    if (h) {
      memcpy(h, src, bytes);
      HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, cur));
    } else {
      HIP_CHECK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, cur));
      sync(cur);
    }
    return d;
  }
  // asynchronous D2H into the pinned ring; valid after the next stream synchronisation
  template <class T>
  T* download_async(const T* dev, size_t count) {
    T* h = (T*)pin_alloc(count * sizeof(T));
    if (!h) throw std::runtime_error("pinned staging ring exhausted");
    HIP_CHECK(hipMemcpyAsync(h, dev, count * sizeof(T), hipMemcpyDeviceToHost, stream));
    return h;
  }

  // Pool accounting and its cap: `pool_bytes` = everything this pool holds from hipMalloc (handed out + cached). `pool_limit` (0: none;
  // zkm_ctx_set_memory_limit, or ZKM_POOL_LIMIT_MB for every context of the process) bounds it: a request that would cross it first gives the
  // cached blocks back to the driver and, if that is not enough, fails with an error the caller can act on — the context stays usable.
  size_t pool_bytes = 0;
  size_t pool_limit = getenv("ZKM_POOL_LIMIT_MB") ? (size_t)atoll(getenv("ZKM_POOL_LIMIT_MB")) << 20 : 0;
  // buffers handed out since the outermost API call on this context began (CallScope): what a failing call gives back
  std::vector<void*> call_allocs;
  int call_depth = 0;
  void drop_cached() {            // stream-ordered reuse is over for these: wait for the work that may still touch them
    if (free_list.empty()) return;
    (void)hipStreamSynchronize(stream);
    if (stream2) (void)hipStreamSynchronize(stream2);
    for (auto& kv : free_list) { (void)hipFree(kv.second); pool_bytes -= kv.first; }
    free_list.clear();
  }
  void* alloc(size_t bytes) {
    if (bytes == 0) bytes = 4;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_list.find(bytes);
    void* p;
    if (it != free_list.end()) {
      p = it->second;
      free_list.erase(it);
    } else {
      if (pool_limit && pool_bytes + bytes > pool_limit) drop_cached();
      if (pool_limit && pool_bytes + bytes > pool_limit)
        throw std::runtime_error("out of device memory: " + std::to_string(bytes) + " bytes on top of " + std::to_string(pool_bytes) +
                                 " held would cross the context's pool limit of " + std::to_string(pool_limit) + " bytes");
      hipError_t e = hipMalloc(&p, bytes);
      if (e != hipSuccess) {        // the device is full: hand the cached blocks back and try once more
        (void)hipGetLastError();
        drop_cached();
        e = hipMalloc(&p, bytes);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        throw std::runtime_error(std::string("out of device memory: hipMalloc of ") + std::to_string(bytes) + " bytes failed (" + hipGetErrorString(e) + "), " +
                                 std::to_string(pool_bytes) + " bytes held by this context's pool");
      }
      pool_bytes += bytes;
    }
    live[p] = bytes;
    if (call_depth > 0) call_allocs.push_back(p);
    return p;
  }
  template <class T>
  T* alloc_n(size_t n) { return (T*)alloc(n * sizeof(T)); }
  // stream-ordered: buffers are only reused by later work on the same stream
  void release(void* p) {
    if (!p) return;
    auto it = live.find(p);
    if (it == live.end()) return;
    free_list.insert({it->second, p});
    live.erase(it);
  }
  void mark(const char* name) {
    hipEvent_t e = get_event();
    HIP_CHECK(hipEventRecord(e, stream));
    marks.push_back({name, e});
  }
  void begin_timing() {
    begin_call();
    for (auto& m : marks) event_pool.push_back(m.second);
    marks.clear();
    for (auto& r : krecs) { event_pool.push_back(r.start); if (r.stop) event_pool.push_back(r.stop); }
    krecs.clear();
    mark("begin");
  }
  void end_timing(bool append) {
    sync(stream);
    sync(stream2);
    if (!append) { timing_names.clear(); timing_ms.clear(); kstats.clear(); }
    for (auto& r : krecs) {
      float ms = 0;
      if (r.stop) HIP_CHECK(hipEventElapsedTime(&ms, r.start, r.stop));
      KStat& k = kstats[r.name];
      k.ms += ms; k.bytes += r.bytes; k.calls++;
    }
    for (size_t i = 1; i < marks.size(); i++) {
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second));
      timing_names.push_back(marks[i].first);
      timing_ms.push_back(ms);
    }
  }
  const uint32_t* twiddles(int log_size, bool inverse) {
    auto& tab = inverse ? tw_inv : tw_fwd;
    auto it = tab.find(log_size);
    if (it != tab.end()) return it->second;
    size_t count = (size_t)1 << log_size;  // stage-major table: n - 1 entries
    uint32_t* d;
    HIP_CHECK(hipMalloc(&d, count * 4));
    uint32_t w = kb::two_adic_generator(log_size);
    if (inverse) w = kb::inv(w);
    if (log_size > 0) {
      hipLaunchKernelGGL(lde::fill_stage_twiddles, dim3(div_up(count / 2, 256), log_size), dim3(256), 0, stream, d, w, log_size);
      LAUNCH_CHECK();
    }
    tab[log_size] = d;
    return d;
  }
  // sub-problem twiddles (lde::fill_ct_twiddles) of the strided passes' A-point transforms
  std::map<int, uint32_t*> ct_fwd, ct_inv;
  const uint32_t* twiddles_ct(int log_size, bool inverse) {
    auto& tab = inverse ? ct_inv : ct_fwd;
    auto it = tab.find(log_size);
    if (it != tab.end()) return it->second;
    const size_t count = (size_t)1 << log_size;
    uint32_t* d;
    HIP_CHECK(hipMalloc(&d, count * 4));
    uint32_t w = kb::two_adic_generator(log_size);
    if (inverse) w = kb::inv(w);
    hipLaunchKernelGGL(lde::fill_ct_twiddles, dim3(div_up(count, 256)), dim3(256), 0, stream, d, w, log_size);
    LAUNCH_CHECK();
    tab[log_size] = d;
    return d;
  }
  // w_n^e by two lookups (lde::Group::pw_lo / pw_hi), per log2 n >= 10
  std::map<int, std::pair<uint32_t*, uint32_t*>> pow_tabs;
  std::pair<const uint32_t*, const uint32_t*> pow_tables(int k) {
    auto it = pow_tabs.find(k);
    if (it != pow_tabs.end()) return {it->second.first, it->second.second};
    const uint32_t n_hi = (uint32_t)(((size_t)1 << k) >> 10);
    uint32_t *lo, *hi;
    HIP_CHECK(hipMalloc(&lo, 1024 * 4));
    HIP_CHECK(hipMalloc(&hi, std::max<uint32_t>(n_hi, 1) * 4));
    hipLaunchKernelGGL(lde::fill_pow_tables, dim3(div_up(std::max<uint32_t>(1024, n_hi), 256)), dim3(256), 0, stream, lo, hi,
                       kb::two_adic_generator(k), n_hi);
    LAUNCH_CHECK();
    pow_tabs[k] = {lo, hi};
    return {lo, hi};
  }
  // the quotient kernels' selector table per (log2 n, log quotient degree): stark::fill_selectors, 3 Q words
  std::map<std::pair<int, int>, uint32_t*> selector_tabs;
  const uint32_t* selectors(int log_n, int lqd, uint32_t w_q, uint32_t g_inv, const uint32_t* d_consts) {
    auto key = std::make_pair(log_n, lqd);
    auto it = selector_tabs.find(key);
    if (it != selector_tabs.end()) return it->second;
    const size_t Q = (size_t)1 << (log_n + lqd);
    uint32_t* d;
    HIP_CHECK(hipMalloc(&d, 3 * Q * 4));
    flush_staged();
    hipLaunchKernelGGL(stark::fill_selectors, dim3(div_up(Q, 256)), dim3(256), 0, cur, d, log_n + lqd, lqd, w_q, g_inv, d_consts);
    LAUNCH_CHECK();
    selector_tabs[key] = d;
    return d;
  }
  // per row of the four-step decomposition the twiddles of the inverse row transform (lde::fill_row_twiddles): n words per height
  std::map<int, uint32_t*> row_tabs;
  const uint32_t* row_twiddles(int k) {
    auto it = row_tabs.find(k);
    if (it != row_tabs.end()) return it->second;
    const int lb = lde::LOG_ROW_MAX, la = k - lb;
    const size_t n = (size_t)1 << k;
    uint32_t* d;
    HIP_CHECK(hipMalloc(&d, n * 4));
    auto pw = pow_tables(k);
    hipLaunchKernelGGL(lde::fill_row_twiddles, dim3(div_up(n, 256)), dim3(256), 0, stream, d, la, lb, pw.first, pw.second);
    LAUNCH_CHECK();
    row_tabs[k] = d;
    return d;
  }
  // the tables of lde_rows_big that depend on the coset shift: per coset the scaled forward stage twiddles of the B-point row
  // transform and the row constants shift_j^k1 / n; keyed by (log2 n, log_blowup, shift of coset 0). A prover meets a handful of
  // (height, shift) pairs — traces are extended onto 3 K, quotient chunks onto 3 w_2n^-i K — so the cache stays small; zkm_ctx_trim clears it.
  struct CosetTabs { uint32_t* twf; uint32_t* cs; };
  std::map<std::tuple<int, int, uint32_t>, CosetTabs> coset_tabs;
  CosetTabs coset_tables(int k, int bl, uint32_t shift) {
    auto key = std::make_tuple(k, bl, shift);
    auto it = coset_tabs.find(key);
    if (it != coset_tabs.end()) return it->second;
    const int lb = lde::LOG_ROW_MAX, la = k - lb;
    const size_t B = (size_t)1 << lb, A = (size_t)1 << la, cosets = (size_t)1 << bl;
    CosetTabs t;
    HIP_CHECK(hipMalloc(&t.twf, cosets * B * 4));
    HIP_CHECK(hipMalloc(&t.cs, cosets * A * 4));
    const uint32_t w_N = kb::two_adic_generator(k + bl), w_B = kb::two_adic_generator(lb);
    const uint32_t n_inv = kb::inv(kb::to_monty((uint32_t)(((size_t)1 << k) % kb::P)));
    uint32_t sj = shift;
    for (size_t j = 0; j < cosets; j++) {
      uint32_t sA = sj;
      for (int i = 0; i < la; i++) sA = kb::sqr(sA);
      hipLaunchKernelGGL(lde::fill_scaled_stage_twiddles, dim3(div_up(B / 2, 256), lb), dim3(256), 0, stream, t.twf + j * B, w_B, lb, sA);
      LAUNCH_CHECK();
      sj = kb::mul(sj, w_N);
    }
    hipLaunchKernelGGL(lde::fill_row_scales, dim3(div_up(A * cosets, 256)), dim3(256), 0, stream, t.cs, shift, w_N, n_inv, (uint32_t)A, (uint32_t)cosets);
    LAUNCH_CHECK();
    coset_tabs[key] = t;
    return t;
  }
  void drop_coset_tables() {   // stream must be idle
    for (auto& kv : coset_tabs) { (void)hipFree(kv.second.twf); (void)hipFree(kv.second.cs); }
    coset_tabs.clear();
  }
};

struct zkm_matrix : Handle<H_MATRIX> {
  uint32_t* d = nullptr;  // column-major: column c at d + c * h
  size_t h = 0, w = 0;
  bool owned = true;
  bool uniform_rows = false;   // set by a trace generator that had no events: every row is the chip's padding row (lde::Mat::uniform)
  hipEvent_t ready = nullptr;  // set by zkm_matrix_upload_async: fires when the matrix is complete in HBM
};

// make `stream` wait until an asynchronously uploaded matrix is complete (no-op for any other matrix)
static inline void wait_ready(hipStream_t stream, const zkm_matrix& m) {
  if (m.ready) HIP_CHECK(hipStreamWaitEvent(stream, m.ready, 0));
}

struct zkm_byte_lookups {
  uint32_t* counts = nullptr;  // [NUM_BYTE_OPS][BYTE_ROWS] plain counters: record.byte_lookups on the device
};

struct Tree {
  const uint32_t* h_root = nullptr;     // set when the kernel that finished the tree wrote the root to page-locked host memory itself
  uint32_t* digests = nullptr;          // all layers, 8 words per digest
  std::vector<size_t> layer_off;        // in digests
  size_t max_height = 0;
  int log_max = 0;
  const uint32_t* node(int layer, size_t i) const { return digests + (layer_off[layer] + i) * 8; }
};

struct zkm_pcs_data : Handle<H_PCS_DATA> {
  std::vector<zkm_matrix> ldes;           // owned; bit-reversed rows, height = h << log_blowup
  std::vector<const uint32_t*> evals;     // borrowed: the committed evaluations (column-major, natural order)
  std::vector<size_t> eval_heights;
  std::vector<uint32_t> domain_shifts;    // evals[i] live on domain_shifts[i] * H
  std::vector<zkm_matrix> owned_evals;    // evaluations owned by this object (perm traces, quotient chunks)
  Tree tree;
  uint32_t root[8];
  int log_blowup = 1;
  // lde::Mat::cflag of the four-step matrices, kept with the commitment (two words per column: [2c] = 0 while no two different words
  // were seen, [2c + 1] = the column's first word): the opening kernels read a constant column's word instead of the column
  uint32_t* cflags = nullptr;
  std::vector<const uint32_t*> col_flags;   // per matrix: into cflags, or null
};

struct zkm_pk : Handle<H_PK> {
  std::vector<zkm_matrix> prep;  // borrowed device matrices
  std::vector<uint32_t> local_only;
  zkm_pcs_data* data = nullptr;
  uint32_t commit[8];
  uint32_t pc_start;
  uint32_t igcs[14];
};

struct zkm_main_data : Handle<H_MAIN_DATA> {
  std::vector<size_t> order;             // sorted position -> caller index
  std::vector<zkm_matrix> traces;        // borrowed, sorted order
  zkm_pcs_data* data = nullptr;
  std::vector<uint32_t> public_values;
};

