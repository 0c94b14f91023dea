// Part of libzkm_hip.so's host side (one translation unit: csrc/zkm_hip.hip includes this file). The zkm_tracegen_* entry points (device trace generation); included inside the extern "C" block.
#pragma once
// ---- device trace generation (ALU chips) ---------------------------------------------------------------------
size_t zkm_tracegen_alu_width(int chip) { return chip >= 0 && chip < tracegen::NUM_ALU_CHIPS ? (size_t)tracegen::chip_width(chip) : 0; }

static size_t padded_trace_rows(size_t n_records, int fixed_log2_rows, const char* what) {
  // utils::next_power_of_two (crates/core/machine/src/utils/mod.rs): the shape's fixed size, else >= 16
  size_t height = 16;
  if (fixed_log2_rows >= 0) {
    if (fixed_log2_rows > 30) throw std::runtime_error(std::string(what) + ": fixed log2 rows out of range");
    height = (size_t)1 << fixed_log2_rows;
    if (n_records > height) throw std::runtime_error(std::string(what) + ": fixed log2 rows is too small");
  } else {
    while (height < n_records) height <<= 1;
  }
  return height;
}

// Trace generators queued back to back on the compute stream, with no host synchronisation between them: every generator below is an
// `enqueue_*` function that allocates its matrix, finds its events in HBM (a prefetched address: the stream waits for the copy; a host
// pointer: the copy is queued in front of the kernel) and launches. `finish` makes the one synchronisation of the batch, reads the error
// words the kernels may have set, and hands scratch back to the pool. The single-chip entry points are batches of one;
// zkm_tracegen_shard is a batch of a whole shard's chips (crates/stark/src/prover.rs:70-108, generate_traces).
struct TraceBatch {
  zkm_ctx* ctx;
  std::vector<void*> scratch;                 // event copies, programs, scan levels: released by finish / abort
  std::vector<zkm_matrix*> made;              // matrices of this batch: deleted by abort
  std::vector<std::vector<zkm_syscall_event>> host_keep;   // host-side filtered events: alive until the copy has run
  enum { F_CPU, F_GLOBAL, F_SYSCALL_CORE };
  static constexpr int MAX_FLAGS = 256;
  uint32_t* d_flags = nullptr;
  std::vector<int> flag_kind;
  explicit TraceBatch(zkm_ctx* c) : ctx(c) {}
  zkm_matrix* matrix(size_t h, size_t w) {
    zkm_matrix* m = new zkm_matrix();
    made.push_back(m);
    m->h = h; m->w = w;
    m->d = ctx->alloc_n<uint32_t>(std::max<size_t>(h * w, 1));
    return m;
  }
  void* temp(size_t bytes) { void* p = ctx->alloc(bytes); scratch.push_back(p); return p; }
  const uint32_t* events(const void* ev, size_t bytes) {
    uint32_t* owned = nullptr;
    const uint32_t* d = ctx->events_on_device(ev, bytes, &owned);
    if (owned) scratch.push_back(owned);
    return d;
  }
  uint32_t* flag(int kind) {     // a zeroed device word a kernel of this batch may set; read back by finish
    if (!d_flags) {
      d_flags = (uint32_t*)temp(MAX_FLAGS * 4);
      HIP_CHECK(hipMemsetAsync(d_flags, 0, MAX_FLAGS * 4, ctx->stream));
    }
    if ((int)flag_kind.size() == MAX_FLAGS) throw std::runtime_error("zkm_tracegen: too many generators in one batch");
    flag_kind.push_back(kind);
    return d_flags + flag_kind.size() - 1;
  }
  void finish(const char* mark) {
    const uint32_t* h_flags = flag_kind.empty() ? nullptr : ctx->download_async(d_flags, flag_kind.size());
    ctx->mark(mark);
    ctx->end_timing(false);     // synchronises: the callers' event buffers are free again, the flags are on the host
    for (size_t i = 0; i < flag_kind.size(); i++) {
      const uint32_t f = h_flags[i];
      if (!f) continue;
      if (flag_kind[i] == F_CPU) {
        if (f & 1) throw std::runtime_error("zkm_tracegen_cpu: an event's pc lies outside the program");
        throw std::runtime_error("zkm_tracegen_cpu: an event's clock does not fit 24 bits (a shard holds fewer than 2^24 / 5 cycles)");
      }
      if (flag_kind[i] == F_GLOBAL) {
        if (f & tracegen::GLOBAL_ERR_NOT_U16) throw std::runtime_error("zkm_tracegen_global: message[0] of an event is not a u16");
        if (f & tracegen::GLOBAL_ERR_NO_POINT) throw std::runtime_error("zkm_tracegen_global: a message has no curve point within 256 offsets");
        if (f & tracegen::GLOBAL_ERR_INFINITY) throw std::runtime_error("zkm_tracegen_global: the running sum reached the point at infinity");
        throw std::runtime_error("zkm_tracegen_global: a message's point has the running sum's x-coordinate");
      }
      throw std::runtime_error("zkm_tracegen_syscall: fixed log2 rows is too small");
    }
    for (void* p : scratch) ctx->release(p);
    scratch.clear();
    made.clear();
    host_keep.clear();
  }
  void abort() {     // a generator threw, or a kernel flagged an error: nothing of the batch survives
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : scratch) ctx->release(p);
    for (zkm_matrix* m : made) { if (m->d) ctx->release(m->d); delete m; }
    scratch.clear(); made.clear(); host_keep.clear();
  }
};

// the chips whose rows are a function of one event each (tracegen::alu_rows<CHIP>). n_dev: the event count in device memory
// (syscall_core_compact), n_events then being its upper bound
static zkm_matrix* enqueue_rows(TraceBatch& b, int chip, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                const uint32_t* n_dev = nullptr, const uint32_t* dev_events_given = nullptr, size_t height_given = 0) {
  static_assert(sizeof(zkm_alu_event) == 28 && sizeof(zkm_jump_event) == 28 && sizeof(zkm_mov_cond_event) == 28 &&
                sizeof(zkm_comp_alu_event) == 64, "event records mirror the #[repr(C)] executor structs");
  zkm_ctx* ctx = b.ctx;
  if (chip < 0 || chip >= tracegen::NUM_CHIPS) throw std::runtime_error("zkm_tracegen: unknown chip");
  if (n_events && !events && !dev_events_given) throw std::runtime_error("zkm_tracegen: null events");
  const size_t height = height_given ? height_given : padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_alu");
  const size_t w = (size_t)tracegen::chip_width(chip);
  zkm_matrix* m = b.matrix(height, w);
  // no events: every row is padding_row<CHIP> (tracegen::alu_rows_body), the same on every row — the shape step adds such chips to every
  // shard (eight of the benchmarked shard's eighteen); the commitment's LDE then skips the pass that would find that out by reading them
  m->uniform_rows = n_events == 0 && !n_dev && !dev_events_given;
  const size_t event_bytes = 4 * (size_t)tracegen::event_words(chip);
  const uint32_t* dev_events = dev_events_given ? dev_events_given : b.events(events, n_events * event_bytes);
  uint32_t* counts = blu ? blu->counts : nullptr;
  if (n_dev) {
    if (chip != tracegen::SYSCALL_CORE) throw std::runtime_error("zkm_tracegen: a device-side event count is for SyscallCore only");
    const int tiles = counts ? tracegen::TILES_PER_BLOCK : 1;
    KLAUNCH(ctx, "tracegen_alu", event_bytes * n_events + 4.0 * height * w, tracegen::alu_rows_counted<tracegen::SYSCALL_CORE>,
            dim3(div_up(height, (size_t)tiles * tracegen::THREADS)), dim3(tracegen::THREADS), counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0,
            dev_events, n_dev, height, m->d, counts, tiles);
    return m;
  }
  switch (chip) {
    case tracegen::ADD_SUB: launch_alu_rows<tracegen::ADD_SUB>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::BITWISE: launch_alu_rows<tracegen::BITWISE>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::LT: launch_alu_rows<tracegen::LT>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::SHIFT_LEFT: launch_alu_rows<tracegen::SHIFT_LEFT>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::SHIFT_RIGHT: launch_alu_rows<tracegen::SHIFT_RIGHT>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::CLO_CLZ: launch_alu_rows<tracegen::CLO_CLZ>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::JUMP: launch_alu_rows<tracegen::JUMP>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::MOV_COND: launch_alu_rows<tracegen::MOV_COND>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::BRANCH: launch_alu_rows<tracegen::BRANCH>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::MUL: launch_alu_rows<tracegen::MUL>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::DIVREM: launch_alu_rows<tracegen::DIVREM>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::MEMORY_INSTRS: launch_alu_rows<tracegen::MEMORY_INSTRS>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::SYSCALL_INSTRS: launch_alu_rows<tracegen::SYSCALL_INSTRS>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::MISC_INSTRS: launch_alu_rows<tracegen::MISC_INSTRS>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::SYSCALL_CORE: launch_alu_rows<tracegen::SYSCALL_CORE>(ctx, dev_events, n_events, height, m->d, counts); break;
    case tracegen::SYSCALL_PRECOMPILE: launch_alu_rows<tracegen::SYSCALL_PRECOMPILE>(ctx, dev_events, n_events, height, m->d, counts); break;
  }
  return m;
}

// a batch of one generator behind a public entry point
static int tracegen_single(zkm_ctx* ctx, zkm_matrix** out, const std::function<zkm_matrix*(TraceBatch&)>& enqueue) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  ctx->begin_timing();
  TraceBatch b(ctx);
  zkm_matrix* m = nullptr;
  try {
    m = enqueue(b);
    b.finish("trace generation");
  } catch (...) {
    b.abort();
    throw;
  }
  *out = m;
  API_END
}

static int tracegen_events(zkm_ctx* ctx, int chip, const void* events, size_t n_events, int fixed_log2_rows,
                           zkm_byte_lookups* blu, zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_rows(b, chip, events, n_events, fixed_log2_rows, blu); });
}

int zkm_tracegen_alu(zkm_ctx* ctx, int chip, const zkm_alu_event* events, size_t n_events, int fixed_log2_rows,
                     zkm_byte_lookups* blu, zkm_matrix** out) {
  if (chip < 0 || chip >= tracegen::NUM_ALU_CHIPS) { g_err = "zkm_tracegen_alu: unknown chip"; return -1; }
  return tracegen_events(ctx, chip, events, n_events, fixed_log2_rows, blu, out);
}
size_t zkm_tracegen_jump_width(void) { return (size_t)tracegen::chip_width(tracegen::JUMP); }
int zkm_tracegen_jump(zkm_ctx* ctx, const zkm_jump_event* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out) {
  return tracegen_events(ctx, tracegen::JUMP, events, n_events, fixed_log2_rows, nullptr, out);
}

// A recursion chip whose trace is its records end to end (zkm_tracegen_flat): the words — a host pointer, or an address
// zkm_events_upload_async returned — transposed into the padded column-major matrix, zero rows behind them.
static zkm_matrix* enqueue_flat(TraceBatch& b, const uint32_t* words, size_t n_words, size_t width, int fixed_log2_rows) {
  zkm_ctx* ctx = b.ctx;
  if (width == 0) throw std::runtime_error("zkm_tracegen_flat: zero width");
  if (n_words && !words) throw std::runtime_error("zkm_tracegen_flat: null records");
  const size_t rows = (n_words + width - 1) / width;
  size_t height = 16;
  if (fixed_log2_rows >= 0) {
    if (fixed_log2_rows > 30) throw std::runtime_error("zkm_tracegen_flat: fixed log2 rows out of range");
    height = (size_t)1 << fixed_log2_rows;
    if (rows > height) throw std::runtime_error("zkm_tracegen_flat: fixed log2 rows is too small");
  } else {
    while (height < rows) height <<= 1;
  }
  zkm_matrix* m = b.matrix(height, width);
  m->uniform_rows = n_words == 0;
  const uint32_t* dev = n_words ? b.events(words, n_words * 4) : nullptr;
  ctx->flush_staged();
  hipLaunchKernelGGL(open::flat_rows, dim3(div_up(width, 32), div_up(height, 32)), dim3(32, 8), 0, ctx->stream, dev, n_words, m->d, width, height);
  LAUNCH_CHECK();
  return m;
}
int zkm_tracegen_flat(zkm_ctx* ctx, const uint32_t* words, size_t n_words, size_t width, int fixed_log2_rows, zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_flat(b, words, n_words, width, fixed_log2_rows); });
}

size_t zkm_tracegen_branch_width(void) { return (size_t)tracegen::chip_width(tracegen::BRANCH); }
int zkm_tracegen_branch(zkm_ctx* ctx, const zkm_branch_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                        zkm_matrix** out) {
  return tracegen_events(ctx, tracegen::BRANCH, events, n_events, fixed_log2_rows, blu, out);
}
size_t zkm_tracegen_mul_width(void) { return (size_t)tracegen::chip_width(tracegen::MUL); }
int zkm_tracegen_mul(zkm_ctx* ctx, const zkm_comp_alu_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                     zkm_matrix** out) {
  return tracegen_events(ctx, tracegen::MUL, events, n_events, fixed_log2_rows, blu, out);
}
size_t zkm_tracegen_memory_instrs_width(void) { return (size_t)tracegen::chip_width(tracegen::MEMORY_INSTRS); }
int zkm_tracegen_memory_instrs(zkm_ctx* ctx, const zkm_mem_instr_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                               zkm_matrix** out) {
  static_assert(sizeof(zkm_mem_instr_event) == 64, "event records mirror the #[repr(C)] executor structs");
  return tracegen_events(ctx, tracegen::MEMORY_INSTRS, events, n_events, fixed_log2_rows, blu, out);
}
size_t zkm_tracegen_syscall_instrs_width(void) { return (size_t)tracegen::chip_width(tracegen::SYSCALL_INSTRS); }
int zkm_tracegen_syscall_instrs(zkm_ctx* ctx, const zkm_syscall_event* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out) {
  static_assert(sizeof(zkm_syscall_event) == 56, "event records mirror the #[repr(C)] executor structs");
  return tracegen_events(ctx, tracegen::SYSCALL_INSTRS, events, n_events, fixed_log2_rows, nullptr, out);
}
size_t zkm_tracegen_misc_instrs_width(void) { return (size_t)tracegen::chip_width(tracegen::MISC_INSTRS); }
int zkm_tracegen_misc_instrs(zkm_ctx* ctx, const zkm_misc_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                             zkm_matrix** out) {
  static_assert(sizeof(zkm_misc_event) == 60, "event records mirror the #[repr(C)] executor structs");
  return tracegen_events(ctx, tracegen::MISC_INSTRS, events, n_events, fixed_log2_rows, blu, out);
}
size_t zkm_tracegen_divrem_width(void) { return (size_t)tracegen::chip_width(tracegen::DIVREM); }
int zkm_tracegen_divrem(zkm_ctx* ctx, const zkm_comp_alu_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                        zkm_matrix** out) {
  return tracegen_events(ctx, tracegen::DIVREM, events, n_events, fixed_log2_rows, blu, out);
}
size_t zkm_tracegen_mov_cond_width(void) { return (size_t)tracegen::chip_width(tracegen::MOV_COND); }
int zkm_tracegen_mov_cond(zkm_ctx* ctx, const zkm_mov_cond_event* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out) {
  return tracegen_events(ctx, tracegen::MOV_COND, events, n_events, fixed_log2_rows, nullptr, out);
}

size_t zkm_tracegen_cpu_width(void) { return (size_t)tracegen::CPU_WIDTH; }
static zkm_matrix* enqueue_cpu(TraceBatch& b, const zkm_cpu_event* events, size_t n_events, const zkm_instruction* program, size_t n_instr,
                               uint32_t pc_base, uint32_t shard, int fixed_log2_rows, int program_fixed_log2_rows, zkm_byte_lookups* blu,
                               zkm_matrix** program_mults_out) {
  static_assert(sizeof(zkm_cpu_event) == 4 * tracegen::CPU_EVENT_WORDS && sizeof(zkm_instruction) == 4 * tracegen::INSTRUCTION_WORDS,
                "event records mirror the #[repr(C)] executor structs");
  zkm_ctx* ctx = b.ctx;
  if (n_events && (!events || !program || !n_instr)) throw std::runtime_error("zkm_tracegen_cpu: null events or program");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_cpu");
  const size_t pheight = program_mults_out ? padded_trace_rows(n_instr, program_fixed_log2_rows, "zkm_tracegen_cpu (program)") : 0;
  zkm_matrix* m = b.matrix(height, tracegen::CPU_WIDTH);
  zkm_matrix* pm = program_mults_out ? b.matrix(pheight, 1) : nullptr;
  if (pm) HIP_CHECK(hipMemsetAsync(pm->d, 0, pheight * 4, ctx->stream));
  const uint32_t* dev_events = b.events(events, n_events * sizeof(zkm_cpu_event));
  // the program goes through the staged ring when it fits (one copy with the batch's other small tables), else its own buffer
  const uint32_t* d_program = (const uint32_t*)ctx->upload_staged(program, n_instr * sizeof(zkm_instruction), &b.scratch);
  uint32_t* d_bad = b.flag(TraceBatch::F_CPU);
  uint32_t* counts = blu ? blu->counts : nullptr;
  const int tiles = counts ? tracegen::TILES_PER_BLOCK : 1;
  KLAUNCH(ctx, "tracegen_cpu", 280.0 * n_events + 4.0 * height * tracegen::CPU_WIDTH, tracegen::cpu_rows,
          dim3(div_up(height, (size_t)tiles * tracegen::THREADS)), dim3(tracegen::THREADS),
          counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0, dev_events, n_events, d_program, n_instr,
          pc_base, shard, height, m->d, counts, tiles, (int*)d_bad, pm ? pm->d : (uint32_t*)nullptr);
  if (pm) {
    hipLaunchKernelGGL(tracegen::counts_to_field, dim3(div_up(pheight, 256)), dim3(256), 0, ctx->stream, pm->d, pheight);
    LAUNCH_CHECK();
    *program_mults_out = pm;
  }
  return m;
}

int zkm_tracegen_cpu_and_program(zkm_ctx* ctx, const zkm_cpu_event* events, size_t n_events, const zkm_instruction* program, size_t n_instr,
                                 uint32_t pc_base, uint32_t shard, int fixed_log2_rows, int program_fixed_log2_rows, zkm_byte_lookups* blu,
                                 zkm_matrix** out, zkm_matrix** program_mults_out) {
  zkm_matrix* pm = nullptr;
  const int rc = tracegen_single(ctx, out, [&](TraceBatch& b) {
    return enqueue_cpu(b, events, n_events, program, n_instr, pc_base, shard, fixed_log2_rows, program_fixed_log2_rows, blu, program_mults_out ? &pm : nullptr);
  });
  if (rc == 0 && program_mults_out) *program_mults_out = pm;
  return rc;
}

int zkm_tracegen_cpu(zkm_ctx* ctx, const zkm_cpu_event* events, size_t n_events, const zkm_instruction* program, size_t n_instr,
                     uint32_t pc_base, uint32_t shard, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  return zkm_tracegen_cpu_and_program(ctx, events, n_events, program, n_instr, pc_base, shard, fixed_log2_rows, -1, blu, out, nullptr);
}

int zkm_tracegen_program(zkm_ctx* ctx, const zkm_instruction* program, size_t n_instr, uint32_t pc_base, int fixed_log2_rows,
                         zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_instr && !program) throw std::runtime_error("zkm_tracegen_program: null program");
  const size_t height = padded_trace_rows(n_instr, fixed_log2_rows, "zkm_tracegen_program");
  ctx->begin_call();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::PROGRAM_PREP_WIDTH;
  uint32_t* d_program = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    d_program = (uint32_t*)ctx->alloc(std::max<size_t>(n_instr * sizeof(zkm_instruction), 4));
    if (n_instr) HIP_CHECK(hipMemcpyAsync(d_program, program, n_instr * sizeof(zkm_instruction), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(tracegen::program_rows, dim3(div_up(height, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_program, n_instr,
                       pc_base, height, m->d);
    LAUNCH_CHECK();
    ctx->sync(ctx->stream);
  } catch (...) {
    if (d_program) ctx->release(d_program);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_program);
  *out = m;
  API_END
}

int zkm_tracegen_program_mults(zkm_ctx* ctx, const zkm_cpu_event* events, size_t n_events, size_t n_instr, uint32_t pc_base,
                               int fixed_log2_rows, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_program_mults: null events");
  const size_t height = padded_trace_rows(n_instr, fixed_log2_rows, "zkm_tracegen_program_mults");
  ctx->begin_call();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = 1;
  uint32_t* d_events = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height);
    HIP_CHECK(hipMemsetAsync(m->d, 0, height * 4, ctx->stream));
    const uint32_t* dev_events = ctx->events_on_device(events, n_events * sizeof(zkm_cpu_event), &d_events);
    if (n_events) {
      hipLaunchKernelGGL(tracegen::program_count, dim3(div_up(n_events, 256)), dim3(256), 0, ctx->stream, dev_events, n_events,
                         n_instr, pc_base, m->d);
      LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(tracegen::counts_to_field, dim3(div_up(height, 256)), dim3(256), 0, ctx->stream, m->d, height);
    LAUNCH_CHECK();
    ctx->sync(ctx->stream);
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  *out = m;
  API_END
}

static zkm_matrix* enqueue_memory_local(TraceBatch& b, const zkm_memory_local_event* events, size_t n_events, int fixed_log2_rows) {
  static_assert(sizeof(zkm_memory_local_event) == 28, "event records mirror the #[repr(C)] executor structs");
  zkm_ctx* ctx = b.ctx;
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_memory_local: null events");
  const size_t height = padded_trace_rows(div_up(n_events, (size_t)tracegen::MEMORY_LOCAL_ENTRIES), fixed_log2_rows, "zkm_tracegen_memory_local");
  zkm_matrix* m = b.matrix(height, tracegen::MEMORY_LOCAL_WIDTH);
  const uint32_t* dev_events = b.events(events, n_events * sizeof(zkm_memory_local_event));
  ctx->flush_staged();
  hipLaunchKernelGGL(tracegen::memory_local_rows, dim3(div_up(height * tracegen::MEMORY_LOCAL_ENTRIES, (size_t)256)), dim3(256), 0, ctx->stream,
                     dev_events, n_events, height, m->d);
  LAUNCH_CHECK();
  return m;
}

int zkm_tracegen_memory_local(zkm_ctx* ctx, const zkm_memory_local_event* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_memory_local(b, events, n_events, fixed_log2_rows); });
}

static zkm_matrix* enqueue_global(TraceBatch& b, const zkm_global_lookup_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu) {
  static_assert(sizeof(zkm_global_lookup_event) == 32, "event records mirror the #[repr(C)] executor structs");
  zkm_ctx* ctx = b.ctx;
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_global: null events");
  if (!blu) throw std::runtime_error("zkm_tracegen_global: null byte lookups");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_global");
  zkm_matrix* m = b.matrix(height, tracegen::GLOBAL_WIDTH);
  // the events are read on the device only (the u16 check of message[0] included: GLOBAL_ERR_NOT_U16), so a prefetched address does as well
  const uint32_t* d_events = b.events(events, n_events * sizeof(zkm_global_lookup_event));
  uint32_t* d_err = b.flag(TraceBatch::F_GLOBAL);
  std::vector<uint32_t*> levels;     // scan buffers: the points behind the start digest, then the chunk sums of each level
  std::vector<size_t> sizes;
  for (size_t n = n_events + 1;; n = div_up(n, (size_t)tracegen::SCAN_CHUNK)) {
    levels.push_back((uint32_t*)b.temp(n * tracegen::POINT_WORDS * sizeof(uint32_t)));
    sizes.push_back(n);
    if (n <= (size_t)tracegen::SCAN_BLOCK) break;
  }
  const double bytes = 32.0 * n_events + 4.0 * height * tracegen::GLOBAL_WIDTH;
  KLAUNCH(ctx, "tracegen_global_points", bytes, tracegen::global_point_rows, dim3(div_up(height, (size_t)256)), dim3(256), 0,
          d_events, n_events, height, m->d, levels[0], blu->counts, d_err);
  for (size_t l = 0; l + 1 < levels.size(); l++)
    KLAUNCH(ctx, "tracegen_global_scan", 64.0 * sizes[l], tracegen::global_scan_reduce, dim3(div_up(sizes[l + 1], (size_t)64)), dim3(64), 0,
            (const uint32_t*)levels[l], sizes[l], levels[l + 1], sizes[l + 1]);
  KLAUNCH(ctx, "tracegen_global_scan", 128.0 * sizes.back(), tracegen::global_scan_block, dim3(1), dim3(tracegen::SCAN_BLOCK), 0, levels.back(),
          sizes.back());
  for (size_t l = levels.size() - 1; l-- > 0;)
    KLAUNCH(ctx, "tracegen_global_scan", 128.0 * sizes[l], tracegen::global_scan_apply, dim3(div_up(sizes[l + 1], (size_t)64)), dim3(64), 0, levels[l],
            sizes[l], (const uint32_t*)levels[l + 1], sizes[l + 1]);
  KLAUNCH(ctx, "tracegen_global_accum", bytes, tracegen::global_accum_rows, dim3(div_up(height, (size_t)256)), dim3(256), 0,
          (const uint32_t*)levels[0], n_events, height, m->d, d_err);
  return m;
}

int zkm_tracegen_global(zkm_ctx* ctx, const zkm_global_lookup_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                        zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_global(b, events, n_events, fixed_log2_rows, blu); });
}

static zkm_matrix* enqueue_poseidon2_wide(TraceBatch& b, const uint32_t* events, size_t n_events, int fixed_log2_rows) {
  zkm_ctx* ctx = b.ctx;
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_poseidon2_wide: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_poseidon2_wide");
  zkm_matrix* m = b.matrix(height, tracegen::POSEIDON2_WIDE_WIDTH);
  const uint32_t* d_events = n_events ? b.events(events, n_events * 128) : (const uint32_t*)b.temp(4);
  KLAUNCH(ctx, "tracegen_poseidon2_wide", 128.0 * n_events + 4.0 * height * tracegen::POSEIDON2_WIDE_WIDTH, tracegen::poseidon2_wide_rows,
          dim3(div_up(height, (size_t)tracegen::THREADS)), dim3(tracegen::THREADS), 0, d_events, n_events, height, m->d);
  return m;
}
int zkm_tracegen_poseidon2_wide(zkm_ctx* ctx, const uint32_t* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_poseidon2_wide(b, events, n_events, fixed_log2_rows); });
}

// SyscallCore keeps the events whose code has the send-to-table byte set or names a Linux syscall (syscall/chip.rs:252-259): host events are
// filtered here, events that are already in HBM (zkm_events_upload_async) by tracegen::syscall_core_compact, the row kernel then reading
// the count from device memory — no host read either way for a prefetched shard.
static zkm_matrix* enqueue_syscall(TraceBatch& b, const zkm_syscall_event* events, size_t n_events, int precompile, int fixed_log2_rows, zkm_byte_lookups* blu) {
  static_assert(sizeof(zkm_syscall_event) == 56, "event records mirror the #[repr(C)] executor structs");
  zkm_ctx* ctx = b.ctx;
  if (precompile) return enqueue_rows(b, tracegen::SYSCALL_PRECOMPILE, events, n_events, fixed_log2_rows, blu);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen: null events");
  if (n_events && ctx->prefetched.count(events)) {
    const uint32_t* d_all = b.events(events, n_events * sizeof(zkm_syscall_event));
    const size_t cap = fixed_log2_rows >= 0 ? std::min(n_events, padded_trace_rows(0, fixed_log2_rows, "zkm_tracegen_syscall")) : n_events;
    uint32_t* d_kept = (uint32_t*)b.temp(cap * sizeof(zkm_syscall_event));
    uint32_t* d_n = (uint32_t*)b.temp(4);
    uint32_t* d_flag = b.flag(TraceBatch::F_SYSCALL_CORE);
    KLAUNCH(ctx, "tracegen_syscall_compact", 2.0 * n_events * sizeof(zkm_syscall_event), tracegen::syscall_core_compact, dim3(1), dim3(1024), 0,
            d_all, n_events, d_kept, d_n, cap, d_flag);
    if (fixed_log2_rows < 0) {
      // no shape: the trace's height is the next power of two of the kept count, which the host has to learn first (one round trip; a
      // shaped shard — the reference's default — fixes the height and takes the branch below)
      const uint32_t* h_n = ctx->download_async(d_n, 1);
      ctx->sync(ctx->stream);
      return enqueue_rows(b, tracegen::SYSCALL_CORE, nullptr, *h_n, -1, blu, nullptr, d_kept);
    }
    // the kept count stays on the device: the row kernel reads it there
    return enqueue_rows(b, tracegen::SYSCALL_CORE, nullptr, cap, fixed_log2_rows, blu, d_n, d_kept, (size_t)1 << fixed_log2_rows);
  }
  b.host_keep.emplace_back();
  std::vector<zkm_syscall_event>& kept = b.host_keep.back();
  for (size_t i = 0; i < n_events; i++) {
    const uint32_t code = events[i].a_record.prev_value;
    if (((code >> 16) & 0xff) == 1 || ((code >> 8) & 0xff) != 0) kept.push_back(events[i]);
  }
  return enqueue_rows(b, tracegen::SYSCALL_CORE, kept.empty() ? nullptr : (const void*)kept.data(), kept.size(), fixed_log2_rows, blu);
}

int zkm_tracegen_syscall(zkm_ctx* ctx, const zkm_syscall_event* events, size_t n_events, int precompile, int fixed_log2_rows, zkm_byte_lookups* blu,
                         zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_syscall(b, events, n_events, precompile, fixed_log2_rows, blu); });
}

int zkm_tracegen_memory_global(zkm_ctx* ctx, const zkm_memory_init_finalize_event* events, size_t n_events, uint32_t previous_addr, int fixed_log2_rows,
                               zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_memory_init_finalize_event) == 16, "event records mirror the #[repr(C)] executor structs");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_memory_global: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_memory_global");
  // generate_trace sorts the events by address first (memory/global.rs:131)
  std::vector<zkm_memory_init_finalize_event> sorted(events, events + n_events);
  std::stable_sort(sorted.begin(), sorted.end(), [](const zkm_memory_init_finalize_event& a, const zkm_memory_init_finalize_event& b) { return a.addr < b.addr; });
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::MEMORY_GLOBAL_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(n_events * 16, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, sorted.data(), n_events * 16, hipMemcpyHostToDevice, ctx->stream));
    KLAUNCH(ctx, "tracegen_memory_global", 16.0 * n_events + 4.0 * height * m->w, tracegen::memory_global_rows, dim3(div_up(height, (size_t)tracegen::THREADS)),
            dim3(tracegen::THREADS), 0, (const uint32_t*)d_events, n_events, previous_addr, height, m->d, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error("zkm_tracegen_memory_global: addresses are not strictly increasing (from the previous shard's last address on)");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_poseidon2_permute(zkm_ctx* ctx, const zkm_poseidon2_permute_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                   zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_poseidon2_permute_event) == 4 * tracegen::POSEIDON2_PERMUTE_EVENT_WORDS, "flattened Poseidon2PermuteEvent is 99 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_poseidon2_permute: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_poseidon2_permute");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::POSEIDON2_PERMUTE_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_poseidon2_permute_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_poseidon2_permute", (double)ev_bytes + 4.0 * height * m->w, tracegen::poseidon2_permute_rows,
            dim3(div_up(height, (size_t)tracegen::THREADS)), dim3(tracegen::THREADS), counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0,
            (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error("zkm_tracegen_poseidon2_permute: a state word is not a field element, or the post-state is not the permutation of the pre-state");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_keccak_sponge(zkm_ctx* ctx, const zkm_keccak_sponge_block* blocks, size_t n_blocks, int fixed_log2_rows, zkm_byte_lookups* blu,
                               zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_keccak_sponge_block) == 4 * tracegen::KECCAK_SPONGE_BLOCK_WORDS, "a KeccakSpongeEvent block is 337 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_blocks && !blocks) throw std::runtime_error("zkm_tracegen_keccak_sponge: null blocks");
  const size_t height = padded_trace_rows(24 * n_blocks, fixed_log2_rows, "zkm_tracegen_keccak_sponge");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::KECCAK_SPONGE_WIDTH;
  uint32_t* d_blocks = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_blocks * sizeof(zkm_keccak_sponge_block);
    d_blocks = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_blocks) HIP_CHECK(hipMemcpyAsync(d_blocks, blocks, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_keccak_sponge", (double)ev_bytes + 4.0 * height * m->w, tracegen::keccak_sponge_rows,
            dim3(div_up(height, (size_t)tracegen::THREADS)), dim3(tracegen::THREADS), counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0,
            (const uint32_t*)d_blocks, n_blocks, height, m->d, counts, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    static const char* const why[] = {"", "input length is not a positive multiple of 36 words, or the block index is past it",
                                      "the first block of a call is not absorbed into the zero state", "the input length record does not hold the input length",
                                      "the output records are not the squeezed state", "the blocks of one call do not follow each other and chain"};
    if (bad) throw std::runtime_error(std::string("zkm_tracegen_keccak_sponge: ") + why[16 - bad >= 1 && 16 - bad < 6 ? 16 - bad : 0]);
  } catch (...) {
    if (d_blocks) ctx->release(d_blocks);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_blocks);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_sha_extend(zkm_ctx* ctx, const zkm_sha_extend_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_sha_extend_event) == 4 * tracegen::SHA_EXTEND_EVENT_WORDS, "flattened EVENTsha_extend");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_sha_extend: null events");
  const size_t height = padded_trace_rows(48 * n_events, fixed_log2_rows, "zkm_tracegen_sha_extend");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::SHA_EXTEND_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_sha_extend_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_sha_extend", (double)ev_bytes + 4.0 * height * m->w, tracegen::sha_extend_rows, dim3(div_up(height, (size_t)tracegen::THREADS)),
            dim3(tracegen::THREADS), counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0, (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error("zkm_tracegen_sha_extend: a write record does not hold the schedule word its reads give");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_sha_compress(zkm_ctx* ctx, const zkm_sha_compress_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_sha_compress_event) == 4 * tracegen::SHA_COMPRESS_EVENT_WORDS, "flattened EVENTsha_compress");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_sha_compress: null events");
  const size_t height = padded_trace_rows(80 * n_events, fixed_log2_rows, "zkm_tracegen_sha_compress");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::SHA_COMPRESS_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_sha_compress_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_sha_compress", (double)ev_bytes + 4.0 * height * m->w, tracegen::sha_compress_rows, dim3(div_up(height, (size_t)tracegen::THREADS)),
            dim3(tracegen::THREADS), counts ? 2 * tracegen::HASH_SLOTS * sizeof(uint32_t) : 0, (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error("zkm_tracegen_sha_compress: the write records are not the state read plus the compressed state");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

// The U8Range lookups of the byte-limb columns of a big-field table's real rows (tracegen::u8_pair_histogram), added to the lookup counters
static void count_u8_pairs(zkm_ctx* ctx, const zkm_matrix* m, size_t n_real, const tracegen::U8Segments& seg, uint32_t* counts) {
  if (!counts || !n_real) return;
  int log_slab_rows = n_real > 16384 ? 10 : 8;      // 1024-row slabs (a slab's byte columns: a few MB) until there would be more than U8H_MAX_SLABS of them
  while (div_up(n_real, (size_t)1 << log_slab_rows) > (size_t)tracegen::U8H_MAX_SLABS) log_slab_rows++;
  const size_t slabs = div_up(n_real, (size_t)1 << log_slab_rows);
  size_t columns = 0;
  for (int k = 0; k < seg.n; k++) {
    const int reps = seg.reps[k] > 1 ? seg.reps[k] : 1;
    if (seg.cols[k] < 0 || seg.start[k] < 0 || seg.stride[k] < 0 || (size_t)(seg.start[k] + (reps - 1) * seg.stride[k] + seg.cols[k]) > m->w)
      throw std::runtime_error("count_u8_pairs: bad column segment");
    columns += (size_t)seg.cols[k] * reps;
  }
  uint32_t* partial = ctx->alloc_n<uint32_t>(slabs * 65536);
  try {
    KLAUNCH(ctx, "tracegen_u8_pairs", 4.0 * tracegen::U8H_RANGES * n_real * columns, tracegen::u8_pair_histogram, dim3(slabs * tracegen::U8H_RANGES),
            dim3(tracegen::U8H_THREADS), (size_t)tracegen::U8H_KEYS * 4, (const uint32_t*)m->d, m->h, n_real, seg, log_slab_rows, partial);
    KLAUNCH(ctx, "tracegen_u8_pairs_reduce", 4.0 * slabs * 65536, tracegen::u8_pair_reduce, dim3(65536 / 256), dim3(256), 0, (const uint32_t*)partial, (int)slabs,
            counts + (size_t)tracegen::B_U8RANGE * tracegen::BYTE_ROWS);
  } catch (...) {
    ctx->release(partial);
    throw;
  }
  ctx->release(partial);
}

int zkm_tracegen_ed_add(zkm_ctx* ctx, const zkm_ed_add_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_ed_add_event) == 4 * tracegen::ED_ADD_EVENT_WORDS, "flattened EllipticCurveAddEvent is 180 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_ed_add: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_ed_add");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::ED_ADD_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_ed_add_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_ed_add", (double)ev_bytes + 4.0 * height * m->w, tracegen::ed_add_rows, dim3(div_up(height, (size_t)tracegen::bf_threads(8))),
            dim3(tracegen::bf_threads(8)), tracegen::bf_lds_bytes(8, counts != nullptr), (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{1, {5 + 16 * 13 + 16 * 9}, {8 * tracegen::ED_GADGET}}, counts);      // the eight gadgets
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error("zkm_tracegen_ed_add: the words an event writes to p are not p + q on Ed25519");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_ed_decompress(zkm_ctx* ctx, const zkm_ed_decompress_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_ed_decompress_event) == 4 * tracegen::ED_DECOMPRESS_EVENT_WORDS, "flattened EdDecompressEvent is 92 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_ed_decompress: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_ed_decompress");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::ED_DECOMPRESS_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_ed_decompress_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_ed_decompress", (double)ev_bytes + 4.0 * height * m->w, tracegen::ed_decompress_rows,
            dim3(div_up(height, (size_t)tracegen::bf_threads(8))), dim3(tracegen::bf_threads(8)), tracegen::bf_lds_bytes(8, counts != nullptr), (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    // yy .. x's multiplication (six gadgets; the last one's result columns hold the root, FieldSqrtCols range-checks both the root and the
    // product — the product's bytes are u_div_v's result columns), neg_x
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{3, {215, 1378, 967}, {6 * tracegen::ED_GADGET, tracegen::ED_GADGET, tracegen::ED_LIMBS}}, counts);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    static const char* const why[] = {"", "y is not below the field modulus, or the sign is not a bit", "(y^2 - 1) / (d y^2 + 1) is not a square: no such point",
                                      "the words an event writes are not the x its y and sign give"};
    if (bad) throw std::runtime_error(std::string("zkm_tracegen_ed_decompress: ") + why[16 - bad >= 1 && 16 - bad <= 3 ? 16 - bad : 0]);
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_uint256_mul(zkm_ctx* ctx, const zkm_uint256_mul_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_uint256_mul_event) == 4 * tracegen::UINT256_MUL_EVENT_WORDS, "flattened Uint256MulEvent is 132 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_uint256_mul: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_uint256_mul");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::UINT256_MUL_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_uint256_mul_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_uint256_mul", (double)ev_bytes + 4.0 * height * m->w, tracegen::uint256_mul_rows, dim3(div_up(height, (size_t)tracegen::U256_THREADS)),
            dim3(tracegen::U256_THREADS), tracegen::u256_lds_bytes(counts != nullptr), (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    // output: result and carry (32 + 32), witness_low, witness_high (63 each: the last limb is checked alone)
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{3, {255, 255 + 64, 255 + 64 + 63}, {64, 63, 63}}, counts);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad == 15) throw std::runtime_error("zkm_tracegen_uint256_mul: x * y / modulus does not fit 256 bits (the carry columns hold 32 bytes)");
    if (bad) throw std::runtime_error("zkm_tracegen_uint256_mul: the words an event writes to x are not x * y mod modulus");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_u256x2048_mul(zkm_ctx* ctx, const zkm_u256x2048_mul_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_u256x2048_mul_event) == 4 * tracegen::U256X2048_MUL_EVENT_WORDS, "flattened U256xU2048MulEvent is 808 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_u256x2048_mul: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_u256x2048_mul");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::U256X2048_MUL_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_u256x2048_mul_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_u256x2048_mul", (double)ev_bytes + 4.0 * height * m->w, tracegen::u256x2048_mul_rows, dim3(div_up(height, (size_t)tracegen::U256_THREADS)),
            dim3(tracegen::U256_THREADS), tracegen::u256_lds_bytes(counts != nullptr), (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    // per gadget: result and carry (32 + 32), witness_low, witness_high (63 each); eight gadgets 190 columns apart
    const int G = tracegen::U256_GADGET, g0 = 1608;
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{3, {g0, g0 + 64, g0 + 127}, {64, 63, 63}, {8, 8, 8}, {G, G, G}}, counts);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad == 15) throw std::runtime_error("zkm_tracegen_u256x2048_mul: lo_ptr / hi_ptr are not what the register records hold");
    if (bad) throw std::runtime_error("zkm_tracegen_u256x2048_mul: the words an event writes to lo and hi are not a * b");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_boolean_circuit_garble(zkm_ctx* ctx, const zkm_garble_row* rows, size_t n_rows, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_garble_row) == 4 * tracegen::GARBLE_ROW_WORDS, "a BooleanCircuitGarble row record is 103 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_rows && !rows) throw std::runtime_error("zkm_tracegen_boolean_circuit_garble: null rows");
  const size_t height = padded_trace_rows(n_rows, fixed_log2_rows, "zkm_tracegen_boolean_circuit_garble");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::GARBLE_WIDTH;
  uint32_t* d_rows = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t bytes_in = n_rows * sizeof(zkm_garble_row);
    d_rows = (uint32_t*)ctx->alloc(std::max<size_t>(bytes_in, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_rows) HIP_CHECK(hipMemcpyAsync(d_rows, rows, bytes_in, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_boolean_circuit_garble", (double)bytes_in + 4.0 * height * m->w, tracegen::garble_rows, dim3(div_up(height, (size_t)256)), dim3(256),
            counts ? 2 * tracegen::BF_HASH_SLOTS * sizeof(uint32_t) : 0, (const uint32_t*)d_rows, n_rows, height, m->d, counts, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    static const char* reasons[] = {"", "a header row does not read the gate count and delta", "a gate row does not continue the row before it",
                                    "a gate type is neither 0 (AND) nor 7 (OR)", "the last gate's row does not write the result", "a call is cut short"};
    if (bad) throw std::runtime_error(std::string("zkm_tracegen_boolean_circuit_garble: ") + reasons[std::min(std::max(16 - bad, 1), 5)]);
  } catch (...) {
    if (d_rows) ctx->release(d_rows);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_rows);
  ctx->release(d_bad);
  *out = m;
  API_END
}

int zkm_tracegen_sys_linux(zkm_ctx* ctx, const zkm_linux_event* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  static_assert(sizeof(zkm_linux_event) == 4 * tracegen::LINUX_EVENT_WORDS, "flattened LinuxEvent is 23 words");
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_sys_linux: null events");
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, "zkm_tracegen_sys_linux");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::SYS_LINUX_WIDTH;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * sizeof(zkm_linux_event);
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    KLAUNCH(ctx, "tracegen_sys_linux", (double)ev_bytes + 4.0 * height * m->w, tracegen::sys_linux_rows, dim3(div_up(height, (size_t)256)), dim3(256),
            counts ? 2 * tracegen::BF_HASH_SLOTS * sizeof(uint32_t) : 0, (const uint32_t*)d_events, n_events, height, m->d, counts, d_bad);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error("zkm_tracegen_sys_linux: an event does not return what its syscall returns (v0, the value written to $a3, the new heap)");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

// Base fields of the short-Weierstrass curves (crates/curves/src/weierstrass/{secp256k1,secp256r1,bn254,bls12_381}.rs): modulus, its Barrett
// constant, the curve's `a`, as 32-bit limbs (generated from the reference's MODULUS bytes; tests compare the widths and costs they give)
struct Curve8 { bigfield::Modulus<8> m; uint32_t a[8]; };
struct Curve12 { bigfield::Modulus<12> m; uint32_t a[12]; };
static const Curve8 k_curves8[3] = {
  // Secp256k1: p, floor(2^512 / p), a
  {{{0xfffffc2fu, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0x000003d1u, 0x00000001u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u}}, {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}},
  // Secp256r1: p, floor(2^512 / p), a
  {{{0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu}, {0x00000003u, 0x00000000u, 0xffffffffu, 0xfffffffeu, 0xfffffffeu, 0xfffffffeu, 0xffffffffu, 0x00000000u, 0x00000001u}}, {0xfffffffcu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu}},
  // Bn254: p, floor(2^512 / p), a
  {{{0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}, {0x9bf90e51u, 0xf3aed8a1u, 0x7cd4c086u, 0xe965e176u, 0x8073013au, 0xb074a586u, 0x23a04a7au, 0x4a474626u, 0x00000005u}}, {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}},
};
// Bls12381: p, floor(2^768 / p), a
static const Curve12 k_curve_bls12381 = {{{0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau}, {0x6591ba2eu, 0x13e207f5u, 0x58f1c07bu, 0x997167a0u, 0x286779d3u, 0xdf4771e0u, 0xf6a0a94bu, 0x1b82741fu, 0xc7a6ba29u, 0x28101b0cu, 0xcc9e45ceu, 0xd835d2f3u, 0x00000009u}}, {0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}};
extern "C++" {
template <int NL, bool DOUBLE>
static void launch_weierstrass(zkm_ctx* ctx, const tracegen::CurveField<NL>& field, const uint32_t* d_events, size_t n_events, size_t height, uint32_t* out,
                               uint32_t* counts, int* d_bad, double bytes) {
  KLAUNCH(ctx, DOUBLE ? "tracegen_weierstrass_double" : "tracegen_weierstrass_add", bytes, (tracegen::weierstrass_rows<NL, DOUBLE>),
          dim3(div_up(height, (size_t)tracegen::bf_threads(NL))), dim3(tracegen::bf_threads(NL)), tracegen::bf_lds_bytes(NL, counts != nullptr), d_events, n_events,
          height, out, counts,
          d_bad, field);
}
}  // extern "C++"
static int tracegen_weierstrass(zkm_ctx* ctx, int curve, bool dbl, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                zkm_matrix** out, const char* who) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (curve < 0 || curve > 3) throw std::runtime_error(std::string(who) + ": unknown curve");
  if (n_events && !events) throw std::runtime_error(std::string(who) + ": null events");
  const int nl = curve == 3 ? 12 : 8, W = 2 * nl, G = 6 * 4 * nl - 4;
  const size_t ev_words = dbl ? 3 + 6 * W : 4 + 11 * W, width = (dbl ? 4 + 13 * W + 11 * G : 5 + 22 * W + 9 * G);
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, who);
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = width;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * ev_words * 4;
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    const double bytes = (double)ev_bytes + 4.0 * height * m->w;
    if (nl == 8) {
      tracegen::CurveField<8> f;
      f.m = k_curves8[curve].m;
      memcpy(f.a, k_curves8[curve].a, sizeof f.a);
      f.witness_offset = 1 << 14;
      if (dbl) launch_weierstrass<8, true>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
      else launch_weierstrass<8, false>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
    } else {
      tracegen::CurveField<12> f;
      f.m = k_curve_bls12381.m;
      memcpy(f.a, k_curve_bls12381.a, sizeof f.a);
      f.witness_offset = 1 << 15;
      if (dbl) launch_weierstrass<12, true>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
      else launch_weierstrass<12, false>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
    }
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{1, {(dbl ? 4 : 5) + 13 * W + (dbl ? 0 : 9 * W)}, {(dbl ? 11 : 9) * G}}, counts);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error(std::string(who) + ": a coordinate is not below the field modulus, or the words written to p are not the result point");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}
int zkm_tracegen_weierstrass_add(zkm_ctx* ctx, int curve, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  return tracegen_weierstrass(ctx, curve, false, events, n_events, fixed_log2_rows, blu, out, "zkm_tracegen_weierstrass_add");
}
int zkm_tracegen_weierstrass_double(zkm_ctx* ctx, int curve, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                    zkm_matrix** out) {
  return tracegen_weierstrass(ctx, curve, true, events, n_events, fixed_log2_rows, blu, out, "zkm_tracegen_weierstrass_double");
}

// <Curve>Decompress: the curve's b and the generator's x (the padding rows' input) as 32-bit limbs (crates/curves/src/weierstrass/secp256k1.rs,
// secp256r1.rs, bls12_381.rs: WeierstrassParameters::B, GENERATOR); indexed like k_curves8 (Bn254 has no decompress chip)
static const uint32_t k_decompress_b8[2][8] = {
  {0x00000007u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u},
  {0x27d2604bu, 0x3bce3c3eu, 0xcc53b0f6u, 0x651d06b0u, 0x769886bcu, 0xb3ebbd55u, 0xaa3a93e7u, 0x5ac635d8u}};
static const uint32_t k_decompress_gx8[2][8] = {
  {0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu},
  {0xd898c296u, 0xf4a13945u, 0x2deb33a0u, 0x77037d81u, 0x63a440f2u, 0xf8bce6e5u, 0xe12c4247u, 0x6b17d1f2u}};
static const uint32_t k_decompress_b12[12] = {4u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
static const uint32_t k_decompress_gx12[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                               0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};
extern "C++" {
template <int NL>
static tracegen::DecompressCurve<NL> decompress_curve(const bigfield::Modulus<NL>& m, const uint32_t* a, const uint32_t* b, const uint32_t* gx, int32_t offset) {
  tracegen::DecompressCurve<NL> c;
  c.f.m = m;
  c.f.witness_offset = offset;
  uint64_t carry = 1;      // (p + 1) / 4: the exponent of the square root for p = 3 (mod 4)
  uint32_t plus[NL + 1];
  for (int i = 0; i < NL; i++) { carry += m.p[i]; plus[i] = (uint32_t)carry; carry >>= 32; }
  plus[NL] = (uint32_t)carry;
  if ((m.p[0] & 3) != 3) throw std::runtime_error("decompress: the modulus is not 3 (mod 4)");
  for (int i = 0; i < NL; i++) { c.f.a[i] = a[i]; c.b[i] = b[i]; c.generator_x[i] = gx[i]; c.sqrt_exp[i] = plus[i] >> 2 | plus[i + 1] << 30; }
  return c;
}
template <int NL, bool LEX>
static void launch_weierstrass_decompress(zkm_ctx* ctx, const tracegen::DecompressCurve<NL>& curve, const uint32_t* d_events, size_t n_events, size_t height,
                                          uint32_t* out, uint32_t* counts, int* d_bad, double bytes) {
  KLAUNCH(ctx, "tracegen_weierstrass_decompress", bytes, (tracegen::weierstrass_decompress_rows<NL, LEX>), dim3(div_up(height, (size_t)tracegen::bf_threads(NL))),
          dim3(tracegen::bf_threads(NL)), tracegen::bf_lds_bytes(NL, counts != nullptr), d_events, n_events, height, out, counts, d_bad, curve);
}
}  // extern "C++"
int zkm_tracegen_weierstrass_decompress(zkm_ctx* ctx, int curve, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                                        zkm_matrix** out) {
  API_BEGIN
  const char* who = "zkm_tracegen_weierstrass_decompress";
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (curve != 0 && curve != 1 && curve != 3) throw std::runtime_error(std::string(who) + ": the curve is ZKM_CURVE_SECP256K1, ZKM_CURVE_SECP256R1 or ZKM_CURVE_BLS12381");
  if (n_events && !events) throw std::runtime_error(std::string(who) + ": null events");
  const int nl = curve == 3 ? 12 : 8, N = 4 * nl, W = nl, G = 6 * N - 4;
  const bool lex = curve == 3;
  const size_t ev_words = 4 + 11 * W, width = 5 + 22 * W + (N + 2) + 4 * G + (G + N + 3) + G + (lex ? 2 * (N + 2) + 3 : 0);
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, who);
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = width;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * ev_words * 4;
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    const double bytes = (double)ev_bytes + 4.0 * height * m->w;
    if (nl == 8)
      launch_weierstrass_decompress<8, false>(ctx, decompress_curve<8>(k_curves8[curve].m, k_curves8[curve].a, k_decompress_b8[curve], k_decompress_gx8[curve], 1 << 14),
                                              d_events, n_events, height, m->d, counts, d_bad, bytes);
    else
      launch_weierstrass_decompress<12, true>(ctx, decompress_curve<12>(k_curve_bls12381.m, k_curve_bls12381.a, k_decompress_b12, k_decompress_gx12, 1 << 15),
                                              d_events, n_events, height, m->d, counts, d_bad, bytes);
    // x_2 .. x_3_plus_b_plus_ax, the root's multiplication (its result columns hold the root; FieldSqrtCols also range-checks the product's
    // bytes, which are x_3_plus_b_plus_ax's result columns), neg_y
    const int x_2 = 5 + 22 * W + N + 2;
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{4, {x_2, x_2 + 4 * G, x_2 + 5 * G + N + 3, x_2 + 3 * G}, {4 * G, G, G, N}}, counts);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    static const char* reasons[] = {"", "the sign bit is not 0 or 1, or x is not below the field modulus", "x is not on the curve",
                                    "the words written to y do not write a root of x^3 + a x + b", "the words written to y are not the root the sign bit asks for"};
    if (bad) throw std::runtime_error(std::string(who) + ": " + reasons[std::min(std::max(16 - bad, 1), 4)]);
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}

extern "C++" {
template <int NL, int KIND>
static void launch_fp_tower(zkm_ctx* ctx, const tracegen::CurveField<NL>& field, const uint32_t* d_events, size_t n_events, size_t height, uint32_t* out,
                            uint32_t* counts, int* d_bad, double bytes) {
  KLAUNCH(ctx, KIND == 0 ? "tracegen_fp_op" : KIND == 1 ? "tracegen_fp2_addsub" : "tracegen_fp2_mul", bytes, (tracegen::fp_tower_rows<NL, KIND>),
          dim3(div_up(height, (size_t)tracegen::bf_threads(NL))), dim3(tracegen::bf_threads(NL)), tracegen::bf_lds_bytes(NL, counts != nullptr), d_events, n_events,
          height, out, counts,
          d_bad, field);
}
}  // extern "C++"
static int tracegen_fp_tower(zkm_ctx* ctx, int field, int kind, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu,
                             zkm_matrix** out, const char* who) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (field != 2 && field != 3) throw std::runtime_error(std::string(who) + ": the field is ZKM_CURVE_BN254 or ZKM_CURVE_BLS12381");
  if (n_events && !events) throw std::runtime_error(std::string(who) + ": null events");
  const int nl = field == 3 ? 12 : 8, W = kind == 0 ? nl : 2 * nl, G = 6 * 4 * nl - 4;
  const size_t ev_words = (kind == 2 ? 4 : 5) + 11 * W, width = (kind == 0 ? 8 : kind == 1 ? 6 : 5) + 22 * W + (kind == 0 ? 1 : kind == 1 ? 2 : 6) * G;
  const size_t height = padded_trace_rows(n_events, fixed_log2_rows, who);
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = width;
  uint32_t* d_events = nullptr;
  int* d_bad = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    const size_t ev_bytes = n_events * ev_words * 4;
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(ev_bytes, 4));
    d_bad = (int*)ctx->alloc(4);
    HIP_CHECK(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    if (n_events) HIP_CHECK(hipMemcpyAsync(d_events, events, ev_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* counts = blu ? blu->counts : nullptr;
    const double bytes = (double)ev_bytes + 4.0 * height * m->w;
    if (nl == 8) {
      tracegen::CurveField<8> f;
      f.m = k_curves8[2].m;
      memcpy(f.a, k_curves8[2].a, sizeof f.a);
      f.witness_offset = 1 << 14;
      if (kind == 0) launch_fp_tower<8, 0>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
      else if (kind == 1) launch_fp_tower<8, 1>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
      else launch_fp_tower<8, 2>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
    } else {
      tracegen::CurveField<12> f;
      f.m = k_curve_bls12381.m;
      memcpy(f.a, k_curve_bls12381.a, sizeof f.a);
      f.witness_offset = 1 << 15;
      if (kind == 0) launch_fp_tower<12, 0>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
      else if (kind == 1) launch_fp_tower<12, 1>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
      else launch_fp_tower<12, 2>(ctx, f, d_events, n_events, height, m->d, counts, d_bad, bytes);
    }
    count_u8_pairs(ctx, m, n_events, tracegen::U8Segments{1, {(kind == 0 ? 8 : kind == 1 ? 6 : 5) + 22 * W}, {(kind == 0 ? 1 : kind == 1 ? 2 : 6) * G}}, counts);
    int bad = 0;
    HIP_CHECK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("trace generation");
    ctx->end_timing(false);
    if (bad) throw std::runtime_error(std::string(who) + ": an operand is not below the field modulus, the operation is not one this chip has, or the words written to x are not the result");
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (d_bad) ctx->release(d_bad);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  ctx->release(d_bad);
  *out = m;
  API_END
}
int zkm_tracegen_fp_op(zkm_ctx* ctx, int field, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  return tracegen_fp_tower(ctx, field, 0, events, n_events, fixed_log2_rows, blu, out, "zkm_tracegen_fp_op");
}
int zkm_tracegen_fp2_addsub(zkm_ctx* ctx, int field, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  return tracegen_fp_tower(ctx, field, 1, events, n_events, fixed_log2_rows, blu, out, "zkm_tracegen_fp2_addsub");
}
int zkm_tracegen_fp2_mul(zkm_ctx* ctx, int field, const void* events, size_t n_events, int fixed_log2_rows, zkm_byte_lookups* blu, zkm_matrix** out) {
  return tracegen_fp_tower(ctx, field, 2, events, n_events, fixed_log2_rows, blu, out, "zkm_tracegen_fp2_mul");
}

int zkm_tracegen_exp_reverse_bits(zkm_ctx* ctx, const uint32_t* bases, const uint32_t* bits, const uint32_t* offsets, size_t n_events,
                                  int fixed_log2_rows, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && (!bases || !bits || !offsets)) throw std::runtime_error("zkm_tracegen_exp_reverse_bits: null events");
  const size_t rows = n_events ? offsets[n_events] : 0;
  for (size_t e = 0; e < n_events; e++)
    if (offsets[e + 1] < offsets[e]) throw std::runtime_error("zkm_tracegen_exp_reverse_bits: offsets must not decrease");
  const size_t height = padded_trace_rows(rows, fixed_log2_rows, "zkm_tracegen_exp_reverse_bits");
  ctx->begin_call();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::EXP_REVERSE_BITS_WIDTH;
  uint32_t *d_bases = nullptr, *d_bits = nullptr, *d_off = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    HIP_CHECK(hipMemsetAsync(m->d, 0, height * m->w * 4, ctx->stream));
    d_bases = ctx->alloc_n<uint32_t>(std::max<size_t>(n_events, 1));
    d_bits = ctx->alloc_n<uint32_t>(std::max<size_t>(rows, 1));
    d_off = ctx->alloc_n<uint32_t>(n_events + 1);
    if (n_events) {
      HIP_CHECK(hipMemcpyAsync(d_bases, bases, n_events * 4, hipMemcpyHostToDevice, ctx->stream));
      if (rows) HIP_CHECK(hipMemcpyAsync(d_bits, bits, rows * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_CHECK(hipMemcpyAsync(d_off, offsets, (n_events + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(tracegen::exp_reverse_bits_rows, dim3(div_up(n_events, (size_t)256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_bases,
                         (const uint32_t*)d_bits, (const uint32_t*)d_off, n_events, rows, height, m->d);
      LAUNCH_CHECK();
    }
    ctx->sync(ctx->stream);
  } catch (...) {
    for (uint32_t* p : {d_bases, d_bits, d_off, m->d})
      if (p) ctx->release(p);
    delete m;
    throw;
  }
  ctx->release(d_bases);
  ctx->release(d_bits);
  ctx->release(d_off);
  *out = m;
  API_END
}

// ExpReverseBitsLen inside a batch (ZKM_TG_EXP_REVERSE_BITS): one buffer [bases (n) | offsets (n + 1) | bits (rows)] — a host pointer or
// a prefetched address; `rows` = offsets[n], given by the caller (the offsets may already be in HBM). The kernel bounds every row it
// writes by the height, so offsets that disagree with `rows` cannot write outside the matrix.
static zkm_matrix* enqueue_exp_reverse_bits(TraceBatch& b, const uint32_t* packed, size_t n_events, size_t rows, int fixed_log2_rows) {
  zkm_ctx* ctx = b.ctx;
  if (n_events && !packed) throw std::runtime_error("zkm_tracegen_exp_reverse_bits: null events");
  const size_t height = padded_trace_rows(rows, fixed_log2_rows, "zkm_tracegen_exp_reverse_bits");
  zkm_matrix* m = b.matrix(height, tracegen::EXP_REVERSE_BITS_WIDTH);
  HIP_CHECK(hipMemsetAsync(m->d, 0, height * m->w * 4, ctx->stream));
  m->uniform_rows = n_events == 0;
  if (n_events) {
    const uint32_t* d = b.events(packed, (2 * n_events + 1 + rows) * 4);
    ctx->flush_staged();
    hipLaunchKernelGGL(tracegen::exp_reverse_bits_rows, dim3(div_up(n_events, (size_t)256)), dim3(256), 0, ctx->stream, d, d + 2 * n_events + 1,
                       d + n_events, n_events, rows, height, m->d);
    LAUNCH_CHECK();
  }
  return m;
}

int zkm_tracegen_poseidon2_skinny(zkm_ctx* ctx, const uint32_t* events, size_t n_events, int fixed_log2_rows, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n_events && !events) throw std::runtime_error("zkm_tracegen_poseidon2_skinny: null events");
  const size_t height = padded_trace_rows(n_events * tracegen::SKINNY_ROWS, fixed_log2_rows, "zkm_tracegen_poseidon2_skinny");
  ctx->begin_timing();
  zkm_matrix* m = new zkm_matrix();
  m->h = height; m->w = tracegen::SKINNY_WIDTH;
  uint32_t* d_events = nullptr;
  try {
    m->d = ctx->alloc_n<uint32_t>(height * m->w);
    HIP_CHECK(hipMemsetAsync(m->d, 0, height * m->w * 4, ctx->stream));
    d_events = (uint32_t*)ctx->alloc(std::max<size_t>(n_events * 128, 4));
    if (n_events) {
      HIP_CHECK(hipMemcpyAsync(d_events, events, n_events * 128, hipMemcpyHostToDevice, ctx->stream));
      KLAUNCH(ctx, "tracegen_poseidon2_skinny", 128.0 * n_events + 4.0 * n_events * tracegen::SKINNY_ROWS * tracegen::SKINNY_WIDTH,
              tracegen::poseidon2_skinny_rows, dim3(div_up(n_events, (size_t)tracegen::THREADS)), dim3(tracegen::THREADS), 0,
              (const uint32_t*)d_events, n_events, height, m->d);
    }
    ctx->mark("trace generation");
    ctx->end_timing(false);
  } catch (...) {
    if (d_events) ctx->release(d_events);
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  ctx->release(d_events);
  *out = m;
  API_END
}

int zkm_tracegen_byte_table(zkm_ctx* ctx, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  ctx->begin_call();
  zkm_matrix* m = new zkm_matrix();
  m->h = tracegen::BYTE_ROWS; m->w = tracegen::BYTE_PREP_COLS;
  try {
    m->d = ctx->alloc_n<uint32_t>(m->h * m->w);
    hipLaunchKernelGGL(tracegen::byte_table, dim3(tracegen::BYTE_ROWS / 256), dim3(256), 0, ctx->stream, m->d);
    LAUNCH_CHECK();
    ctx->sync(ctx->stream);
  } catch (...) {
    if (m->d) ctx->release(m->d);
    delete m;
    throw;
  }
  *out = m;
  API_END
}

int zkm_byte_lookups_create(zkm_ctx* ctx, zkm_byte_lookups** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  const size_t cells = (size_t)tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS;
  zkm_byte_lookups* b = new zkm_byte_lookups();
  try {
    b->counts = ctx->alloc_n<uint32_t>(cells);
    HIP_CHECK(hipMemsetAsync(b->counts, 0, cells * 4, ctx->stream));
  } catch (...) {
    delete b;
    throw;
  }
  *out = b;
  API_END
}
void zkm_byte_lookups_free(zkm_ctx* ctx, zkm_byte_lookups* b) {
  if (!b) return;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->release(b->counts);
  delete b;
}

static zkm_matrix* enqueue_byte_mults(TraceBatch& b, const zkm_byte_lookups* blu, const uint32_t* extra_counts) {
  zkm_ctx* ctx = b.ctx;
  if (!blu) throw std::runtime_error("zkm_tracegen_byte_mults: null byte lookups");
  const size_t cells = (size_t)tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS;
  zkm_matrix* m = b.matrix(tracegen::BYTE_ROWS, tracegen::NUM_BYTE_OPS);
  uint32_t* d_extra = nullptr;
  if (extra_counts) {
    d_extra = (uint32_t*)b.temp(cells * 4);
    HIP_CHECK(hipMemcpyAsync(d_extra, extra_counts, cells * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  ctx->flush_staged();
  hipLaunchKernelGGL(tracegen::byte_mults_finish, dim3(div_up(cells, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)blu->counts,
                     (const uint32_t*)d_extra, m->d, cells);
  LAUNCH_CHECK();
  return m;
}

int zkm_tracegen_byte_mults(zkm_ctx* ctx, const zkm_byte_lookups* blu, const uint32_t* extra_counts, zkm_matrix** out) {
  return tracegen_single(ctx, out, [&](TraceBatch& b) { return enqueue_byte_mults(b, blu, extra_counts); });
}

// generate_traces of a core shard (crates/stark/src/prover.rs:70-108) in one call: every descriptor's generator is queued on the compute
// stream behind the prefetch of its events, none waits for the host, and the call synchronises once at the end. Byte lookups of all the
// chips are counted into one table (the caller's, or one that lives for the call); a ZKM_TG_BYTE_MULTS descriptor reads it after every
// other generator, wherever it stands in the list; ZKM_TG_PROGRAM_MULTS is filled by the ZKM_TG_CPU descriptor's pass over the events.
int zkm_tracegen_shard(zkm_ctx* ctx, const zkm_tracegen_desc* descs, size_t n, zkm_byte_lookups* blu, zkm_matrix** out) {
  API_BEGIN
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_CHECK(hipSetDevice(ctx->device));
  CallScope call_scope(ctx);
  if (n && (!descs || !out)) throw std::runtime_error("zkm_tracegen_shard: null descriptors");
  ctx->begin_timing();
  TraceBatch b(ctx);
  zkm_byte_lookups own;
  try {
    if (!blu) {
      const size_t cells = (size_t)tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS;
      own.counts = (uint32_t*)b.temp(cells * 4);
      HIP_CHECK(hipMemsetAsync(own.counts, 0, cells * 4, ctx->stream));
      blu = &own;
    }
    long cpu_at = -1, pm_at = -1;
    for (size_t i = 0; i < n; i++) {
      out[i] = nullptr;
      if (descs[i].kind == ZKM_TG_CPU) { if (cpu_at >= 0) throw std::runtime_error("zkm_tracegen_shard: two Cpu descriptors"); cpu_at = (long)i; }
      if (descs[i].kind == ZKM_TG_PROGRAM_MULTS) { if (pm_at >= 0) throw std::runtime_error("zkm_tracegen_shard: two Program descriptors"); pm_at = (long)i; }
    }
    if (pm_at >= 0 && cpu_at < 0) throw std::runtime_error("zkm_tracegen_shard: the Program multiplicities come from the Cpu descriptor's events, and there is none");
    for (size_t i = 0; i < n; i++) {
      const zkm_tracegen_desc& d = descs[i];
      zkm_byte_lookups* use = d.no_byte_lookups ? nullptr : blu;
      switch (d.kind) {
        case ZKM_TG_ALU:
          if (d.chip < 0 || d.chip >= tracegen::NUM_ALU_CHIPS) throw std::runtime_error("zkm_tracegen_shard: unknown ALU chip");
          out[i] = enqueue_rows(b, d.chip, d.events, d.n_events, d.fixed_log2_rows, use);
          break;
        case ZKM_TG_JUMP: out[i] = enqueue_rows(b, tracegen::JUMP, d.events, d.n_events, d.fixed_log2_rows, nullptr); break;
        case ZKM_TG_MOV_COND: out[i] = enqueue_rows(b, tracegen::MOV_COND, d.events, d.n_events, d.fixed_log2_rows, nullptr); break;
        case ZKM_TG_BRANCH: out[i] = enqueue_rows(b, tracegen::BRANCH, d.events, d.n_events, d.fixed_log2_rows, use); break;
        case ZKM_TG_MUL: out[i] = enqueue_rows(b, tracegen::MUL, d.events, d.n_events, d.fixed_log2_rows, use); break;
        case ZKM_TG_DIVREM: out[i] = enqueue_rows(b, tracegen::DIVREM, d.events, d.n_events, d.fixed_log2_rows, use); break;
        case ZKM_TG_MEMORY_INSTRS: out[i] = enqueue_rows(b, tracegen::MEMORY_INSTRS, d.events, d.n_events, d.fixed_log2_rows, use); break;
        case ZKM_TG_MISC_INSTRS: out[i] = enqueue_rows(b, tracegen::MISC_INSTRS, d.events, d.n_events, d.fixed_log2_rows, use); break;
        case ZKM_TG_SYSCALL_INSTRS: out[i] = enqueue_rows(b, tracegen::SYSCALL_INSTRS, d.events, d.n_events, d.fixed_log2_rows, nullptr); break;
        case ZKM_TG_SYSCALL_CORE: out[i] = enqueue_syscall(b, (const zkm_syscall_event*)d.events, d.n_events, 0, d.fixed_log2_rows, use); break;
        case ZKM_TG_SYSCALL_PRECOMPILE: out[i] = enqueue_syscall(b, (const zkm_syscall_event*)d.events, d.n_events, 1, d.fixed_log2_rows, use); break;
        case ZKM_TG_MEMORY_LOCAL: out[i] = enqueue_memory_local(b, (const zkm_memory_local_event*)d.events, d.n_events, d.fixed_log2_rows); break;
        case ZKM_TG_GLOBAL: out[i] = enqueue_global(b, (const zkm_global_lookup_event*)d.events, d.n_events, d.fixed_log2_rows, blu); break;
        case ZKM_TG_CPU:
          out[i] = enqueue_cpu(b, (const zkm_cpu_event*)d.events, d.n_events, d.program, d.n_instr, d.pc_base, d.shard, d.fixed_log2_rows,
                               pm_at >= 0 ? descs[pm_at].fixed_log2_rows : -1, use, pm_at >= 0 ? &out[pm_at] : nullptr);
          break;
        case ZKM_TG_FLAT:
          if (d.chip <= 0) throw std::runtime_error("zkm_tracegen_shard: a flat descriptor carries the trace width in `chip`");
          out[i] = enqueue_flat(b, (const uint32_t*)d.events, d.n_events, (size_t)d.chip, d.fixed_log2_rows);
          break;
        case ZKM_TG_POSEIDON2_WIDE: out[i] = enqueue_poseidon2_wide(b, (const uint32_t*)d.events, d.n_events, d.fixed_log2_rows); break;
        case ZKM_TG_EXP_REVERSE_BITS: out[i] = enqueue_exp_reverse_bits(b, (const uint32_t*)d.events, d.n_events, d.n_instr, d.fixed_log2_rows); break;
        case ZKM_TG_PROGRAM_MULTS: case ZKM_TG_BYTE_MULTS: break;
        default: throw std::runtime_error("zkm_tracegen_shard: unknown descriptor kind");
      }
    }
    for (size_t i = 0; i < n; i++)
      if (descs[i].kind == ZKM_TG_BYTE_MULTS) out[i] = enqueue_byte_mults(b, blu, nullptr);
    b.finish("trace generation");
  } catch (...) {
    b.abort();
    for (size_t i = 0; i < n; i++) out[i] = nullptr;
    throw;
  }
  API_END
}

