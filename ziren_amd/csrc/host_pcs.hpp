// Part of libzkm_hip.so's host side (one translation unit: csrc/zkm_hip.hip includes this file). Pcs::commit on the device: coset LDE dispatch, Merkle tree construction (mixed heights, fused leaf kernel, lane-parallel top), pcs_commit.
#pragma once
// ---- device helpers ------------------------------------------------------------------------------
// Coset LDE of a batch of column-major matrices (any heights), one launch per kernel for the whole batch (lde.cuh: Batch).
struct LdeJob {
  const uint32_t* in; size_t n, w; uint32_t lde_shift; uint32_t* out;
  uint32_t* cflag = nullptr;   // optional: 2 w words, zeroed by the caller, that keep the constant-column flags (lde::Mat::cflag) after the batch — written
                               // only for four-step matrices (n > 2^LOG_ROW_MAX); without it the batch uses scratch of its own
  bool uniform = false;        // zkm_matrix::uniform_rows: the trace generator made every row the same (a chip without events): the strided inverse pass
                               // does not read the matrix to find that out
};

static void lde_batch_chunk(zkm_ctx* ctx, const std::vector<LdeJob>& jobs, int bl) {
  // groups = distinct heights, tallest first (the biggest blocks are queued first)
  std::vector<size_t> order(jobs.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].n > jobs[b].n; });
  std::vector<lde::Batch> hb(1);
  lde::Batch& b = hb[0];
  memset(&b, 0, sizeof b);
  b.log_blowup = bl;
  const size_t cosets = (size_t)1 << bl;
  size_t tmp1_words = 0, tmp2_words = 0;
  for (size_t oi : order) if (jobs[oi].n > ((size_t)1 << lde::LOG_ROW_MAX)) { tmp1_words += jobs[oi].n * jobs[oi].w; tmp2_words += (jobs[oi].n * jobs[oi].w) << bl; }
  uint32_t* tmp1 = tmp1_words ? ctx->alloc_n<uint32_t>(tmp1_words) : nullptr;
  uint32_t* tmp2 = tmp2_words ? ctx->alloc_n<uint32_t>(tmp2_words) : nullptr;
  size_t t1 = 0, t2 = 0;
  // constant-column flags of the four-step matrices (lde::Mat::cflag): two words per column, zero = "constant so far"
  size_t big_cols = 0, fc = 0;
  for (size_t oi : order) if (jobs[oi].n > ((size_t)1 << lde::LOG_ROW_MAX) && !jobs[oi].cflag) big_cols += jobs[oi].w;
  uint32_t* cflags = big_cols ? ctx->alloc_n<uint32_t>(2 * big_cols) : nullptr;
  if (cflags) HIP_CHECK(hipMemsetAsync(cflags, 0, 2 * big_cols * sizeof(uint32_t), ctx->cur));
  uint32_t blk[4] = {0, 0, 0, 0};
  size_t lds_cols = 0, lds_small = 0;
  double bytes_inv = 0, bytes_rows = 0, bytes_fwd = 0, bytes_small = 0;
  int nm = 0;
  for (size_t p = 0; p < order.size();) {
    const size_t n = jobs[order[p]].n;
    const int k = log2_strict(n);
    const int lb = std::min(k, lde::LOG_ROW_MAX), la = k - lb;
    lde::Group& g = b.g[b.n_groups];
    g.la = la; g.lb = lb;
    // every block of the strided passes takes a tile of 2^14 points (A x T), whatever the height: one LDS size (~66 KiB, two blocks per
    // CU) and one amount of work per block for the whole batch; shorter columns get wider tiles (longer contiguous segments)
    g.logT = la ? std::min(14 - la, lb) : 0;  // la >= 1 implies lb = 13, so T >= 8
    g.first_mat = nm;
    g.w_N = kb::two_adic_generator(k + bl);
    g.n_inv = kb::inv(kb::to_monty((uint32_t)(n % kb::P)));
    g.twb_inv = lb > 0 ? ctx->twiddles(lb, true) : nullptr;
    g.twb_fwd = lb > 0 ? ctx->twiddles(lb, false) : nullptr;
    if (la) {
      g.twa_inv = ctx->twiddles_ct(la, true);
      g.twa_fwd = ctx->twiddles_ct(la, false);
      auto pw = ctx->pow_tables(k);
      g.pw_lo = pw.first; g.pw_hi = pw.second;
      g.tw_rows = ctx->row_twiddles(k);
    }
    uint32_t cols = 0;
    for (; p < order.size() && jobs[order[p]].n == n; p++) {
      const LdeJob& j = jobs[order[p]];
      lde::Mat& m = b.m[nm++];
      m.in = j.in; m.out = j.out; m.col0 = cols; m.w = (uint32_t)j.w; m.shift = j.lde_shift; m.uniform = j.uniform ? 1u : 0u;
      if (la) {
        m.tmp1 = tmp1 + t1; t1 += n * j.w;
        m.tmp2 = tmp2 + t2; t2 += (n * j.w) << bl;
        if (j.cflag) m.cflag = j.cflag;
        else { m.cflag = cflags + fc; fc += 2 * j.w; }
        auto ct = ctx->coset_tables(k, bl, j.lde_shift);
        m.twf = ct.twf; m.cs = ct.cs;
      }
      cols += (uint32_t)j.w;
    }
    g.n_cols = cols;
    g.n_mats = nm - g.first_mat;
    const double cells = (double)n * cols;
    if (la) {
      const size_t per_col = ((size_t)1 << lb) >> g.logT;
      blk[lde::K_COLS_INV] += (uint32_t)(per_col * cols);
      blk[lde::K_ROWS_BIG] += (uint32_t)(((size_t)1 << la) * cols);
      blk[lde::K_COLS_FWD] += (uint32_t)(per_col * cols * cosets);
      lds_cols = std::max(lds_cols, ((size_t)1 << la) * (((size_t)1 << g.logT) + 1) * 4);
      bytes_inv += 8.0 * cells; bytes_rows += 4.0 * cells * (1 + cosets); bytes_fwd += 8.0 * cells * cosets;
      for (int mi = (int)g.first_mat; mi < nm; mi++) if (b.m[mi].uniform) bytes_inv -= 8.0 * (double)n * b.m[mi].w;
    } else {
      blk[lde::K_ROWS_SMALL] += cols;
      const size_t B = (size_t)1 << lb;
      lds_small = std::max(lds_small, (2 * (B + (B >> 5)) + 64 + (B > 64 ? (B >> 6) : 1)) * 4);
      bytes_small += 4.0 * cells * (1 + cosets);
    }
    for (int q = 0; q < 4; q++) g.blk_end[q] = blk[q];
    b.n_groups++;
  }
  const lde::Batch* d = (const lde::Batch*)ctx->upload_staged(&b, sizeof b);
  if (blk[lde::K_COLS_INV]) {
    constexpr size_t B = (size_t)1 << lde::LOG_ROW_MAX;
    const size_t big_lds = (B + (B >> 5)) * 4;
    KLAUNCH(ctx, "lde_cols_inverse", bytes_inv, lde::lde_cols<false>, dim3(blk[lde::K_COLS_INV]), dim3(lde::THREADS), lds_cols, d);
    KLAUNCH(ctx, "lde_rows", bytes_rows, lde::lde_rows_big, dim3(blk[lde::K_ROWS_BIG]), dim3(lde::THREADS), big_lds, d);
    KLAUNCH(ctx, "lde_cols_forward", bytes_fwd, lde::lde_cols<true>, dim3(blk[lde::K_COLS_FWD]), dim3(lde::THREADS), lds_cols, d);
  }
  if (blk[lde::K_ROWS_SMALL])
    KLAUNCH(ctx, "lde_rows", bytes_small, lde::lde_rows, dim3(blk[lde::K_ROWS_SMALL]), dim3(lde::THREADS), lds_small, d);
  ctx->release_here(tmp1);
  ctx->release_here(tmp2);
  ctx->release_here(cflags);
  ctx->release_here((void*)d);
}

static void lde_batch(zkm_ctx* ctx, const std::vector<LdeJob>& all, int bl) {
  // a descriptor holds MAX_MATS matrices and MAX_GROUPS heights (24 covers every power of two up to the field's two-adicity minus the blow-up)
  // ... and a chunk's scratch (the four-step matrices' intermediates: n w words after the inverse strided pass, n w << bl before the
  // forward one) stays under a budget: batching exists to fill the GPU with one launch per kernel, which a few GB of cells already do,
  // while the scratch of a whole commit of several billion cells (1.5x its LDE bytes, cached by the exact-size pool afterwards) would
  // not be small beside the LDEs themselves. A single matrix larger than the budget still goes alone.
  constexpr size_t SCRATCH_BUDGET_WORDS = ((size_t)12 << 30) / 4;
  std::vector<LdeJob> jobs;
  size_t scratch = 0;
  for (auto& j : all) {
    if (j.w == 0) continue;
    const size_t need = j.n > ((size_t)1 << lde::LOG_ROW_MAX) ? j.n * j.w + ((j.n * j.w) << bl) : 0;
    if (!jobs.empty() && ((int)jobs.size() == lde::MAX_MATS || scratch + need > SCRATCH_BUDGET_WORDS)) { lde_batch_chunk(ctx, jobs, bl); jobs.clear(); scratch = 0; }
    jobs.push_back(j);
    scratch += need;
  }
  if (!jobs.empty()) lde_batch_chunk(ctx, jobs, bl);
}

static void lde_columns(zkm_ctx* ctx, const uint32_t* in, size_t n, size_t w, int bl, uint32_t lde_shift, uint32_t* out) {
  lde_batch(ctx, {LdeJob{in, n, w, lde_shift, out}}, bl);
}

// Upload an array of device pointers (one per column) and return the device copy.
static const uint32_t** upload_ptrs(zkm_ctx* ctx, const std::vector<const uint32_t*>& ptrs) {
  return (const uint32_t**)ctx->upload_staged(ptrs.data(), ptrs.size() * sizeof(void*));
}

// Layers of at most LANES_MAX nodes without injection: lane-parallel compression; returns true when it
// finished the tree (tail launch, which also writes the root to pinned host memory: t.h_root, valid after the next stream
// synchronisation), false when the caller should go on with the next layer.
static bool compress_small_layer(zkm_ctx* ctx, Tree& t, int layer, size_t len, merkle::DevChallenger* d_ch = nullptr, kb::E4* d_beta = nullptr,
                                 uint32_t* h_root_slot = nullptr) {
  // ZKM_LANES_MAX: the largest layer the lane-parallel kernel takes (A/B knob; 4096 is the measured crossover: at 8192 nodes two waves per
  // SIMD share the issue slots and the level is no faster than the thread-per-node kernel's serial permutation)
  static const size_t LANES_MAX = getenv("ZKM_LANES_MAX") ? (size_t)atol(getenv("ZKM_LANES_MAX")) : 4096;
  const size_t TAIL = 64;
  if (len > LANES_MAX) {
    KLAUNCH(ctx, "compress_layer", 96.0 * len, merkle::compress_layer, dim3(div_up(len, merkle::THREADS)), dim3(merkle::THREADS), 0,
            (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8, len,
            (const uint32_t* const*)nullptr, 0, (const uint32_t*)nullptr);
    return false;
  }
  if (len <= TAIL) {
    // d_ch (an FRI commit-phase tree): the tail observes the root into the device challenger and samples beta (merkle.cuh)
    uint32_t* h_root = h_root_slot ? h_root_slot : (uint32_t*)ctx->pin_alloc(32);
    if (h_root) for (int k = 0; k < 8; k++) ((volatile uint32_t*)h_root)[k] = 0xffffffffu;   // not a field word: wait_root can watch the root arrive
    KLAUNCH(ctx, "compress_tail", 96.0 * len, merkle::compress_tail_lanes, dim3(1), dim3(1024), 0, t.digests + t.layer_off[layer] * 8, len, h_root, d_ch, d_beta);
    t.h_root = h_root;
    return true;
  }
  KLAUNCH(ctx, "compress_small", 96.0 * len, merkle::compress_layer_lanes, dim3(div_up(len * 16, merkle::THREADS)), dim3(merkle::THREADS),
          0, (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8, len);
  return false;
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  __asm__ __volatile__("yield");
#endif
}

// The root a tail launch writes to page-locked host memory, read without a stream synchronisation (ZKM_ROOT_POLL=0: with one): the host watches the
// eight words change from the 0xffffffff they were set to before the launch (a Montgomery word is below 2^31; each word is one 32-bit
// store to coherent host memory) and goes on queueing the next kernels behind the still-finishing launch — the completion signal, the
// wake-up and the return through hipStreamSynchronize stay off the critical path of the 22 dependent FRI layers. Falls back to the
// synchronisation after a second (a failed launch must still surface).
static void wait_root(zkm_ctx* ctx, const uint32_t* h_root, bool pollable) {
  if (ctx->root_poll && pollable) {
    const volatile uint32_t* v = h_root;
    auto t0 = std::chrono::steady_clock::now();
    zkm_ctx::TimerSlack slack;
    for (int spins = 0;; spins++) {
      bool all = true;
      for (int k = 0; k < 8; k++) all &= v[k] != 0xffffffffu;
      if (all) { std::atomic_thread_fence(std::memory_order_acquire); return; }
      if ((ctx->host_wait_blocking || (spins & 4095) == 4095) && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) {
        // a second without the root. "Not finished yet" and "finished but not visible" are different things: on an oversubscribed device
        // (eight ranks sharing one GPU, a tail launch queued behind seconds of another lane's work) the launch may simply not have run.
        // Only a stream that HAS completed while the words are still 0xffffffff means the stores do not reach the host here (non-coherent
        // host memory): then every further root would cost the same wait, and this context synchronises from now on.
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipErrorNotReady) { t0 = std::chrono::steady_clock::now(); continue; }     // still running or queued: keep watching
        if (q != hipSuccess) HIP_CHECK(q);                                                  // a failed launch surfaces here
        bool now = true;
        for (int k = 0; k < 8; k++) now &= v[k] != 0xffffffffu;
        if (now) { std::atomic_thread_fence(std::memory_order_acquire); return; }
        ctx->root_poll = false;
        g_err = "a finished launch's tree root is not visible in page-locked memory: root polling is off for this context (stream synchronisation instead)";
        break;
      }
      // after a short burst of spinning give the core away: a lane's host thread shares its CPUs with the other lanes and ranks of the node
      // a blocking context looks at the root between short sleeps; a spinning one spins, and past a burst gives the core away when asked
      if (ctx->host_wait_blocking) { if (spins >= 64) zkm_ctx::sleep_ns(std::min(ctx->wait_sleep_ns, 10000), slack); else cpu_relax(); }
      else if (spins < ctx->root_spin_before_yield) cpu_relax();
      else if (ctx->root_sleep_ns > 0) zkm_ctx::sleep_ns(ctx->root_sleep_ns, slack);
      else sched_yield();
    }
  }
  ctx->sync(ctx->stream);
}

// MerkleTreeMmcs::commit over column-major matrices of power-of-two heights.
static void build_tree(zkm_ctx* ctx, const std::vector<zkm_matrix>& mats, Tree& t,
                       const std::function<void(size_t)>& prepare_height = nullptr, const std::vector<const uint32_t*>* col_flags = nullptr,
                       bool rows_up_front = false) {
  // rows_up_front (pcs_commit, when every matrix is on the device already): the rows of all shorter heights are hashed by ONE launch
  // (merkle::hash_rows) right after the leaves, and a layer with injection takes two permutations per node
  // (merkle::compress_layer_rowdig) instead of running the row's sponge inside compress_layer.
  // col_flags (pcs_commit): per matrix the constant-column flags its LDE left (two words per column, lde::Mat::cflag), or null. Where
  // the row injected at a layer starts with constant columns, the sponge over them is computed once (merkle::sponge_prefix).
  // prepare_height(h), when given, is called right before the matrices of height h are first read: pcs_commit extends
  // them there, so a commit's kernels are queued tallest matrix first, layer by layer (extend, hash, extend the next
  // height, inject, ...), and whatever is still arriving over PCIe is only waited for when its layer is reached
  auto wait_height = [&](size_t h) {
    if (prepare_height) prepare_height(h);
  };
  size_t maxh = 0;
  for (auto& m : mats) maxh = std::max(maxh, m.h);
  t.max_height = maxh;
  t.log_max = log2_strict(maxh);
  t.layer_off.clear();
  size_t off = 0;
  for (size_t len = maxh; len >= 1; len >>= 1) { t.layer_off.push_back(off); off += len; if (len == 1) break; }
  t.digests = ctx->alloc_n<uint32_t>(off * 8);
  auto cols_of_height = [&](size_t h) {
    std::vector<const uint32_t*> ptrs;
    for (auto& m : mats)
      if (m.h == h)
        for (size_t c = 0; c < m.w; c++) ptrs.push_back(m.d + c * m.h);
    return ptrs;
  };
  std::vector<const uint32_t**> to_free;
  // the flag-pointer table of a height below the top whose first matrix kept flags (one entry per column of the concatenated row)
  std::map<size_t, const uint32_t**> flag_tables;
  uint32_t* prefixes = nullptr;   // 32 words per such height: sponge_prefix's output
  if (col_flags) {
    size_t n_tab = 0;
    for (size_t i = 0; i < mats.size(); i++) {
      const size_t h = mats[i].h;
      if (h == maxh || flag_tables.count(h)) continue;
      size_t first = 0;
      while (mats[first].h != h) first++;
      if (!(*col_flags)[first]) { flag_tables[h] = nullptr; continue; }
      std::vector<const uint32_t*> fp;
      for (size_t j = 0; j < mats.size(); j++)
        if (mats[j].h == h)
          for (size_t c = 0; c < mats[j].w; c++) fp.push_back((*col_flags)[j] ? (*col_flags)[j] + 2 * c : nullptr);
      flag_tables[h] = upload_ptrs(ctx, fp);
      to_free.push_back(flag_tables[h]);
      n_tab++;
    }
    if (n_tab) prefixes = ctx->alloc_n<uint32_t>(32 * n_tab);
  }
  size_t prefix_no = 0;
  // the column-pointer table of every height is staged before the first launch (the LDE buffers exist already): one transfer per tree
  std::map<size_t, std::pair<const uint32_t**, size_t>> tables;
  for (auto& m : mats)
    if (!tables.count(m.h)) {
      auto ptrs = cols_of_height(m.h);
      const uint32_t** d = ptrs.empty() ? nullptr : upload_ptrs(ctx, ptrs);
      if (d) to_free.push_back(d);
      tables[m.h] = {d, ptrs.size()};
    }
  // the tree levels right above the leaves that no shorter matrix is injected into can be reduced inside the leaf kernel's blocks
  int fuse = 0;
  if (maxh >= (size_t)merkle::FUSE_LEAVES) {
    while (fuse < merkle::FUSE_MAX_LEVELS && (maxh >> (fuse + 1)) >= 1) {
      bool injected = false;
      for (auto& m : mats) injected |= m.h == (maxh >> (fuse + 1));
      if (injected) break;
      fuse++;
    }
  }
  {
    const uint32_t** d = tables[maxh].first;
    struct { size_t n; size_t size() const { return n; } } ptrs{tables[maxh].second};
    wait_height(maxh);
    if (fuse > 0)
      KLAUNCH(ctx, "hash_leaves_tree", 4.0 * maxh * ptrs.size() + 32.0 * maxh * (2.0 - 1.0 / (1 << fuse)), merkle::hash_leaves_tree,
              dim3(maxh / merkle::FUSE_LEAVES), dim3(merkle::FUSE_LEAVES), merkle::FUSE_LEAVES * 12 * sizeof(uint32_t), d, (int)ptrs.size(), maxh,
              t.digests, fuse);
    else
      KLAUNCH(ctx, "hash_leaves", 4.0 * maxh * ptrs.size() + 32.0 * maxh, merkle::hash_leaves, dim3(div_up(maxh, merkle::THREADS)),
              dim3(merkle::THREADS), 0, d, (int)ptrs.size(), maxh, t.digests);
  }
  // near the root (no shorter matrix left to inject) layers switch to 16 lanes per node, and the last
  // <= 64-node layers go in one launch
  size_t min_h = maxh;
  for (auto& m : mats) min_h = std::min(min_h, m.h);
  // rows_up_front: every shorter height's row digests, one launch; groups ordered by permutations per row, longest first
  std::map<size_t, uint32_t*> rowdig;
  uint32_t* rowdig_buf = nullptr;
  if (rows_up_front && min_h < maxh) {
    std::vector<size_t> hs;
    size_t rows = 0;
    for (auto& kv : tables)
      if (kv.first < maxh && kv.second.first) { hs.push_back(kv.first); rows += kv.first; }
    std::sort(hs.begin(), hs.end(), [&](size_t a, size_t b) {
      const size_t pa = (tables[a].second + 7) / 8, pb = (tables[b].second + 7) / 8;
      return pa != pb ? pa > pb : a > b;
    });
    rowdig_buf = ctx->alloc_n<uint32_t>(rows * 8);
    std::vector<merkle::RowGroup> groups;
    size_t off = 0, blocks = 0;
    double bytes = 0;
    for (size_t h : hs) {
      wait_height(h);
      const uint32_t* prefix = nullptr;
      if (flag_tables.count(h) && flag_tables[h]) {   // after wait_height: the flags are final once the matrices' LDE is queued
        uint32_t* out = prefixes + 32 * prefix_no++;
        KLAUNCH(ctx, "sponge_prefix", 0.0, merkle::sponge_prefix, dim3(1), dim3(64), 0, (const uint32_t* const*)flag_tables[h], (int)tables[h].second, out);
        prefix = out;
      }
      rowdig[h] = rowdig_buf + off * 8;
      groups.push_back(merkle::RowGroup{(const uint32_t* const*)tables[h].first, prefix, rowdig[h], (uint32_t)h, (int)tables[h].second, (uint32_t)blocks});
      off += h;
      blocks += div_up(h, merkle::THREADS);
      bytes += 4.0 * h * tables[h].second + 32.0 * h;
    }
    const merkle::RowGroup* d_groups = (const merkle::RowGroup*)ctx->upload_staged(groups.data(), groups.size() * sizeof(merkle::RowGroup));
    to_free.push_back((const uint32_t**)d_groups);
    KLAUNCH(ctx, "hash_rows", bytes, merkle::hash_rows, dim3((unsigned)blocks), dim3(merkle::THREADS), 0, d_groups, (int)groups.size());
  }
  int layer = fuse;
  for (size_t len = maxh >> (fuse + 1); len >= 1; len >>= 1, layer++) {
    if (min_h > len) {
      if (compress_small_layer(ctx, t, layer, len)) break;
      continue;
    }
    if (rowdig.count(len)) {
      KLAUNCH(ctx, "compress_layer_rowdig", 96.0 * len + 32.0 * len, merkle::compress_layer_rowdig, dim3(div_up(len, merkle::THREADS)), dim3(merkle::THREADS), 0,
              (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8, len, (const uint32_t*)rowdig[len]);
      if (len == 1) break;
      continue;
    }
    const uint32_t** d = tables.count(len) ? tables[len].first : nullptr;
    struct { size_t n; size_t size() const { return n; } } ptrs{tables.count(len) ? tables[len].second : 0};
    if (d) wait_height(len);
    const uint32_t* prefix = nullptr;
    if (d && flag_tables.count(len) && flag_tables[len]) {   // after wait_height: the flags are final once the matrices' LDE is queued
      uint32_t* out = prefixes + 32 * prefix_no++;
      KLAUNCH(ctx, "sponge_prefix", 0.0, merkle::sponge_prefix, dim3(1), dim3(64), 0, (const uint32_t* const*)flag_tables[len], (int)ptrs.size(), out);
      prefix = out;
    }
    KLAUNCH(ctx, "compress_layer", 96.0 * len + 4.0 * len * ptrs.size(), merkle::compress_layer, dim3(div_up(len, merkle::THREADS)),
            dim3(merkle::THREADS), 0, (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8,
            len, (const uint32_t* const*)d, (int)ptrs.size(), prefix);
    if (len == 1) break;
  }
  for (auto d : to_free) ctx->release((void*)d);
  ctx->release(prefixes);
  ctx->release(rowdig_buf);
}

static void free_pcs_data(zkm_ctx* ctx, zkm_pcs_data* d) {
  if (!d) return;
  for (auto& m : d->ldes) ctx->release(m.d);
  for (auto& m : d->owned_evals) ctx->release(m.d);
  ctx->release(d->tree.digests);
  ctx->release(d->cflags);
  delete d;
}

// TwoAdicFriPcs::commit: LDE every matrix onto 3 * K (shift = GENERATOR / domain_shift), one tree.
static zkm_pcs_data* pcs_commit(zkm_ctx* ctx, const std::vector<zkm_matrix>& mats, const std::vector<uint32_t>& shifts,
                                int log_blowup) {
  zkm_pcs_data* d = new zkm_pcs_data();
  uint32_t* cflags = nullptr;
  try {
    d->log_blowup = log_blowup;
    for (size_t i = 0; i < mats.size(); i++) {
      const zkm_matrix& m = mats[i];
      zkm_matrix l;
      l.h = m.h << log_blowup;
      l.w = m.w;
      l.d = ctx->alloc_n<uint32_t>(l.h * l.w);
      d->ldes.push_back(l);
      d->evals.push_back(m.d);
      d->eval_heights.push_back(m.h);
      d->domain_shifts.push_back(shifts.empty() ? kb::ONE : shifts[i]);
    }
    // Matrices that are complete in HBM are all extended up front, one launch per kernel for the whole commit. A matrix still
    // arriving over PCIe (zkm_matrix_upload_async) is extended right before the tree layer that reads it (see build_tree), so only
    // the tallest trace's upload is exposed.
    auto shift_of = [&](size_t i) { return kb::mul(kb::GEN, kb::inv(d->domain_shifts[i])); };
    // constant-column flags of the four-step matrices, kept until the tree is built (build_tree: col_flags)
    size_t flag_words = 0;
    for (auto& m : mats) if (m.h > ((size_t)1 << lde::LOG_ROW_MAX)) flag_words += 2 * m.w;
    cflags = flag_words ? ctx->alloc_n<uint32_t>(flag_words) : nullptr;
    if (cflags) HIP_CHECK(hipMemsetAsync(cflags, 0, flag_words * sizeof(uint32_t), ctx->stream));
    std::vector<const uint32_t*> col_flags(mats.size(), nullptr);
    {
      size_t off = 0;
      for (size_t i = 0; i < mats.size(); i++)
        if (mats[i].h > ((size_t)1 << lde::LOG_ROW_MAX)) { col_flags[i] = cflags + off; off += 2 * mats[i].w; }
    }
    auto job_of = [&](size_t i) { return LdeJob{mats[i].d, mats[i].h, mats[i].w, shift_of(i), d->ldes[i].d, const_cast<uint32_t*>(col_flags[i]), mats[i].uniform_rows}; };
    std::vector<char> extended(mats.size(), 0);
    size_t top = 0;
    for (auto& l : d->ldes) top = std::max(top, l.h);
    {
      // The tallest matrices first, on the main stream: the leaf hashing that follows needs only them. The shorter ones are extended on
      // the side stream while the leaves are hashed (ZKM_LDE_OVERLAP, default on): the strided LDE passes run at the memory system's
      // pace and leave issue slots free, the leaf hashing is bound by issue and leaves the memory system idle. The tree joins the
      // side stream right before the first layer that reads a shorter matrix (extend_height below).
      const bool overlap = ctx->lde_overlap;
      std::vector<LdeJob> tall, rest;
      size_t rest_cells = 0;
      for (size_t i = 0; i < mats.size(); i++)
        if (!mats[i].ready || hipEventQuery(mats[i].ready) == hipSuccess) {   // never uploaded asynchronously, or landed already
          (d->ldes[i].h == top ? tall : rest).push_back(job_of(i));
          if (d->ldes[i].h != top) rest_cells += mats[i].h * mats[i].w;
          extended[i] = 1;
        }
      if (overlap && !tall.empty() && rest_cells >= ((size_t)1 << 22)) {
        lde_batch(ctx, tall, log_blowup);
        ctx->side_begin();
        try { lde_batch(ctx, rest, log_blowup); } catch (...) { ctx->side_end(); ctx->side_join(); throw; }
        ctx->side_end();
      } else {
        tall.insert(tall.end(), rest.begin(), rest.end());
        lde_batch(ctx, tall, log_blowup);
      }
    }
    auto extend_height = [&](size_t lde_height) {
      if (lde_height < top) ctx->side_join();
      std::vector<LdeJob> jobs;
      for (size_t i = 0; i < mats.size(); i++) {
        if (extended[i] || d->ldes[i].h != lde_height) continue;
        wait_ready(ctx->stream, mats[i]);
        jobs.push_back(job_of(i));
        extended[i] = 1;
      }
      lde_batch(ctx, jobs, log_blowup);
    };
    // every matrix on the device and extended (or queued) already: the shorter heights' rows are hashed in one launch (ZKM_ROWS_UP_FRONT=0:
    // inside compress_layer, as before round 5's last step); a matrix still crossing PCIe keeps the layer-by-layer order that waits for it late
    const bool rows_up_front = ctx->rows_up_front && std::all_of(extended.begin(), extended.end(), [](char e) { return e != 0; });
    build_tree(ctx, d->ldes, d->tree, extend_height, &col_flags, rows_up_front);
    ctx->side_join();
    d->cflags = cflags;          // stay with the commitment: open reads them (host_open.hpp: column tables)
    d->col_flags = col_flags;
    cflags = nullptr;
    for (size_t i = 0; i < mats.size(); i++)
      if (!extended[i]) throw std::runtime_error("pcs_commit: a matrix was not reached by the tree (internal error)");
    const uint32_t* h_root = d->tree.h_root ? d->tree.h_root : ctx->download_async(d->tree.node(d->tree.log_max, 0), 8);
    ctx->sync(ctx->stream);
    d->tree.h_root = nullptr;   // the pinned ring is rewound at the next top-level call
    memcpy(d->root, h_root, 32);
  } catch (...) {
    ctx->cur = ctx->stream;
    try { ctx->side_join(); } catch (...) {}   // nothing of this commit may still be running on the side stream when its buffers go back
    ctx->release(cflags);
    free_pcs_data(ctx, d);
    throw;
  }
  return d;
}

