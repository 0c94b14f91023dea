// Part of libzkm_hip.so's host side (one translation unit: csrc/zkm_hip.hip includes this file). Pcs::commit on the device: coset LDE dispatch, Merkle tree construction (mixed heights, fused leaf kernel, lane-parallel top), pcs_commit.
#pragma once
// ---- device helpers ------------------------------------------------------------------------------
static void lde_columns(zkm_ctx* ctx, const uint32_t* in, size_t n, size_t w, int bl, uint32_t lde_shift, uint32_t* out) {
  if (w == 0) return;
  int k = log2_strict(n);
  int lb = std::min(k, lde::LOG_ROW_MAX), la = k - lb;
  size_t N = n << bl;
  size_t B = (size_t)1 << lb;
  uint32_t w_n = kb::two_adic_generator(k), w_n_inv = kb::inv(w_n), w_N = kb::two_adic_generator(k + bl);
  uint32_t n_inv = kb::inv(kb::to_monty((uint32_t)(n % kb::P)));
  int nhi = B > 64 ? (int)(B >> 6) : 1;
  size_t rows_lds = (2 * (B + (B >> 5)) + 64 + nhi) * 4;
  const uint32_t* twf = lb > 0 ? ctx->twiddles(lb, false) : nullptr;
  const uint32_t* twi = lb > 0 ? ctx->twiddles(lb, true) : nullptr;
  if (la == 0) {
    KLAUNCH(ctx, "lde_rows", 4.0 * n * w + 4.0 * N * w, lde::lde_rows, dim3(1, (unsigned)w), dim3(lde::THREADS), rows_lds, in, out,
            lb, n, N, bl, lde_shift, w_N, n_inv, twf, twi);
    return;
  }
  size_t A = (size_t)1 << la;
  int logT = std::min(std::min(6, 14 - la), lb);  // la >= 1 implies lb = 13, so T >= 8
  size_t T = (size_t)1 << logT;
  size_t cols_lds = A * (T + 1) * 4;  // padded tile rows
  uint32_t* tmp1 = ctx->alloc_n<uint32_t>(n * w);
  uint32_t* tmp2 = ctx->alloc_n<uint32_t>((n * w) << bl);
  const uint32_t* twa_inv = ctx->twiddles(la, true);
  const uint32_t* twa_fwd = ctx->twiddles(la, false);
  KLAUNCH(ctx, "lde_cols_inverse", 8.0 * n * w, lde::lde_cols<false>, dim3((unsigned)(B / T), (unsigned)w, 1), dim3(lde::THREADS),
          cols_lds, in, tmp1, la, lb, logT, n, (size_t)0, n, bl, twa_inv);
  size_t big_lds = (B + (B >> 5) + 64 + nhi) * 4;
  KLAUNCH(ctx, "lde_rows", 4.0 * n * w + 4.0 * N * w, lde::lde_rows_big, dim3((unsigned)A, (unsigned)w), dim3(lde::THREADS), big_lds,
          (const uint32_t*)tmp1, tmp2, la, n, n, n * w, bl, lde_shift, w_n, w_n_inv, w_N, n_inv, twf, twi);
  KLAUNCH(ctx, "lde_cols_forward", 8.0 * N * w, lde::lde_cols<true>, dim3((unsigned)(B / T), (unsigned)w, 1u << bl),
          dim3(lde::THREADS), cols_lds, (const uint32_t*)tmp2, out, la, lb, logT, n, n * w, N, bl, twa_fwd);
  ctx->release(tmp1);
  ctx->release(tmp2);
}

// Upload an array of device pointers (one per column) and return the device copy.
static const uint32_t** upload_ptrs(zkm_ctx* ctx, const std::vector<const uint32_t*>& ptrs) {
  return (const uint32_t**)ctx->upload(ptrs.data(), ptrs.size() * sizeof(void*), nullptr);
}

// Layers of at most LANES_MAX nodes without injection: lane-parallel compression; returns true when it
// finished the tree (tail launch), false when the caller should go on with the next layer.
static bool compress_small_layer(zkm_ctx* ctx, Tree& t, int layer, size_t len) {
  const size_t LANES_MAX = 4096, TAIL = 64;
  if (len > LANES_MAX) {
    KLAUNCH(ctx, "compress_layer", 96.0 * len, merkle::compress_layer, dim3(div_up(len, merkle::THREADS)), dim3(merkle::THREADS), 0,
            (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8, len,
            (const uint32_t* const*)nullptr, 0);
    return false;
  }
  if (len <= TAIL) {
    KLAUNCH(ctx, "compress_tail", 96.0 * len, merkle::compress_tail_lanes, dim3(1), dim3(1024), 0, t.digests + t.layer_off[layer] * 8, len);
    return true;
  }
  KLAUNCH(ctx, "compress_small", 96.0 * len, merkle::compress_layer_lanes, dim3(div_up(len * 16, merkle::THREADS)), dim3(merkle::THREADS),
          0, (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8, len);
  return false;
}

// MerkleTreeMmcs::commit over column-major matrices of power-of-two heights.
static void build_tree(zkm_ctx* ctx, const std::vector<zkm_matrix>& mats, Tree& t,
                       const std::function<void(size_t)>& prepare_height = nullptr) {
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
    auto ptrs = cols_of_height(maxh);
    const uint32_t** d = upload_ptrs(ctx, ptrs);
    to_free.push_back(d);
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
  int layer = fuse;
  for (size_t len = maxh >> (fuse + 1); len >= 1; len >>= 1, layer++) {
    if (min_h > len) {
      if (compress_small_layer(ctx, t, layer, len)) break;
      continue;
    }
    auto ptrs = cols_of_height(len);
    const uint32_t** d = nullptr;
    if (!ptrs.empty()) { d = upload_ptrs(ctx, ptrs); to_free.push_back(d); wait_height(len); }
    KLAUNCH(ctx, "compress_layer", 96.0 * len + 4.0 * len * ptrs.size(), merkle::compress_layer, dim3(div_up(len, merkle::THREADS)),
            dim3(merkle::THREADS), 0, (const uint32_t*)(t.digests + t.layer_off[layer] * 8), t.digests + t.layer_off[layer + 1] * 8,
            len, (const uint32_t* const*)d, (int)ptrs.size());
    if (len == 1) break;
  }
  for (auto d : to_free) ctx->release((void*)d);
}

static void free_pcs_data(zkm_ctx* ctx, zkm_pcs_data* d) {
  if (!d) return;
  for (auto& m : d->ldes) ctx->release(m.d);
  for (auto& m : d->owned_evals) ctx->release(m.d);
  ctx->release(d->tree.digests);
  delete d;
}

// TwoAdicFriPcs::commit: LDE every matrix onto 3 * K (shift = GENERATOR / domain_shift), one tree.
static zkm_pcs_data* pcs_commit(zkm_ctx* ctx, const std::vector<zkm_matrix>& mats, const std::vector<uint32_t>& shifts,
                                int log_blowup) {
  zkm_pcs_data* d = new zkm_pcs_data();
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
    // Each height's matrices are extended right before the tree layer that reads them (see build_tree).
    std::vector<char> extended(mats.size(), 0);
    auto extend_height = [&](size_t lde_height) {
      for (size_t i = 0; i < mats.size(); i++) {
        if (extended[i] || d->ldes[i].h != lde_height) continue;
        wait_ready(ctx->stream, mats[i]);
        lde_columns(ctx, mats[i].d, mats[i].h, mats[i].w, log_blowup, kb::mul(kb::GEN, kb::inv(d->domain_shifts[i])), d->ldes[i].d);
        extended[i] = 1;
      }
    };
    build_tree(ctx, d->ldes, d->tree, extend_height);
    for (size_t i = 0; i < mats.size(); i++)
      if (!extended[i]) throw std::runtime_error("pcs_commit: a matrix was not reached by the tree (internal error)");
    const uint32_t* h_root = ctx->download_async(d->tree.node(d->tree.log_max, 0), 8);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(d->root, h_root, 32);
  } catch (...) {
    free_pcs_data(ctx, d);
    throw;
  }
  return d;
}

