// Part of libzkm_hip.so's host side (one translation unit: csrc/zkm_hip.hip includes this file). MachineProver::open (crates/stark/src/prover.rs:298-653): permutation traces, quotient, openings, FRI, queries, the proof stream.
#pragma once
static E4 host_pow2k(E4 a, int k) { return kb::epow2k(a, k); }

// ---- proof stream writer ---------------------------------------------------------------------------
// The proof stream goes straight into the caller's buffer (no growing vector and final copy: ~0.2 ms of host time per 1.2 MB proof
// during which the GPU had nothing to do). Words beyond the capacity are counted, not written, so that the length a too-small buffer
// would have needed can still be reported.
struct Writer {
  uint32_t* dst;
  size_t cap, n = 0;
  Writer(uint32_t* d, size_t c) : dst(d), cap(d ? c : 0) {}
  void u(uint32_t v) { if (n < cap) dst[n] = v; n++; }
  void words(const uint32_t* p, size_t cnt) {
    if (n < cap) memcpy(dst + n, p, std::min(cnt, cap - n) * 4);
    n += cnt;
  }
  void ext(const E4& e) { words(e.c, 4); }
  size_t size() const { return n; }
};

// ---- open --------------------------------------------------------------------------------------
struct RoundMat {
  const uint32_t* evals; size_t n; size_t width; uint32_t shift;
  const zkm_matrix* lde;
  int n_points;  // 1 or 2
  std::vector<E4> y[2];
  size_t col0 = 0;   // of this matrix in the opening's column tables (open::build_col_tables)
};
struct Round { const zkm_pcs_data* data; std::vector<RoundMat> mats; };

static const uint32_t SEPTIC_X[7] = {637514027, 1595065213, 1998064738, 72333738, 1211544370, 822986770, 1518535784};
static const uint32_t SEPTIC_Y[7] = {1604177449, 90440090, 259343427, 140470264, 1162099742, 941559812, 1064053343};

struct ChipMeta {
  const zkm_chip_desc* desc;
  int log_n;
  size_t n;
  int n_lookups, n_sends, perm_ext_w, max_values;
};

// Parse and validate a chip descriptor. The blobs are indices into device arrays: every column, register and
// table index is checked here so that a malformed descriptor is an error code, never an out-of-bounds access
// on the GPU.
static ChipMeta chip_meta(const zkm_chip_desc* d, size_t n, size_t n_public_values) {
  ChipMeta m;
  m.desc = d; m.n = n; m.log_n = log2_strict(n);
  m.n_lookups = m.n_sends = m.max_values = 0;
  const std::string who = std::string(" (chip ") + (d->name ? d->name : "?") + ")";
  if (d->log_quotient_degree > 3) throw std::runtime_error("log_quotient_degree > 3 unsupported" + who);
  if (d->lookups_len) {
    const uint32_t* w = d->lookups;
    const size_t len = d->lookups_len;
    size_t pos = 0;
    auto need = [&](size_t k) { if (pos + k > len) throw std::runtime_error("lookup blob truncated" + who); };
    need(2);
    uint32_t ns = w[pos++], nr = w[pos++];
    if ((uint64_t)ns + nr > 4096) throw std::runtime_error("too many lookups" + who);
    m.n_sends = ns; m.n_lookups = ns + nr;
    for (uint32_t i = 0; i < ns + nr; i++) {
      need(2);
      pos++;  // kind
      uint32_t nv = w[pos++];
      if (nv > 64) throw std::runtime_error("lookup with more than 64 values" + who);
      m.max_values = std::max<int>(m.max_values, nv);
      for (uint32_t v = 0; v <= nv; v++) {
        need(2);
        uint32_t nt = w[pos++];
        if (w[pos++] >= kb::P) throw std::runtime_error("lookup constant not a field element" + who);
        need(2 * (size_t)nt);
        for (uint32_t t = 0; t < nt; t++) {
          uint32_t cw = w[pos++], weight = w[pos++];
          uint32_t col = cw & 0x7fffffffu;
          if ((cw >> 31) ? col >= d->main_width : col >= d->prep_width) throw std::runtime_error("lookup column out of range" + who);
          if (weight >= kb::P) throw std::runtime_error("lookup weight not a field element" + who);
        }
      }
    }
    if (pos != len) throw std::runtime_error("lookup blob length mismatch" + who);
  }
  int batch = 1 << d->log_quotient_degree;
  m.perm_ext_w = m.n_lookups ? (m.n_lookups + batch - 1) / batch + 1 : 0;
  if (d->program_len) {
    if (d->program_len < 4 || d->program_len != 4 + 2 * (size_t)d->program[0]) throw std::runtime_error("program blob length mismatch" + who);
    if (d->program[2] != d->num_constraints) throw std::runtime_error("program constraint count mismatch" + who);
    const uint32_t ne = std::max<uint32_t>(d->program[1], 1), nb = std::max<uint32_t>(d->program[3], 1);
    if (ne > 256 || nb > 256) throw std::runtime_error("program register count out of range" + who);
    size_t asserts = 0;
    for (uint32_t k = 0; k < d->program[0]; k++) {
      uint32_t w0 = d->program[4 + 2 * k], imm = d->program[5 + 2 * k];
      uint32_t op = w0 & 0xff, dst = (w0 >> 8) & 0xff, ra = (w0 >> 16) & 0xff, rb = w0 >> 24;
      bool ok = true;
      switch (op) {
        case ZKM_OP_LD_MAIN: ok = dst < nb && ra < 2 && imm < d->main_width; break;
        case ZKM_OP_LD_PREP: ok = dst < nb && ra < 2 && imm < d->prep_width; break;
        case ZKM_OP_LD_PERM: ok = dst < ne && ra < 2 && imm < (uint32_t)m.perm_ext_w; break;
        case ZKM_OP_LD_CONST: ok = dst < nb && imm < kb::P; break;
        case ZKM_OP_LD_PV: ok = dst < nb && imm < n_public_values; break;
        case ZKM_OP_LD_CHALLENGE: ok = dst < ne && imm < 2; break;
        case ZKM_OP_LD_LOCAL_SUM: ok = dst < ne; break;
        case ZKM_OP_LD_GLOBAL_SUM: ok = dst < nb && imm < 14; break;
        case ZKM_OP_LD_IS_FIRST: case ZKM_OP_LD_IS_LAST: case ZKM_OP_LD_IS_TRANS: ok = dst < nb; break;
        case ZKM_OP_ADD_B: case ZKM_OP_SUB_B: case ZKM_OP_MUL_B: ok = dst < nb && ra < nb && rb < nb; break;
        case ZKM_OP_NEG_B: ok = dst < nb && ra < nb; break;
        case ZKM_OP_ADD_E: case ZKM_OP_SUB_E: case ZKM_OP_MUL_E: ok = dst < ne && ra < ne && rb < ne; break;
        case ZKM_OP_NEG_E: ok = dst < ne && ra < ne; break;
        case ZKM_OP_ADD_EB: case ZKM_OP_SUB_EB: case ZKM_OP_MUL_EB: ok = dst < ne && ra < ne && rb < nb; break;
        case ZKM_OP_ASSERT_B: ok = ra < nb; asserts++; break;
        case ZKM_OP_ASSERT_E: ok = ra < ne; asserts++; break;
        default: ok = false;
      }
      if (!ok) throw std::runtime_error("invalid instruction " + std::to_string(k) + " in constraint program" + who);
    }
    if (asserts != d->num_constraints) throw std::runtime_error("program asserts do not match num_constraints" + who);
  } else if (d->num_constraints) {
    throw std::runtime_error("num_constraints > 0 but no program" + who);
  }
  return m;
}

// One `open` call: the state that crosses Fiat-Shamir phases and one method per phase, in the order of prover.rs:298-653. Every device
// buffer allocated on the way and not yet handed to an owner is in `loose` and goes back to the pool when the object dies (also on an
// exception): a bad shard in a long-running farm must not leak HBM.
// One chip's permutation trace (crates/stark/src/permutation.rs:102-196) into `pt` (allocated, n x 4 perm_ext_w): perm_rows fills the batched
// reciprocal columns and the per-row sums, the three-phase scan turns the last extension column into the running sum. `salloc` hands out
// scratch that lives until the caller's stream has passed.
template <typename Alloc>
static void launch_permutation_trace(zkm_ctx* ctx, const ChipMeta& c, const uint32_t* d_blob, const uint32_t* trace, const uint32_t* prep, const E4& alpha,
                                     const E4* d_beta_powers, zkm_matrix& pt, Alloc&& salloc, std::vector<stark::ScanJob>& scans) {
  const double pbytes = 4.0 * c.n * (c.desc->main_width + c.desc->prep_width + pt.w);
  auto fit = c.desc->lookups_len ? ctx->perm_fns.find(perm_key(c.desc->lookups, c.desc->lookups_len, c.desc->log_quotient_degree)) : ctx->perm_fns.end();
  if (fit != ctx->perm_fns.end()) {
    // chip-specialised kernel (ziren_amd/codegen.py emit_perm_source): the blob walked at generation time, same values
    stark::PermArgs a;
    a.main = trace; a.prep = prep; a.n = c.n; a.alpha = alpha; a.beta_pows = d_beta_powers; a.perm = pt.d;
    size_t arg_size = sizeof(a);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size, HIP_LAUNCH_PARAM_END};
    ctx->flush_staged();
    const bool timed = ctx->kbegin("perm_rows", pbytes);
    HIP_CHECK(hipExtModuleLaunchKernel(fit->second, div_up(c.n, 256) * 256, 1, 1, 256, 1, 1, 0, ctx->cur, nullptr, config,
                                       timed ? ctx->krecs.back().start : nullptr, timed ? ctx->krecs.back().stop : nullptr, 0));
  } else {
    KLAUNCH(ctx, "perm_rows", pbytes, stark::perm_rows, dim3(div_up(c.n, stark::THREADS)), dim3(stark::THREADS), 0,
            d_blob, c.n_lookups, c.n_sends, 1 << c.desc->log_quotient_degree, trace, prep, c.n, alpha, d_beta_powers, pt.d, c.perm_ext_w);
  }
  stark::ScanJob j;
  j.data = pt.d + (size_t)(c.perm_ext_w - 1) * 4 * c.n;      // inclusive scan of the last ext column (4 base columns)
  j.n = c.n;
  j.nchunks = (c.n + stark::SCAN_BLOCK - 1) / stark::SCAN_BLOCK;
  j.totals = (uint32_t*)salloc(j.nchunks * 4 * 4);
  j.blk_end = (scans.empty() ? 0 : scans.back().blk_end) + (uint32_t)(j.nchunks * 4);
  j.pad = 0;
  scans.push_back(j);
}
// the three scan phases over every queued chip: three launches for the shard
static void launch_scans(zkm_ctx* ctx, const std::vector<stark::ScanJob>& scans, std::vector<void*>* scratch) {
  if (scans.empty()) return;
  const stark::ScanJob* d = (const stark::ScanJob*)ctx->upload_staged(scans.data(), scans.size() * sizeof(stark::ScanJob), scratch);
  double bytes = 0;
  bool multi = false;
  for (auto& j : scans) { bytes += 32.0 * j.n; multi |= j.nchunks > 1; }
  KLAUNCH(ctx, "scan", bytes, stark::scan_chunks, dim3(scans.back().blk_end), dim3(stark::THREADS), 0, d);
  if (multi) {
    KLAUNCH(ctx, "scan", 0.0, stark::scan_totals, dim3((unsigned)(4 * scans.size())), dim3(stark::THREADS), 0, d);
    KLAUNCH(ctx, "scan", bytes, stark::scan_add_offsets, dim3(scans.back().blk_end), dim3(stark::THREADS), 0, d);
  }
}

struct ShardOpening {
  struct Loose {
    zkm_ctx* ctx;
    std::vector<void*> v;
    void disown(void* p) { v.erase(std::remove(v.begin(), v.end(), p), v.end()); }
    ~Loose() { for (void* p : v) ctx->release(p); }
  };
  struct Guard { zkm_ctx* c; std::vector<zkm_pcs_data*> d; ~Guard() { for (auto p : d) free_pcs_data(c, p); } };

  zkm_ctx* ctx;
  const zkm_pk* pk;
  zkm_main_data* md;
  const zkm_fri_config* fri;
  uint32_t num_pv_elts;
  zkm_challenger* ch;
  hipStream_t st;
  const int bl;
  const size_t nc;
  std::vector<ChipMeta> chips;
  Loose loose;
  std::vector<void*>& scratch;
  Guard guard;
  // permutation phase
  E4 perm_ch[2];
  uint32_t* d_pv = nullptr;
  std::vector<zkm_matrix> perm_traces;
  std::vector<E4> local_sums;
  std::vector<std::array<uint32_t, 14>> global_sums;
  zkm_pcs_data* perm_data = nullptr;
  // quotient phase
  std::vector<zkm_matrix> qchunks;
  std::vector<uint32_t> qshifts;
  zkm_pcs_data* quot_data = nullptr;
  E4 zeta;
  // openings
  std::vector<Round> rounds;
  int log_max = 0;
  std::vector<E4*> ro = std::vector<E4*>(32, nullptr);
  const uint32_t** col_ptr_lde = nullptr;    // the opening's column tables (device; scratch of this opening)
  const uint32_t** col_ptr_eval = nullptr;
  uint32_t* col_mask = nullptr;
  // FRI
  std::vector<E4*> layers;      // f_t on device
  std::vector<Tree> ftrees;
  std::vector<std::array<uint32_t, 8>> commits;
  E4 final_poly;
  uint32_t pow_witness = 0;
  std::vector<size_t> indices;
  const uint32_t* gathered = nullptr;
  std::vector<uint32_t> gathered_big;

  void* salloc(size_t bytes) { void* p = ctx->alloc(bytes); scratch.push_back(p); return p; }

  ShardOpening(zkm_ctx* ctx_, const zkm_pk* pk_, zkm_main_data* md_, const zkm_chip_desc* chips_in, const zkm_fri_config* fri_, uint32_t num_pv_elts_,
               zkm_challenger* ch_)
      : ctx(ctx_), pk(pk_), md(md_), fri(fri_), num_pv_elts(num_pv_elts_), ch(ch_), st(ctx_->stream), bl((int)fri_->log_blowup), nc(md_->order.size()),
        loose{ctx_, {}}, scratch(loose.v), guard{ctx_, {}} {
  for (size_t i = 0; i < nc; i++) chips.push_back(chip_meta(&chips_in[md->order[i]], md->traces[i].h, md->public_values.size()));
  for (size_t i = 0; i < nc; i++) {
    if (chips[i].desc->main_width != md->traces[i].w) throw std::runtime_error("chip main_width does not match its trace");
    if ((int)chips[i].desc->log_quotient_degree > bl) throw std::runtime_error("log_quotient_degree > log_blowup unsupported");
    if (chips[i].desc->prep_index >= 0 && (!pk->data || (size_t)chips[i].desc->prep_index >= pk->prep.size()))
      throw std::runtime_error("chip references a preprocessed trace the proving key does not hold");
  }
  // FRI parameters as the reference's configurations use them (kb31_poseidon2.rs:203-241: blow-up 1..3 bits, 28..84 queries, 16 PoW
  // bits); anything outside a sane envelope is an error at the boundary, not an out-of-range shift further down
  if (bl < 1 || bl > 4) throw std::runtime_error("fri.log_blowup out of range (1..4)");
  if (fri->proof_of_work_bits > 30) throw std::runtime_error("fri.proof_of_work_bits out of range (0..30)");
  if (fri->num_queries < 1 || fri->num_queries > 1024) throw std::runtime_error("fri.num_queries out of range (1..1024)");
  if (num_pv_elts > md->public_values.size()) throw std::runtime_error("num_pv_elts exceeds public_values length");
  // --- transcript prelude (prover.rs:321-329)
  chal::observe_slice(ch, md->public_values.data(), num_pv_elts);
  chal::observe_slice(ch, md->data->root, 8);
  perm_ch[0] = chal::sample_ext(ch);
  perm_ch[1] = chal::sample_ext(ch);
  d_pv = (uint32_t*)ctx->upload_staged(md->public_values.data(), md->public_values.size() * 4, &scratch);
  }

  // --- permutation traces (prover.rs:337-365), their commitment, the cumulative sums into the transcript (:366-414)
  void permutation_phase() {
  perm_traces.assign(nc, zkm_matrix());
  local_sums.assign(nc, kb::ezero());
  global_sums.assign(nc, std::array<uint32_t, 14>());
  std::vector<uint32_t*> d_blobs(nc, nullptr);
  std::vector<stark::ScanJob> scans;
  std::vector<const uint32_t*> sum_src;
  uint32_t* h_sums = nullptr;
  {
    int maxv = 0;
    for (auto& c : chips) maxv = std::max(maxv, c.max_values);
    std::vector<E4> bp(maxv + 2);
    bp[0] = kb::eone();
    for (int i = 1; i < maxv + 2; i++) bp[i] = kb::emul(bp[i - 1], perm_ch[1]);
    E4* d_bp = (E4*)ctx->upload_staged(bp.data(), bp.size() * sizeof(E4), &scratch);
    // every chip's lookup table is staged before the first launch: one transfer for the whole phase
    for (size_t i = 0; i < nc; i++)
      if (chips[i].perm_ext_w > 0) d_blobs[i] = (uint32_t*)ctx->upload_staged(chips[i].desc->lookups, chips[i].desc->lookups_len * 4, &scratch);
    for (size_t i = 0; i < nc; i++) {
      const ChipMeta& c = chips[i];
      zkm_matrix& pt = perm_traces[i];
      pt.h = c.n; pt.w = (size_t)c.perm_ext_w * 4;
      pt.d = ctx->alloc_n<uint32_t>(std::max<size_t>(pt.h * pt.w, 1));
      scratch.push_back(pt.d);   // until the permutation commitment owns it
      if (c.perm_ext_w > 0) {
        const uint32_t* prep = c.desc->prep_index >= 0 ? pk->prep[c.desc->prep_index].d : nullptr;
        launch_permutation_trace(ctx, c, (const uint32_t*)d_blobs[i], (const uint32_t*)md->traces[i].d, prep, perm_ch[0], (const E4*)d_bp, pt,
                                 [&](size_t bytes) { return salloc(bytes); }, scans);
        uint32_t* last = pt.d + (size_t)(c.perm_ext_w - 1) * 4 * c.n;
        for (int e = 0; e < 4; e++) sum_src.push_back(last + (size_t)e * c.n + (c.n - 1));
      } else {
        for (int e = 0; e < 4; e++) sum_src.push_back(nullptr);
      }
      if (c.desc->commit_scope_global) {
        const zkm_matrix& m = md->traces[i];
        for (int k = 0; k < 14; k++) sum_src.push_back(m.d + (m.w - 14 + k) * m.h + (m.h - 1));
      } else {
        for (int k = 0; k < 14; k++) sum_src.push_back(nullptr);
      }
    }
    launch_scans(ctx, scans, &scratch);
    // one gather for every cumulative-sum word (18 per chip), read back after the commit's synchronisation
    const uint32_t* d_zero = (const uint32_t*)ctx->upload_staged("\0\0\0\0", 4, &scratch);
    for (auto& p : sum_src) if (!p) p = d_zero;
    const uint32_t** d_sum_src = (const uint32_t**)ctx->upload_staged(sum_src.data(), sum_src.size() * sizeof(void*), &scratch);
    uint32_t* d_sums = (uint32_t*)salloc(sum_src.size() * 4);
    KLAUNCH(ctx, "gather_words", 0.0, open::gather_words, dim3(div_up(sum_src.size(), open::THREADS)), dim3(open::THREADS), 0,
            (const uint32_t* const*)d_sum_src, sum_src.size(), d_sums);
    h_sums = ctx->download_async(d_sums, sum_src.size());
  }
  ctx->mark("permutation traces");
  perm_data = pcs_commit(ctx, perm_traces, {}, bl);  // synchronises: sums are on the host now
  guard.d.push_back(perm_data);
  perm_data->owned_evals = perm_traces;
  for (auto& m : perm_traces) loose.disown(m.d);
  ctx->mark("commit permutation");
  for (size_t i = 0; i < nc; i++) {
    for (int e = 0; e < 4; e++) local_sums[i].c[e] = h_sums[18 * i + e];
    if (chips[i].desc->commit_scope_global) {
      for (int k = 0; k < 14; k++) global_sums[i][k] = h_sums[18 * i + 4 + k];
    } else {
      for (int k = 0; k < 7; k++) { global_sums[i][k] = kb::to_monty(SEPTIC_X[k]); global_sums[i][7 + k] = kb::to_monty(SEPTIC_Y[k]); }
    }
  }
  chal::observe_slice(ch, perm_data->root, 8);
  for (size_t i = 0; i < nc; i++) {
    chal::observe_ext(ch, local_sums[i]);
    chal::observe_slice(ch, global_sums[i].data(), 14);
  }
  }

  // --- quotient values and their commitment (prover.rs:416-498); ends with zeta
  void quotient_phase() {
  E4 alpha = chal::sample_ext(ch);
  // every chip's tables (alpha powers, constants, program) are staged first and the kernels queued after the loop: one transfer for the phase
  std::vector<std::function<void()>> launches;
  for (size_t i = 0; i < nc; i++) {
    const ChipMeta& c = chips[i];
    const zkm_chip_desc* d = c.desc;
    int lqd = d->log_quotient_degree;
    int lq = c.log_n + lqd;
    size_t Q = (size_t)1 << lq;
    size_t nchunks = (size_t)1 << lqd;
    uint32_t* qbuf = ctx->alloc_n<uint32_t>(Q * 4);
    scratch.push_back(qbuf);   // until the quotient commitment owns it
    // alpha powers, reversed (prover.rs:453-456)
    size_t C = d->num_constraints;
    std::vector<E4> ap(std::max<size_t>(C, 1));
    E4 p = kb::eone();
    for (size_t k = 0; k < C; k++) { ap[C - 1 - k] = p; p = kb::emul(p, alpha); }
    E4* d_ap = (E4*)ctx->upload_staged(ap.data(), ap.size() * sizeof(E4), &scratch);
    uint32_t consts[32] = {0};
    for (int k = 0; k < 14; k++) consts[k] = global_sums[i][k];
    uint32_t w_q = kb::two_adic_generator(lq);
    // Z_H(3 w_Q^i) = 3^n * (w_Q^n)^i - 1 depends on i mod 2^lqd (zerofier_coset.rs:22-51)
    uint32_t s_pow_n = kb::pow(kb::GEN, (uint64_t)c.n);
    uint32_t wr = kb::two_adic_generator(lqd), wp = kb::ONE;
    for (size_t k = 0; k < nchunks; k++) {
      consts[16 + k] = kb::sub(kb::mul(s_pow_n, wp), kb::ONE);
      consts[24 + k] = kb::inv(consts[16 + k]);
      wp = kb::mul(wp, wr);
    }
    uint32_t* d_consts = (uint32_t*)ctx->upload_staged(consts, sizeof consts, &scratch);
    static const uint32_t empty_prog[4] = {0, 1, 0, 1};
    uint32_t* d_prog = (uint32_t*)ctx->upload_staged(d->program_len ? d->program : empty_prog, std::max<size_t>(d->program_len, 4) * 4, &scratch);
    stark::QuotientArgs a;
    a.program = d_prog + 4;
    a.n_instr = d->program_len ? d->program[0] : 0;
    a.n_regs = d->program_len ? d->program[1] : 1;
    a.main_lde = md->data->ldes[i].d; a.main_stride = md->data->ldes[i].h;
    a.prep_lde = d->prep_index >= 0 ? pk->data->ldes[d->prep_index].d : nullptr;
    a.prep_stride = d->prep_index >= 0 ? pk->data->ldes[d->prep_index].h : 0;
    a.perm_lde = perm_data->ldes[i].d; a.perm_stride = perm_data->ldes[i].h;
    a.log_n = c.log_n; a.lqd = lqd;
    a.alpha_pows = d_ap; a.public_values = d_pv;
    a.perm_alpha = perm_ch[0]; a.perm_beta = perm_ch[1];
    a.local_sum = local_sums[i];
    a.consts = d_consts;
    a.w_q = w_q; a.g_inv = kb::inv(kb::two_adic_generator(c.log_n));
    a.out = qbuf;
    a.selectors = ctx->selectors(c.log_n, lqd, w_q, a.g_inv, d_consts);
    a.n_base_regs = d->program_len ? std::max<uint32_t>(d->program[3], 1) : 1;
    size_t per_thread = (size_t)a.n_regs * 16 + (size_t)a.n_base_regs * 4;
    int bd = 256;
    while (per_thread * bd > 64 * 1024 && bd > 64) bd >>= 1;
    size_t lds = per_thread * bd;
    if (lds > 152 * 1024) throw std::runtime_error(std::string("constraint program of chip ") + d->name + " needs too many registers");
    double qbytes = 4.0 * Q * (d->main_width + d->prep_width + 4.0 * c.perm_ext_w) + 16.0 * Q;
    auto fit = d->program_len ? ctx->quotient_fns.find(fnv1a(d->program, d->program_len)) : ctx->quotient_fns.end();
    const bool special = fit != ctx->quotient_fns.end();
    const std::vector<hipFunction_t>* fns = special ? &fit->second : nullptr;
    hipFunction_t uni_fn = nullptr;
    a.uniforms = nullptr;
    if (special) {
      auto uit = ctx->quotient_uniform_fns.find(fit->first);
      if (uit != ctx->quotient_uniform_fns.end()) {
        uni_fn = uit->second;
        a.uniforms = ctx->alloc_n<uint32_t>(stark::QUOTIENT_UNIFORM_WORDS);   // one table per chip: the kernels of a phase may follow each other closely
        scratch.push_back(a.uniforms);
      }
    }
    launches.push_back([this, a, fns, uni_fn, Q, bd, lds, qbytes]() mutable {
      if (fns) {
        // chip-specialised kernel: same arithmetic, values in VGPRs
        size_t arg_size = sizeof(a);
        void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size, HIP_LAUNCH_PARAM_END};
        ctx->flush_staged();
        if (uni_fn)    // one wavefront computes the chip's wave-uniform values into a.uniforms; the kernel behind it on the stream reads them
          HIP_CHECK(hipExtModuleLaunchKernel(uni_fn, 64, 1, 1, 64, 1, 1, 0, st, nullptr, config, nullptr, nullptr, 0));
        for (hipFunction_t fn : *fns) {       // several for a long program: the first stores, the others accumulate
          const bool timed = ctx->kbegin("quotient", qbytes);
          HIP_CHECK(hipExtModuleLaunchKernel(fn, div_up(Q, 256) * 256, 1, 1, 256, 1, 1, 0, st, nullptr, config,
                                             timed ? ctx->krecs.back().start : nullptr, timed ? ctx->krecs.back().stop : nullptr, 0));
        }
      } else {
        KLAUNCH(ctx, "quotient", qbytes, stark::quotient_kernel, dim3(div_up(Q, bd)), dim3(bd), lds, a);
      }
    });
    uint32_t wqp = kb::ONE;
    for (size_t k = 0; k < nchunks; k++) {
      zkm_matrix m; m.h = c.n; m.w = 4; m.d = qbuf + k * 4 * c.n; m.owned = (k == 0);
      qchunks.push_back(m);
      qshifts.push_back(kb::mul(kb::GEN, wqp));
      wqp = kb::mul(wqp, w_q);
    }
    // the first (tallest) chip's kernel goes out as soon as its own tables are staged: the host prepares the other chips' tables
    // (alpha powers, zerofier inverses: ~0.15 ms for a core shard) while it runs, instead of in front of an idle GPU
    if (i == 0) { launches[0](); launches[0] = [] {}; }
  }
  for (auto& l : launches) l();
  ctx->mark("quotient values");
  quot_data = pcs_commit(ctx, qchunks, qshifts, bl);
  for (auto& m : qchunks) if (m.owned) { quot_data->owned_evals.push_back(m); loose.disown(m.d); }
  guard.d.push_back(quot_data);
  ctx->mark("commit quotient");
  chal::observe_slice(ch, quot_data->root, 8);
  zeta = chal::sample_ext(ch);
  }

  // --- opening rounds (prover.rs:503-556): preprocessed, main, permutation, quotient; every column evaluated at zeta (and zeta g)
  void opened_values() {
  if (pk->data) {
    Round r; r.data = pk->data;
    for (size_t j = 0; j < pk->prep.size(); j++)
      r.mats.push_back(RoundMat{pk->prep[j].d, pk->prep[j].h, pk->prep[j].w, kb::ONE, &pk->data->ldes[j], pk->local_only[j] ? 1 : 2, {}});
    rounds.push_back(r);
  }
  {
    Round r; r.data = md->data;
    for (size_t i = 0; i < nc; i++)
      r.mats.push_back(RoundMat{md->traces[i].d, md->traces[i].h, md->traces[i].w, kb::ONE, &md->data->ldes[i], chips[i].desc->local_only ? 1 : 2, {}});
    rounds.push_back(r);
    Round rp; rp.data = perm_data;
    for (size_t i = 0; i < nc; i++)
      rp.mats.push_back(RoundMat{perm_traces[i].d, perm_traces[i].h, perm_traces[i].w, kb::ONE, &perm_data->ldes[i], 2, {}});
    rounds.push_back(rp);
    Round rq; rq.data = quot_data;
    for (size_t i = 0; i < qchunks.size(); i++)
      rq.mats.push_back(RoundMat{qchunks[i].d, qchunks[i].h, 4, qshifts[i], &quot_data->ldes[i], 1, {}});
    rounds.push_back(rq);
  }
  // the column tables of this opening: per column of every matrix a pointer and a row mask — the column itself, or (mask 0) eight copies of
  // the word of a column that never changes, from the flags its commitment kept (open.cuh: build_col_tables)
  {
    std::vector<open::ColJob> cjobs;
    size_t total = 0;
    for (auto& r : rounds)
      for (auto& m : r.mats) {
        m.col0 = total;
        if (m.width == 0) continue;
        const size_t idx = (size_t)(m.lde - r.data->ldes.data());
        const uint32_t* cf = idx < r.data->col_flags.size() ? r.data->col_flags[idx] : nullptr;
        cjobs.push_back(open::ColJob{m.lde->d, m.evals, cf, m.lde->h, m.n, (uint32_t)total, (uint32_t)m.width});
        total += m.width;
      }
    const size_t tn = std::max<size_t>(total, 1);
    col_ptr_lde = (const uint32_t**)salloc(tn * sizeof(void*));
    col_ptr_eval = (const uint32_t**)salloc(tn * sizeof(void*));
    col_mask = (uint32_t*)salloc(tn * sizeof(uint32_t));
    uint32_t* cword = (uint32_t*)salloc(tn * 8 * sizeof(uint32_t));
    if (total) {
      const open::ColJob* d_c = (const open::ColJob*)ctx->upload_staged(cjobs.data(), cjobs.size() * sizeof(open::ColJob), &scratch);
      KLAUNCH(ctx, "col_tables", 0.0, open::build_col_tables, dim3(div_up(total, 256)), dim3(256), 0, d_c, (int)cjobs.size(), (uint32_t)total, col_ptr_lde,
              col_ptr_eval, col_mask, cword);
    }
  }
  // (i) evaluate every column at zeta (and zeta * g): barycentric weights shared per (height, shift)
  {
    std::map<std::pair<size_t, uint32_t>, E4*> wcache;
    size_t total_y = 0;
    for (auto& r : rounds) for (auto& m : r.mats) total_y += m.width * 2;
    E4* d_y = (E4*)salloc(std::max<size_t>(total_y, 1) * sizeof(E4));
    // one launch each for all weight vectors, all column evaluations and all partial sums (job tables: open.cuh)
    std::vector<open::WeightJob> wjobs;
    std::vector<open::EvalJob> ejobs;
    std::vector<open::SumJob> sjobs;
    uint32_t wblk = 0, eblk = 0, sblk = 0;
    double wbytes = 0, ebytes = 0;
    size_t ypos = 0;
    for (auto& r : rounds)
      for (auto& m : r.mats) {
        if (m.width == 0) continue;
        auto key = std::make_pair(m.n, m.shift);
        E4* wts;
        auto it = wcache.find(key);
        if (it == wcache.end()) {
          int ln = log2_strict(m.n);
          open::WeightJob wj;
          wj.u = kb::escale(zeta, kb::inv(m.shift));
          wj.c = kb::escale(kb::esub_base(host_pow2k(wj.u, ln), kb::ONE), kb::inv(kb::to_monty((uint32_t)(m.n % kb::P))));
          wts = (E4*)salloc(m.n * sizeof(E4));
          wj.w_n = kb::two_adic_generator(ln); wj.n = m.n; wj.out = wts;
          wblk += div_up(div_up(m.n, 4), open::THREADS);
          wj.blk_end = wblk;
          wjobs.push_back(wj);
          wbytes += 16.0 * m.n;
          wcache[key] = wts;
        } else wts = it->second;
        open::EvalJob ej;
        ej.mat = m.evals; ej.weights = wts; ej.n = m.n; ej.width = (int)m.width;
        ej.colptr = col_ptr_eval + m.col0; ej.colmask = col_mask + m.col0;
        ej.groups = (int)div_up(m.width, open::EVAL_COLS);
        ej.split = 1;
        if (m.n >= 4 * open::THREADS) {
          ej.split = (int)std::min<size_t>(std::max<size_t>(1, 3072 / ej.groups), m.n / (4 * open::THREADS));
          ej.kind = m.n_points > 1 ? 1 : 0;
        } else {
          ej.kind = m.n_points > 1 ? 3 : 2;
        }
        ej.lead = 0;
        if (ej.split >= 8 && ej.groups > 1) {
          // row slices dealt to XCDs whole (open.cuh, EVAL_XCD_AWARE): a multiple of 8 slices, the job's first block on a multiple of 8
          ej.split &= ~7;
          const uint32_t lead = (8 - eblk % 8) % 8;
          ej.lead = open::EVAL_XCD_AWARE | lead;
          eblk += lead;
        }
        ej.partials = (E4*)salloc((size_t)ej.split * m.width * 2 * sizeof(E4));
        eblk += (uint32_t)ej.groups * (uint32_t)ej.split;
        ej.blk_end = eblk;
        ejobs.push_back(ej);
        ebytes += 4.0 * m.n * m.width + 16.0 * m.n;
        sblk += (uint32_t)(m.width * 2);
        sjobs.push_back(open::SumJob{ej.partials, ej.split, (int)(m.width * 2), (uint32_t)ypos, sblk});
        ypos += m.width * 2;
      }
    if (!ejobs.empty()) {
      const open::WeightJob* d_w = (const open::WeightJob*)ctx->upload_staged(wjobs.data(), wjobs.size() * sizeof(open::WeightJob), &scratch);
      const open::EvalJob* d_e = (const open::EvalJob*)ctx->upload_staged(ejobs.data(), ejobs.size() * sizeof(open::EvalJob), &scratch);
      const open::SumJob* d_s = (const open::SumJob*)ctx->upload_staged(sjobs.data(), sjobs.size() * sizeof(open::SumJob), &scratch);
      KLAUNCH(ctx, "bary_weights", wbytes, open::bary_weights_batch, dim3(wblk), dim3(open::THREADS), 0, d_w);
      KLAUNCH(ctx, "eval_columns", ebytes, open::eval_columns_batch, dim3(eblk), dim3(open::THREADS), 0, d_e);
      KLAUNCH(ctx, "reduce_partials", 0.0, open::reduce_partials_batch, dim3(sblk), dim3(64), 0, d_s, d_y);
    }
    const E4* hy = ctx->download_async(d_y, std::max<size_t>(total_y, 1));
    ctx->sync(st);
    ypos = 0;
    for (auto& r : rounds)
      for (auto& m : r.mats) {
        for (int pt = 0; pt < 2; pt++) m.y[pt].clear();
        if (m.width == 0) continue;
        for (int pt = 0; pt < m.n_points; pt++) {
          m.y[pt].resize(m.width);
          for (size_t c = 0; c < m.width; c++) m.y[pt][c] = hy[ypos + c * 2 + pt];
        }
        ypos += m.width * 2;
      }
  }
  ctx->mark("open: evaluations");
  }

  // (ii) alpha — opened values are not observed (fri.rs:78) — and (iii) the reduced openings per LDE height (fri.rs:103-204)
  void reduced_openings() {
  E4 fa = chal::sample_ext(ch);
  log_max = 0;
  // one table of alpha powers long enough for the largest per-height column count (a matrix's powers are a window of it)
  std::vector<size_t> count(32, 0);
  size_t max_count = 1;
  for (auto& r : rounds) for (auto& m : r.mats) {
    int lh = log2_strict(m.lde->h);
    log_max = std::max(log_max, lh);
    count[lh] += m.width * m.n_points;
    max_count = std::max(max_count, count[lh]);
  }
  std::vector<E4> fap(max_count + 1);
  fap[0] = kb::eone();
  for (size_t i = 1; i <= max_count; i++) fap[i] = kb::emul(fap[i - 1], fa);
  E4* d_fap = (E4*)ctx->upload_staged(fap.data(), fap.size() * sizeof(E4), &scratch);
  {
    std::vector<std::vector<open::ReduceMat>> per_h(32);
    std::vector<size_t> off(32, 0);                     // columns counted so far per height (fri.rs: per (point, column))
    std::vector<std::array<E4, 2>> Y(32, {kb::ezero(), kb::ezero()});
    for (auto& r : rounds)
      for (auto& m : r.mats) {
        int lh = log2_strict(m.lde->h);
        open::ReduceMat rm;
        rm.lde = m.lde->d; rm.width = (int)m.width; rm.n_points = m.n_points;
        rm.colptr = col_ptr_lde + m.col0; rm.colmask = col_mask + m.col0;
        rm.apow_off = (uint32_t)off[lh]; rm.pad = 0;
        rm.A1 = fap[m.width];
        for (int pt = 0; pt < m.n_points; pt++) {
          E4 ysum = kb::ezero();
          for (size_t c = 0; c < m.width; c++) ysum = kb::eadd(ysum, kb::emul(fap[off[lh] + c], m.y[pt][c]));
          Y[lh][pt] = kb::eadd(Y[lh][pt], ysum);
          off[lh] += m.width;
        }
        per_h[lh].push_back(rm);
      }
    std::vector<open::ReduceMat*> d_rms(32, nullptr);
    for (int lh = 0; lh < 32; lh++)
      if (!per_h[lh].empty()) d_rms[lh] = (open::ReduceMat*)ctx->upload_staged(per_h[lh].data(), per_h[lh].size() * sizeof(open::ReduceMat), &scratch);
    for (int lh = 0; lh < 32; lh++) {
      if (per_h[lh].empty()) continue;
      size_t N = (size_t)1 << lh;
      ro[lh] = (E4*)salloc(N * sizeof(E4));
      open::ReduceMat* d_rm = d_rms[lh];
      E4 z1 = kb::escale(zeta, kb::two_adic_generator(lh - bl));
      double rbytes = 16.0 * N;
      for (auto& rm : per_h[lh]) rbytes += 4.0 * N * rm.width;
      std::pair<const uint32_t*, const uint32_t*> pw{nullptr, nullptr};
      if (lh >= 12) pw = ctx->pow_tables(lh);
      KLAUNCH(ctx, "reduce_openings", rbytes, open::reduce_openings, dim3(div_up(N, open::THREADS)), dim3(open::THREADS), 0,
              (const open::ReduceMat*)d_rm, (int)per_h[lh].size(), lh, (const E4*)d_fap, Y[lh][0], Y[lh][1], zeta, z1, kb::two_adic_generator(lh),
              pw.first, pw.second, ro[lh]);
    }
  }
  ctx->mark("open: reduced openings");
  }

  // (iv) FRI commit phase (fri.rs:257-358): per layer a tree over the pairs, its root into the transcript, beta out, fold
  // fri.rs:34-69. Per layer: commit to the pairs (leaf kernel + tree levels), observe the root, sample beta, fold (+ beta^2 ro of the new
  // length). The transcript step runs ON THE DEVICE, in the launch that finishes the layer's tree (merkle::observe_root_sample_beta over a
  // device copy of the challenger), and the fold reads beta from memory: all layers are queued back to back with no host round trip
  // between them (before: one per layer, 22 for the benchmarked shard — most of a small shard's commit phase was waiting). Afterwards the
  // host replays the same steps from the roots the launches left in page-locked memory — it remains the source of truth: the betas the
  // device used must be the ones the host transcript samples, or the proof is refused here.
  void fri_commit_phase() {
  static_assert(sizeof(merkle::DevChallenger) == sizeof(zkm_challenger), "the device challenger is the ABI's challenger");
  E4* f = ro[log_max];
  int lf = log_max;
  uint32_t neg_half = kb::neg(kb::inv(kb::to_monty(2)));
  const int n_layers = log_max - bl;
  merkle::DevChallenger* d_ch = (merkle::DevChallenger*)ctx->upload(ch, sizeof(zkm_challenger), &scratch);    // its own buffer: the launches write it
  E4* d_betas = (E4*)salloc((size_t)std::max(n_layers, 1) * 2 * sizeof(E4));
  uint32_t* h_roots = (uint32_t*)ctx->pin_alloc((size_t)std::max(n_layers, 1) * 32);
  if (!h_roots) throw std::runtime_error("pinned staging ring exhausted");
  for (int i = 0; i < n_layers * 8; i++) ((volatile uint32_t*)h_roots)[i] = 0xffffffffu;
  int li = 0;
  while (lf > bl) {
    size_t len = (size_t)1 << lf, half = len / 2;
    Tree t;
    t.max_height = half; t.log_max = lf - 1;
    size_t off = 0;
    for (size_t l = half; l >= 1; l >>= 1) { t.layer_off.push_back(off); off += l; if (l == 1) break; }
    t.digests = (uint32_t*)salloc(off * 8 * 4);
    int fuse = 0;   // FRI trees have one matrix: the first levels are reduced inside the leaf kernel's blocks
    static const size_t FRI_LANES_MAX = getenv("ZKM_FRI_LANES_MAX") ? (size_t)atol(getenv("ZKM_FRI_LANES_MAX")) : 4096;     // A/B knob
    if (half > FRI_LANES_MAX && half >= (size_t)merkle::FRI_FUSE_LEAVES) fuse = std::min(merkle::FRI_FUSE_MAX_LEVELS, lf - 1);
    if (half <= FRI_LANES_MAX)      // a latency-bound layer: sixteen lanes per leaf, then the lane-parallel levels (compress_small_layer)
      KLAUNCH(ctx, "hash_fri_leaves_lanes", 64.0 * half, merkle::hash_fri_leaves_lanes, dim3(div_up(half * 16, merkle::THREADS)), dim3(merkle::THREADS), 0,
              (const E4*)f, half, t.digests);
    else if (fuse > 0)
      KLAUNCH(ctx, "hash_fri_leaves_tree", 32.0 * half + 32.0 * half * (2.0 - 1.0 / (1 << fuse)), merkle::hash_fri_leaves_tree,
              dim3(half / merkle::FRI_FUSE_LEAVES), dim3(merkle::FRI_FUSE_LEAVES), merkle::FRI_FUSE_LEAVES * 12 * sizeof(uint32_t), (const E4*)f, half, t.digests, fuse);
    else
      KLAUNCH(ctx, "hash_fri_leaves", 64.0 * half, merkle::hash_fri_leaves, dim3(div_up(half, merkle::THREADS)), dim3(merkle::THREADS), 0,
              (const E4*)f, half, t.digests);
    int layer = fuse;
    bool tail = false;
    for (size_t l = half >> (fuse + 1); l >= 1; l >>= 1, layer++)
      if (compress_small_layer(ctx, t, layer, l, d_ch, d_betas + 2 * li, h_roots + 8 * li)) { tail = true; break; }
    if (!tail)      // a tree of one leaf (or one whose levels all ran inside the leaf kernel): the root is in the tree already
      KLAUNCH(ctx, "fri_root_challenge", 0.0, merkle::fri_root_challenge, dim3(1), dim3(64), 0, t.node(t.log_max, 0), h_roots + 8 * li, d_ch, d_betas + 2 * li);
    t.h_root = nullptr;
    E4* g = (E4*)salloc(half * sizeof(E4));
    KLAUNCH(ctx, "fri_fold", 48.0 * half + (ro[lf - 1] ? 16.0 * half : 0.0), open::fri_fold, dim3(div_up(half, open::THREADS)),
            dim3(open::THREADS), 0, (const E4*)f, lf, (const E4*)(d_betas + 2 * li), kb::two_adic_generator(lf),
            kb::inv(kb::two_adic_generator(lf)), neg_half, (const E4*)ro[lf - 1], g);
    layers.push_back(f);
    ftrees.push_back(t);
    f = g;
    lf--;
    li++;
  }
  const size_t nfin = (size_t)1 << lf;
  const E4* fin = ctx->download_async((const E4*)f, nfin);
  const E4* used = n_layers > 0 ? ctx->download_async((const E4*)d_betas, (size_t)n_layers * 2) : nullptr;
  ctx->sync(st);
  // the host's transcript, from the roots: observe, sample — and hold the device's betas against it
  for (int i = 0; i < n_layers; i++) {
    std::array<uint32_t, 8> root;
    memcpy(root.data(), h_roots + 8 * i, 32);
    for (int k = 0; k < 8; k++)
      if (root[k] == 0xffffffffu) throw std::runtime_error("an FRI layer's root did not arrive in page-locked memory (internal error)");
    chal::observe_slice(ch, root.data(), 8);
    commits.push_back(root);
    const E4 beta = chal::sample_ext(ch);
    if (!kb::eq(beta, used[2 * i])) throw std::runtime_error("the device transcript of the FRI commit phase diverged from the host's (internal error)");
  }
  for (size_t i = 1; i < nfin; i++)
    if (!kb::eq(fin[i], fin[0])) throw std::runtime_error("FRI final polynomial is not constant (internal error)");
  final_poly = fin[0];
  chal::observe_ext(ch, final_poly);
  ctx->mark("open: FRI commit phase");
  }

  // proof of work: the smallest canonical witness
  void grind() {
  {
    uint32_t* d_state = (uint32_t*)ctx->upload_staged(ch->sponge_state, 64, &scratch);
    uint32_t* d_in = (uint32_t*)ctx->upload_staged(ch->input_buffer, 64, &scratch);
    unsigned int* d_best = (unsigned int*)salloc(4);
    uint32_t base = 0, found = 0xffffffffu;
    const uint32_t BATCH = 1u << 20;
    while (base < kb::P) {
      HIP_CHECK(hipMemsetAsync(d_best, 0xff, 4, st));
      uint32_t total = std::min<uint64_t>(BATCH, (uint64_t)kb::P - base);
      KLAUNCH(ctx, "grind", 0.0, merkle::grind, dim3(div_up(total, merkle::THREADS)), dim3(merkle::THREADS), 0, (const uint32_t*)d_state,
              (const uint32_t*)d_in, (int)ch->num_inputs, (int)fri->proof_of_work_bits, base, total, d_best);
      const unsigned int* h_best = ctx->download_async((const unsigned int*)d_best, 1);
      ctx->sync(st);
      found = *h_best;
      if (found != 0xffffffffu) break;
      base += total;
    }
    if (found == 0xffffffffu) throw std::runtime_error("proof-of-work search exhausted the field");
    pow_witness = kb::to_monty(found);
    chal::observe(ch, pow_witness);
    if (chal::sample_bits(ch, fri->proof_of_work_bits) != 0) throw std::runtime_error("proof-of-work witness rejected by host transcript");
  }
  ctx->mark("open: grind");
  }

  // query indices out of the transcript; every word the queries open is fetched by one gather in serialisation order
  void query_phase() {
  indices.assign(fri->num_queries, 0);
  for (auto& q : indices) q = chal::sample_bits(ch, log_max);
  // one template entry per word of a query, in serialisation order (open::QueryWord): the positions depend on the index only through shifts
  std::vector<open::QueryWord> tmpl;
  for (auto& r : rounds) {
    const Tree& t = r.data->tree;
    const uint32_t tree_shift = (uint32_t)(log_max - t.log_max);
    for (auto& m : r.mats) {
      const uint32_t sh = tree_shift + (uint32_t)(t.log_max - log2_strict(m.lde->h));
      for (size_t c = 0; c < m.width; c++) tmpl.push_back(open::QueryWord{m.lde->d + c * m.lde->h, sh, 0, 1, 0});
    }
    for (int l = 0; l < t.log_max; l++)
      for (uint32_t k = 0; k < 8; k++) tmpl.push_back(open::QueryWord{t.node(l, 0), tree_shift + (uint32_t)l, 1, 8, k});
  }
  for (size_t tI = 0; tI < ftrees.size(); tI++) {
    for (uint32_t k = 0; k < 4; k++) tmpl.push_back(open::QueryWord{(const uint32_t*)layers[tI], (uint32_t)tI, 1, 4, k});
    const Tree& t = ftrees[tI];
    for (int l = 0; l < t.log_max; l++)
      for (uint32_t k = 0; k < 8; k++) tmpl.push_back(open::QueryWord{t.node(l, 0), (uint32_t)(tI + 1 + l), 1, 8, k});
  }
  const size_t per_query = tmpl.size(), n_gather = per_query * indices.size();
  if (n_gather) {
    std::vector<uint32_t> idx32(indices.begin(), indices.end());
    const open::QueryWord* d_tmpl = (const open::QueryWord*)ctx->upload_staged(tmpl.data(), tmpl.size() * sizeof(open::QueryWord), &scratch);
    const uint32_t* d_idx = (const uint32_t*)ctx->upload_staged(idx32.data(), idx32.size() * 4, &scratch);
    uint32_t* d_dst = (uint32_t*)salloc(n_gather * 4);
    ctx->flush_staged();
    hipLaunchKernelGGL(open::gather_queries, dim3(div_up(n_gather, open::THREADS)), dim3(open::THREADS), 0, st, d_tmpl, per_query, d_idx, indices.size(), d_dst);
    LAUNCH_CHECK();
    uint32_t* h = (uint32_t*)ctx->pin_alloc(n_gather * 4);      // through the pinned ring when it fits: no staged pageable copy
    if (h) {
      HIP_CHECK(hipMemcpyAsync(h, d_dst, n_gather * 4, hipMemcpyDeviceToHost, st));
      gathered = h;
    } else {
      gathered_big.resize(n_gather);
      HIP_CHECK(hipMemcpyAsync(gathered_big.data(), d_dst, n_gather * 4, hipMemcpyDeviceToHost, st));
      gathered = gathered_big.data();
    }
    ctx->sync(st);
  }
  ctx->mark("open: queries");
  }

  // --- serialise (INTEGRATION.md "ShardProof stream"; prover.rs:558-652)
  void serialise(Writer& out) {
  out.words(md->data->root, 8);
  out.words(perm_data->root, 8);
  out.words(quot_data->root, 8);
  out.u((uint32_t)nc);
  size_t ri = pk->data ? 1 : 0;
  Round& rmain = rounds[ri];
  Round& rperm = rounds[ri + 1];
  Round& rquot = rounds[ri + 2];
  size_t qpos = 0;
  auto put_exts = [&](const std::vector<E4>& v) { for (auto& e : v) out.ext(e); };
  for (size_t i = 0; i < nc; i++) {
    const zkm_chip_desc* d = chips[i].desc;
    out.u((uint32_t)md->order[i]);
    out.u((uint32_t)chips[i].log_n);
    if (d->prep_index >= 0) {
      RoundMat& pm = rounds[0].mats[d->prep_index];
      out.u((uint32_t)pm.width);
      put_exts(pm.y[0]);
      if (pm.n_points > 1) put_exts(pm.y[1]); else put_exts(std::vector<E4>(pm.width, kb::ezero()));
    } else out.u(0);
    RoundMat& mm = rmain.mats[i];
    out.u((uint32_t)mm.width);
    put_exts(mm.y[0]);
    if (mm.n_points > 1) put_exts(mm.y[1]); else put_exts(std::vector<E4>(mm.width, kb::ezero()));
    RoundMat& pm = rperm.mats[i];
    out.u((uint32_t)pm.width);
    put_exts(pm.y[0]); put_exts(pm.y[1]);
    size_t nch = (size_t)1 << d->log_quotient_degree;
    out.u((uint32_t)nch);
    for (size_t k = 0; k < nch; k++) put_exts(rquot.mats[qpos++].y[0]);
    out.words(global_sums[i].data(), 14);
    out.ext(local_sums[i]);
  }
  out.u((uint32_t)commits.size());
  for (auto& c : commits) out.words(c.data(), 8);
  out.u((uint32_t)indices.size());
  size_t gp = 0;
  for (size_t qi = 0; qi < indices.size(); qi++) {
    out.u((uint32_t)rounds.size());
    for (auto& r : rounds) {
      out.u((uint32_t)r.mats.size());
      for (auto& m : r.mats) { out.u((uint32_t)m.width); out.words(gathered + gp, m.width); gp += m.width; }
      out.u((uint32_t)r.data->tree.log_max);
      out.words(gathered + gp, (size_t)r.data->tree.log_max * 8); gp += (size_t)r.data->tree.log_max * 8;
    }
    out.u((uint32_t)ftrees.size());
    for (auto& t : ftrees) {
      out.words(gathered + gp, 4); gp += 4;
      out.u((uint32_t)t.log_max);
      out.words(gathered + gp, (size_t)t.log_max * 8); gp += (size_t)t.log_max * 8;
    }
  }
  out.ext(final_poly);
  out.u(pow_witness);
  out.u((uint32_t)md->public_values.size());
  out.words(md->public_values.data(), md->public_values.size());
  }
};

static void open_impl(zkm_ctx* ctx, const zkm_pk* pk, zkm_main_data* md, const zkm_chip_desc* chips_in, const zkm_fri_config* fri,
                      uint32_t num_pv_elts, zkm_challenger* ch, Writer& out) {
  ShardOpening o(ctx, pk, md, chips_in, fri, num_pv_elts, ch);
  o.permutation_phase();
  o.quotient_phase();
  o.opened_values();
  o.reduced_openings();
  o.fri_commit_phase();
  o.grind();
  o.query_phase();
  o.serialise(out);
}

