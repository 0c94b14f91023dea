// TEST INFRASTRUCTURE — C entry points of the CPU oracle (liboracle.so), loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only. The product path
// (ziren_amd/, libzkm_hip.so) never links, imports or executes anything in this directory.
//
// The functions take and return exactly the data the product's C ABI (include/zkm_hip.h)
// does — Montgomery words, row-major host matrices, the same chip descriptors and the same
// ShardProof word stream — so a parity test feeds both sides identical bytes.
#include "stark.hpp"
#include "tracegen.hpp"
#include <cstdio>
#include <omp.h>

using namespace orc;

static thread_local std::string g_err;

static Matrix load_matrix(const uint32_t* p, size_t h, size_t w) {
  Matrix m(h, w);
  for (size_t i = 0; i < h * w; i++) m.v[i] = from_monty(p[i]);
  return m;
}
static void put_digest(std::vector<uint32_t>& o, const Digest& d) { for (int i = 0; i < 8; i++) o.push_back(to_monty(d.d[i])); }
static void put_ext(std::vector<uint32_t>& o, const E& e) { for (int i = 0; i < 4; i++) o.push_back(to_monty(e.c[i])); }
static void put_exts(std::vector<uint32_t>& o, const std::vector<E>& v) { for (auto& e : v) put_ext(o, e); }

static std::vector<uint32_t> serialize_proof(const ShardProof& p) {
  std::vector<uint32_t> o;
  put_digest(o, p.main_commit); put_digest(o, p.perm_commit); put_digest(o, p.quotient_commit);
  o.push_back((uint32_t)p.chips.size());
  for (size_t i = 0; i < p.chips.size(); i++) {
    const ChipOpenedValues& c = p.chips[i];
    o.push_back((uint32_t)p.order[i]);
    o.push_back(c.log_degree);
    o.push_back((uint32_t)c.prep_local.size()); put_exts(o, c.prep_local); put_exts(o, c.prep_next);
    o.push_back((uint32_t)c.main_local.size()); put_exts(o, c.main_local); put_exts(o, c.main_next);
    o.push_back((uint32_t)c.perm_local.size()); put_exts(o, c.perm_local); put_exts(o, c.perm_next);
    o.push_back((uint32_t)c.quotient.size());
    for (auto& q : c.quotient) put_exts(o, q);
    for (int k = 0; k < 14; k++) o.push_back(to_monty(c.global_sum[k]));
    put_ext(o, c.local_sum);
  }
  const FriProof& f = p.fri;
  o.push_back((uint32_t)f.commit_phase_commits.size());
  for (auto& d : f.commit_phase_commits) put_digest(o, d);
  o.push_back((uint32_t)f.queries.size());
  for (auto& q : f.queries) {
    o.push_back((uint32_t)q.input_proof.size());
    for (auto& bo : q.input_proof) {
      o.push_back((uint32_t)bo.opened_values.size());
      for (auto& row : bo.opened_values) { o.push_back((uint32_t)row.size()); for (F v : row) o.push_back(to_monty(v)); }
      o.push_back((uint32_t)bo.proof.size());
      for (auto& d : bo.proof) put_digest(o, d);
    }
    o.push_back((uint32_t)q.steps.size());
    for (auto& s : q.steps) {
      put_ext(o, s.sibling_value);
      o.push_back((uint32_t)s.proof.size());
      for (auto& d : s.proof) put_digest(o, d);
    }
  }
  put_ext(o, f.final_poly);
  o.push_back(to_monty(f.pow_witness));
  o.push_back((uint32_t)p.public_values.size());
  for (F v : p.public_values) o.push_back(to_monty(v));
  return o;
}

struct Reader {
  const uint32_t* p; size_t n, pos = 0;
  uint32_t u() { if (pos >= n) throw std::runtime_error("proof stream truncated"); return p[pos++]; }
  F f() { return from_monty(u()); }
  Digest digest() { Digest d; for (int i = 0; i < 8; i++) d.d[i] = f(); return d; }
  E ext() { E e; for (int i = 0; i < 4; i++) e.c[i] = f(); return e; }
  std::vector<E> exts(size_t k) { std::vector<E> v(k); for (auto& e : v) e = ext(); return v; }
};

static ShardProof parse_proof(const uint32_t* w, size_t n) {
  Reader r{w, n};
  ShardProof p;
  p.main_commit = r.digest(); p.perm_commit = r.digest(); p.quotient_commit = r.digest();
  size_t nc = r.u();
  for (size_t i = 0; i < nc; i++) {
    ChipOpenedValues c;
    p.order.push_back(r.u());
    c.log_degree = r.u();
    size_t k = r.u(); c.prep_local = r.exts(k); c.prep_next = r.exts(k);
    k = r.u(); c.main_local = r.exts(k); c.main_next = r.exts(k);
    k = r.u(); c.perm_local = r.exts(k); c.perm_next = r.exts(k);
    k = r.u(); for (size_t j = 0; j < k; j++) c.quotient.push_back(r.exts(4));
    for (int j = 0; j < 14; j++) c.global_sum[j] = r.f();
    c.local_sum = r.ext();
    p.chips.push_back(std::move(c));
  }
  FriProof& f = p.fri;
  size_t ncp = r.u();
  for (size_t i = 0; i < ncp; i++) f.commit_phase_commits.push_back(r.digest());
  size_t nq = r.u();
  for (size_t q = 0; q < nq; q++) {
    QueryProof qp;
    size_t nr = r.u();
    for (size_t j = 0; j < nr; j++) {
      BatchOpening bo;
      size_t nm = r.u();
      for (size_t m = 0; m < nm; m++) { size_t wd = r.u(); std::vector<F> row(wd); for (auto& v : row) v = r.f(); bo.opened_values.push_back(row); }
      size_t pl = r.u();
      for (size_t l = 0; l < pl; l++) bo.proof.push_back(r.digest());
      qp.input_proof.push_back(std::move(bo));
    }
    size_t ns = r.u();
    for (size_t s = 0; s < ns; s++) {
      CommitPhaseStep st; st.sibling_value = r.ext();
      size_t pl = r.u();
      for (size_t l = 0; l < pl; l++) st.proof.push_back(r.digest());
      qp.steps.push_back(std::move(st));
    }
    f.queries.push_back(std::move(qp));
  }
  f.final_poly = r.ext();
  f.pow_witness = r.f();
  size_t npv = r.u();
  for (size_t i = 0; i < npv; i++) p.public_values.push_back(r.f());
  if (r.pos != n) throw std::runtime_error("proof stream has trailing words");
  return p;
}

static Challenger load_challenger(const zkm_challenger* c) {
  Challenger ch;
  for (int i = 0; i < 16; i++) ch.state[i] = from_monty(c->sponge_state[i]);
  for (uint32_t i = 0; i < c->num_inputs; i++) ch.in.push_back(from_monty(c->input_buffer[i]));
  for (uint32_t i = 0; i < c->num_outputs; i++) ch.out.push_back(from_monty(c->output_buffer[i]));
  return ch;
}
static void store_challenger(const Challenger& ch, zkm_challenger* c) {
  memset(c, 0, sizeof *c);
  for (int i = 0; i < 16; i++) c->sponge_state[i] = to_monty(ch.state[i]);
  c->num_inputs = (uint32_t)ch.in.size();
  for (size_t i = 0; i < ch.in.size(); i++) c->input_buffer[i] = to_monty(ch.in[i]);
  c->num_outputs = (uint32_t)ch.out.size();
  for (size_t i = 0; i < ch.out.size(); i++) c->output_buffer[i] = to_monty(ch.out[i]);
}

#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception& e) { g_err = e.what(); return -1; } return 0;

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }

// field / extension vectors (canonical words in, canonical words out) — for the KAT tests
void orc_field_ops(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* add, uint32_t* sub, uint32_t* mul,
                   uint32_t* inv, uint32_t* a_monty) {
  for (size_t i = 0; i < n; i++) {
    add[i] = fadd(a[i], b[i]); sub[i] = fsub(a[i], b[i]); mul[i] = fmul(a[i], b[i]);
    inv[i] = a[i] ? finv(a[i]) : 0; a_monty[i] = to_monty(a[i]);
  }
}
void orc_from_monty(const uint32_t* in, uint32_t* out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = from_monty(in[i]); }
void orc_to_monty(const uint32_t* in, uint32_t* out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = to_monty(in[i]); }
// ext ops on Montgomery words: out_mul = a*b, out_inv = 1/a
void orc_ext_ops(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out_mul, uint32_t* out_inv) {
  for (size_t i = 0; i < n; i++) {
    E x, y;
    for (int k = 0; k < 4; k++) { x.c[k] = from_monty(a[4 * i + k]); y.c[k] = from_monty(b[4 * i + k]); }
    E m = emul(x, y), iv = eis_zero(x) ? ezero() : einv(x);
    for (int k = 0; k < 4; k++) { out_mul[4 * i + k] = to_monty(m.c[k]); out_inv[4 * i + k] = to_monty(iv.c[k]); }
  }
}
uint32_t orc_two_adic_generator(int bits) { return to_monty(two_adic_generator(bits)); }

// Poseidon2 on n states of 16 Montgomery words, in place.
void orc_poseidon2_permute_batch(uint32_t* states, size_t n) {
#pragma omp parallel for
  for (size_t i = 0; i < n; i++) {
    F s[16];
    for (int k = 0; k < 16; k++) s[k] = from_monty(states[16 * i + k]);
    poseidon2_permute(s);
    for (int k = 0; k < 16; k++) states[16 * i + k] = to_monty(s[k]);
  }
}
void orc_hash(const uint32_t* in, size_t len, uint32_t out[8]) {
  std::vector<F> v(len);
  for (size_t i = 0; i < len; i++) v[i] = from_monty(in[i]);
  Digest d = hash_slice(v.data(), len);
  for (int i = 0; i < 8; i++) out[i] = to_monty(d.d[i]);
}
void orc_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
  Digest a, b;
  for (int i = 0; i < 8; i++) { a.d[i] = from_monty(l[i]); b.d[i] = from_monty(r[i]); }
  Digest d = compress(a, b);
  for (int i = 0; i < 8; i++) out[i] = to_monty(d.d[i]);
}

void orc_challenger_init(zkm_challenger* c) { memset(c, 0, sizeof *c); }
void orc_challenger_observe(zkm_challenger* c, const uint32_t* v, size_t n) {
  Challenger ch = load_challenger(c);
  for (size_t i = 0; i < n; i++) ch.observe(from_monty(v[i]));
  store_challenger(ch, c);
}
uint32_t orc_challenger_sample(zkm_challenger* c) {
  Challenger ch = load_challenger(c);
  F v = ch.sample();
  store_challenger(ch, c);
  return to_monty(v);
}
uint32_t orc_challenger_sample_bits(zkm_challenger* c, uint32_t bits) {
  Challenger ch = load_challenger(c);
  uint32_t v = ch.sample_bits(bits);
  store_challenger(ch, c);
  return v;
}
uint32_t orc_challenger_grind(zkm_challenger* c, uint32_t bits) {
  Challenger ch = load_challenger(c);
  F w = ch.grind(bits);
  store_challenger(ch, c);
  return to_monty(w);
}

int orc_coset_lde_batch(const uint32_t* in, size_t h, size_t w, uint32_t log_blowup, uint32_t lde_shift, uint32_t* out) {
  ORC_TRY
  Matrix m = load_matrix(in, h, w);
  Matrix l = coset_lde_matrix_bitrev(m, log_blowup, from_monty(lde_shift));
  for (size_t i = 0; i < l.v.size(); i++) out[i] = to_monty(l.v[i]);
  ORC_CATCH
}

// Pcs::commit of n matrices; optionally returns every LDE (row-major, bit-reversed rows, back to back)
// and all digest layers (layer 0 first) back to back.
int orc_pcs_commit(size_t n_mats, const uint32_t* const* mats, const size_t* heights, const size_t* widths,
                   const uint32_t* domain_shifts, uint32_t log_blowup, uint32_t root_out[8], uint32_t* ldes_out,
                   uint32_t* layers_out) {
  ORC_TRY
  std::vector<Matrix> ms; std::vector<F> sh;
  for (size_t i = 0; i < n_mats; i++) { ms.push_back(load_matrix(mats[i], heights[i], widths[i])); sh.push_back(domain_shifts ? from_monty(domain_shifts[i]) : 1); }
  PcsData d = pcs_commit(ms, sh, log_blowup);
  for (int i = 0; i < 8; i++) root_out[i] = to_monty(d.tree.root().d[i]);
  if (ldes_out) { size_t pos = 0; for (auto& l : d.tree.leaves) for (F v : l.v) ldes_out[pos++] = to_monty(v); }
  if (layers_out) { size_t pos = 0; for (auto& layer : d.tree.layers) for (auto& dg : layer) for (int k = 0; k < 8; k++) layers_out[pos++] = to_monty(dg.d[k]); }
  ORC_CATCH
}

// open_batch on a fresh commit of the given matrices (test helper): values then proof.
int orc_pcs_open_batch(size_t n_mats, const uint32_t* const* mats, const size_t* heights, const size_t* widths,
                       const uint32_t* domain_shifts, uint32_t log_blowup, size_t index, uint32_t* values_out,
                       uint32_t* proof_out, int* verify_ok) {
  ORC_TRY
  std::vector<Matrix> ms; std::vector<F> sh;
  for (size_t i = 0; i < n_mats; i++) { ms.push_back(load_matrix(mats[i], heights[i], widths[i])); sh.push_back(domain_shifts ? from_monty(domain_shifts[i]) : 1); }
  PcsData d = pcs_commit(ms, sh, log_blowup);
  BatchOpening bo = mmcs_open_batch(d.tree, index);
  size_t pos = 0;
  for (auto& row : bo.opened_values) for (F v : row) values_out[pos++] = to_monty(v);
  pos = 0;
  for (auto& dg : bo.proof) for (int k = 0; k < 8; k++) proof_out[pos++] = to_monty(dg.d[k]);
  std::vector<size_t> dims;
  for (auto& l : d.tree.leaves) dims.push_back(l.h);
  *verify_ok = mmcs_verify_batch(d.tree.root(), dims, index, bo.opened_values, bo.proof);
  ORC_CATCH
}

// Mmcs::verify_batch (crates/recursion/circuit/src/fri.rs:363-405) on caller-supplied openings.
int orc_mmcs_verify_batch(const uint32_t root[8], size_t n_mats, const size_t* heights, const size_t* widths, size_t index,
                          const uint32_t* values, const uint32_t* proof, size_t proof_len, int* ok) {
  ORC_TRY
  Digest commit;
  for (int i = 0; i < 8; i++) commit.d[i] = from_monty(root[i]);
  std::vector<size_t> dims(heights, heights + n_mats);
  std::vector<std::vector<F>> opened;
  size_t pos = 0;
  for (size_t m = 0; m < n_mats; m++) {
    std::vector<F> row(widths[m]);
    for (auto& v : row) v = from_monty(values[pos++]);
    opened.push_back(row);
  }
  std::vector<Digest> path(proof_len);
  for (size_t l = 0; l < proof_len; l++) for (int k = 0; k < 8; k++) path[l].d[k] = from_monty(proof[8 * l + k]);
  *ok = mmcs_verify_batch(commit, dims, index, opened, path);
  ORC_CATCH
}

struct orc_pk { ProvingKey pk; VerifyingKey vk; };

int orc_pk_setup(size_t n_prep, const uint32_t* const* prep, const size_t* heights, const size_t* widths,
                 const uint32_t* local_only, uint32_t pc_start, const uint32_t igcs[14], uint32_t log_blowup,
                 orc_pk** out) {
  ORC_TRY
  orc_pk* k = new orc_pk();
  ProvingKey& pk = k->pk;
  for (size_t i = 0; i < n_prep; i++) { pk.prep_traces.push_back(load_matrix(prep[i], heights[i], widths[i])); pk.prep_local_only.push_back(local_only[i]); }
  pk.has_prep = n_prep > 0;
  if (pk.has_prep) { pk.data = pcs_commit(pk.prep_traces, {}, log_blowup); pk.commit = pk.data.tree.root(); }
  else memset(pk.commit.d, 0, sizeof pk.commit.d);
  pk.pc_start = from_monty(pc_start);
  for (int i = 0; i < 14; i++) pk.initial_global_cumulative_sum[i] = from_monty(igcs[i]);
  k->vk.commit = pk.commit; k->vk.pc_start = pk.pc_start;
  memcpy(k->vk.initial_global_cumulative_sum, pk.initial_global_cumulative_sum, sizeof pk.initial_global_cumulative_sum);
  k->vk.has_prep = pk.has_prep;
  for (auto& t : pk.prep_traces) k->vk.prep_log_heights.push_back(log2_strict(t.h));
  *out = k;
  ORC_CATCH
}
void orc_pk_commitment(const orc_pk* k, uint32_t out[8]) { for (int i = 0; i < 8; i++) out[i] = to_monty(k->pk.commit.d[i]); }
void orc_pk_observe_into(const orc_pk* k, zkm_challenger* c) {
  Challenger ch = load_challenger(c);
  k->pk.observe_into(ch);
  store_challenger(ch, c);
}
void orc_pk_free(orc_pk* k) { delete k; }

// seconds spent in coset LDEs since the last call with reset != 0 (bench.py: the n log n share of the CPU baseline)
double orc_lde_seconds(int reset) { double t = lde_seconds(); if (reset) lde_seconds() = 0; return t; }

// commit + open of one shard. traces: row-major Montgomery host matrices, caller order.
// timings_out (nullable): [commit_seconds, open_seconds].
int orc_prove_shard(const orc_pk* k, size_t n_chips, const zkm_chip_desc* descs, const uint32_t* const* traces,
                    const size_t* heights, const uint32_t* public_values, size_t n_pv, const zkm_fri_config* fri,
                    uint32_t num_pv_elts, zkm_challenger* challenger, uint32_t* proof_out, size_t proof_cap,
                    size_t* proof_len, double* timings_out) {
  ORC_TRY
  std::vector<Chip> chips; std::vector<std::string> names; std::vector<Matrix> ms;
  for (size_t i = 0; i < n_chips; i++) {
    chips.push_back(parse_chip(descs[i]));
    names.push_back(descs[i].name);
    ms.push_back(load_matrix(traces[i], heights[i], descs[i].main_width));
  }
  std::vector<F> pv(n_pv);
  for (size_t i = 0; i < n_pv; i++) pv[i] = from_monty(public_values[i]);
  FriConfig cfg{(int)fri->log_blowup, (int)fri->num_queries, (int)fri->proof_of_work_bits};
  Challenger ch = load_challenger(challenger);
  double t0 = omp_get_wtime();
  MainData md = shard_commit(names, ms, pv, cfg.log_blowup);
  double t1 = omp_get_wtime();
  ShardProof p = shard_open(k->pk, md, chips, cfg, num_pv_elts, ch);
  double t2 = omp_get_wtime();
  if (timings_out) { timings_out[0] = t1 - t0; timings_out[1] = t2 - t1; }
  store_challenger(ch, challenger);
  std::vector<uint32_t> s = serialize_proof(p);
  *proof_len = s.size();
  if (s.size() > proof_cap) throw std::runtime_error("proof buffer too small");
  memcpy(proof_out, s.data(), s.size() * 4);
  ORC_CATCH
}

// generate_permutation_trace (crates/stark/src/permutation.rs:102-196) of one chip on its own: main (and prep, nullable) row-major Montgomery, the two
// permutation challenges as 8 Montgomery words (alpha, beta); out: height x 4 * perm_ext_width row-major Montgomery, local_sum: 4 words.
int orc_permutation_trace(const zkm_chip_desc* desc, const uint32_t* main, const uint32_t* prep, size_t height, const uint32_t challenges[8], uint32_t* out,
                          size_t out_cap, uint32_t local_sum[4]) {
  ORC_TRY
  const Chip chip = parse_chip(*desc);
  const Matrix m = load_matrix(main, height, desc->main_width);
  Matrix pm;
  if (prep) pm = load_matrix(prep, height, desc->prep_width);
  E alpha, beta, sum;
  for (int c = 0; c < 4; c++) { alpha.c[c] = from_monty(challenges[c]); beta.c[c] = from_monty(challenges[4 + c]); }
  const Matrix t = generate_permutation_trace(chip, prep ? &pm : nullptr, m, alpha, beta, sum);
  if (t.h * t.w > out_cap) throw std::runtime_error("permutation trace buffer too small");
  for (size_t r = 0; r < t.h; r++)
    for (size_t c = 0; c < t.w; c++) out[r * t.w + c] = to_monty(t.at(r, c));
  for (int c = 0; c < 4; c++) local_sum[c] = to_monty(sum.c[c]);
  ORC_CATCH
}

// Verifier::verify_shard on a proof stream. challenger: post vk.observe_into. *verdict = 0 accept.
int orc_verify_shard(const orc_pk* k, size_t n_chips, const zkm_chip_desc* descs, const zkm_fri_config* fri,
                     uint32_t num_pv_elts, zkm_challenger* challenger, const uint32_t* proof, size_t proof_len,
                     int* verdict) {
  ORC_TRY
  std::vector<Chip> chips;
  for (size_t i = 0; i < n_chips; i++) chips.push_back(parse_chip(descs[i]));
  FriConfig cfg{(int)fri->log_blowup, (int)fri->num_queries, (int)fri->proof_of_work_bits};
  Challenger ch = load_challenger(challenger);
  ShardProof p = parse_proof(proof, proof_len);
  if (p.chips.size() != n_chips) { *verdict = 1; return 0; }
  *verdict = verify_shard(k->vk, chips, cfg, num_pv_elts, ch, p);
  store_challenger(ch, challenger);
  ORC_CATCH
}

// ---- ALU chip trace generation (tracegen.hpp) ------------------------------------------------------------
// events: n_events packed 28-byte AluEvent records. out: rows x width row-major Montgomery words, the matrix the
// reference's generate_trace returns.
size_t orc_tracegen_alu_width(int chip) { try { return tracegen::chip_width(chip); } catch (...) { return 0; } }
int orc_tracegen_alu_rows(size_t n_events, int fixed_log2_rows, size_t* rows) {
  ORC_TRY
  *rows = tracegen::padded_rows(n_events, fixed_log2_rows);
  ORC_CATCH
}
int orc_tracegen_alu(int chip, const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap) {
  ORC_TRY
  size_t h;
  std::vector<F> t = tracegen::generate(chip, (const tracegen::AluEvent*)events, n_events, fixed_log2_rows, &h);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}
// Index of the first event whose row breaks one of the reference's in-line sanity identities, or -1.
long orc_tracegen_alu_check(int chip, const void* events, size_t n_events) {
  const tracegen::AluEvent* ev = (const tracegen::AluEvent*)events;
  size_t h;
  std::vector<F> t = tracegen::generate(chip, ev, n_events, -1, &h);
  const size_t w = tracegen::chip_width(chip);
  for (size_t i = 0; i < n_events; i++)
    if (!tracegen::check_row(chip, ev[i], t.data() + i * w)) return (long)i;
  return -1;
}

// ByteChip::trace() and ByteChip::generate_trace over ALU event streams: row-major Montgomery matrices
int orc_tracegen_byte_table(uint32_t* out /* 65536 x 12 */) {
  ORC_TRY
  std::vector<F> t = tracegen::byte_table();
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}
int orc_tracegen_byte_mults(size_t n_streams, const int* chips, const void* const* events, const size_t* n_events,
                            const uint32_t* extra_counts, uint32_t* out /* 65536 x 10 */) {
  ORC_TRY
  std::vector<F> t = tracegen::byte_mults(n_streams, chips, (const tracegen::AluEvent* const*)events, n_events, extra_counts);
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}

// Jump chip (JumpEvent records, 28 bytes): row-major Montgomery trace, 66 columns
int orc_tracegen_jump(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap) {
  ORC_TRY
  size_t h;
  std::vector<F> t = tracegen::generate_jump((const tracegen::JumpEvent*)events, n_events, fixed_log2_rows, &h);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}

int orc_tracegen_mov_cond(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap) {
  ORC_TRY
  size_t h;
  std::vector<F> t = tracegen::generate_mov_cond((const tracegen::MovCondEvent*)events, n_events, fixed_log2_rows, &h);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}

// Branch chip (BranchEvent = JumpEvent layout): row-major Montgomery trace, 62 columns; byte_counts (may be NULL) is a
// 65536 x 10 row-major array of plain counters that receives the byte lookups of the rows (not-taken branches)
int orc_tracegen_branch(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_branch((const tracegen::JumpEvent*)events, n_events, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}

// Mul chip (CompAluEvent, 64 bytes): row-major Montgomery trace, 58 columns; byte_counts as for orc_tracegen_branch
int orc_tracegen_mul(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_mul((const tracegen::CompAluEvent*)events, n_events, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}

// DivRem chip (CompAluEvent): row-major Montgomery trace, 106 columns; byte_counts as for orc_tracegen_branch
int orc_tracegen_divrem(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_divrem((const tracegen::CompAluEvent*)events, n_events, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}

// Cpu chip: CpuEventFfi records (280 bytes) + the program's InstructionFfi records (24 bytes); 67 columns
int orc_tracegen_cpu(const void* events, size_t n_events, const void* program, size_t n_instr, uint32_t pc_base, uint32_t shard,
                     int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_cpu((const tracegen::CpuEvent*)events, n_events, (const tracegen::Instruction*)program, n_instr, pc_base,
                                            shard, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}
// Program chip: which = 0 preprocessed table (14 columns), 1 multiplicities (1 column)
int orc_tracegen_program(int which, const void* events, size_t n_events, const void* program, size_t n_instr, uint32_t pc_base,
                         int fixed_log2_rows, uint32_t* out, size_t out_cap) {
  ORC_TRY
  size_t h;
  std::vector<F> t = which == 0 ? tracegen::generate_program_prep((const tracegen::Instruction*)program, n_instr, pc_base, fixed_log2_rows, &h)
                                : tracegen::generate_program_mult((const tracegen::CpuEvent*)events, n_events, n_instr, pc_base, fixed_log2_rows, &h);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}

// MemoryLocal chip: MemoryLocalEvent records (28 bytes), four per row, 56 columns
int orc_tracegen_memory_local(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows) {
  ORC_TRY
  size_t h;
  std::vector<F> t = tracegen::generate_memory_local((const tracegen::MemoryLocalEvent*)events, n_events, fixed_log2_rows, &h);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  }
  ORC_CATCH
}

// MemoryInstructions chip (MemInstrEvent, 64 bytes): 79 columns; byte_counts as for orc_tracegen_branch
int orc_tracegen_memory_instrs(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_memory_instrs((const tracegen::MemInstrEvent*)events, n_events, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}

// septic extension known answers: out[0..42) = (z^i)^p, out[42..84) = (z^i)^(p^2), i = 1..6 (canonical), by exponentiation; out[84..91) = a * b
// and out[91..98) = the square root of a^2 normalised to y6 <= (p - 1) / 2 (0 when the root's last coefficient is 0), for canonical a, b
int orc_septic_known_answers(const uint32_t a[7], const uint32_t b[7], uint32_t out[98]) {
  ORC_TRY
  septic::S7 z = septic::s_zero(), x, y;
  z.c[1] = 1;
  septic::S7 zi = z;
  for (int i = 0; i < 6; i++) {
    const septic::S7 f = septic::s_frob(zi), f2 = septic::s_frob(f);
    for (int k = 0; k < 7; k++) { out[7 * i + k] = f.c[k]; out[42 + 7 * i + k] = f2.c[k]; }
    zi = septic::s_mul(zi, z);
  }
  for (int k = 0; k < 7; k++) { x.c[k] = a[k] % P; y.c[k] = b[k] % P; }
  const septic::S7 prod = septic::s_mul(x, y);
  septic::S7 root;
  if (!septic::s_sqrt(septic::s_mul(x, x), &root)) throw std::runtime_error("septic: a square has no root");
  if (root.c[6] >= (P + 1) / 2) root = septic::s_neg(root);
  for (int k = 0; k < 7; k++) { out[84 + k] = prod.c[k]; out[91 + k] = root.c[k]; }
  ORC_CATCH
}

// the machine-level check on the global digests (machine.rs:657-671): digests = n x 14 Montgomery words (x, y); out = their SepticDigest sum,
// *is_zero = it equals the zero digest
int orc_global_digest_sum(const uint32_t* digests, size_t n, uint32_t out[14], int* is_zero) {
  ORC_TRY
  std::vector<septic::Point> pts(n);
  septic::Point zero;
  for (int k = 0; k < 7; k++) { zero.x.c[k] = SEPTIC_START_X[k]; zero.y.c[k] = SEPTIC_START_Y[k]; }
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 7; k++) { pts[i].x.c[k] = from_monty(digests[14 * i + k]); pts[i].y.c[k] = from_monty(digests[14 * i + 7 + k]); }
  const septic::Point sum = septic::digest_sum(pts.data(), n, zero);
  for (int k = 0; k < 7; k++) { out[k] = to_monty(sum.x.c[k]); out[7 + k] = to_monty(sum.y.c[k]); }
  *is_zero = septic::s_eq(sum.x, zero.x) && septic::s_eq(sum.y, zero.y);
  ORC_CATCH
}

// Global chip (GlobalLookupEvent, 32 bytes): 99 columns; byte_counts as for orc_tracegen_branch
int orc_tracegen_global(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_global((const tracegen::GlobalLookupEvent*)events, n_events, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}

// recursion Poseidon2Wide chip (degree 3): events = n x 32 Montgomery words (input[16], output[16]); 313 columns
int orc_tracegen_poseidon2_wide(const uint32_t* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap) {
  ORC_TRY
  size_t h;
  std::vector<F> ev(n_events * 32);
  for (size_t i = 0; i < ev.size(); i++) ev[i] = from_monty(events[i]);
  std::vector<F> t = tracegen::generate_poseidon2_wide(ev.data(), n_events, fixed_log2_rows, &h);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}

// SyscallInstrs chip (SyscallEvent, 56 bytes)
int orc_tracegen_syscall_instrs(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap) {
  ORC_TRY
  size_t h;
  std::vector<F> t = tracegen::generate_syscall_instrs((const tracegen::SyscallEvent*)events, n_events, fixed_log2_rows, &h);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  ORC_CATCH
}

// MiscInstrs chip (MiscEvent, 60 bytes): 72 columns; byte_counts as for orc_tracegen_branch
int orc_tracegen_misc_instrs(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_misc_instrs((const tracegen::MiscEvent*)events, n_events, fixed_log2_rows, &h, byte_counts ? cnt.data() : nullptr);
  if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
  for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  ORC_CATCH
}

// recursion ExpReverseBitsLen chip: bases (n Montgomery words), bits (Montgomery words, end to end), offsets (n + 1 plain indices)
int orc_tracegen_exp_reverse_bits(const uint32_t* bases, const uint32_t* bits, const uint32_t* offsets, size_t n_events, int fixed_log2_rows,
                                  uint32_t* out, size_t out_cap, size_t* rows) {
  ORC_TRY
  size_t h;
  const size_t nb = n_events ? offsets[n_events] : 0;
  std::vector<F> b(n_events), x(nb);
  for (size_t i = 0; i < n_events; i++) b[i] = from_monty(bases[i]);
  for (size_t i = 0; i < nb; i++) x[i] = from_monty(bits[i]);
  std::vector<F> t = tracegen::generate_exp_reverse_bits(b.data(), x.data(), offsets, n_events, fixed_log2_rows, &h);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  }
  ORC_CATCH
}

// recursion Poseidon2Skinny chip: events as for orc_tracegen_poseidon2_wide; eleven rows of 28 columns per event
int orc_tracegen_poseidon2_skinny(const uint32_t* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows) {
  ORC_TRY
  size_t h;
  std::vector<F> ev(n_events * 32);
  for (size_t i = 0; i < ev.size(); i++) ev[i] = from_monty(events[i]);
  std::vector<F> t = tracegen::generate_poseidon2_skinny(ev.data(), n_events, fixed_log2_rows, &h);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  }
  ORC_CATCH
}

// MemoryGlobalInit / MemoryGlobalFinalize chips: MemoryInitializeFinalizeEvents (16 bytes), 111 columns; previous_addr = the address the
// shard's public values carry in previous_init_addr_bits / previous_finalize_addr_bits. Call with out = NULL to get the row count.
int orc_tracegen_memory_global(const void* events, size_t n_events, uint32_t previous_addr, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows) {
  ORC_TRY
  size_t h;
  std::vector<F> t = tracegen::generate_memory_global((const tracegen::MemoryInitFinalizeEvent*)events, n_events, previous_addr, fixed_log2_rows, &h);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
  }
  ORC_CATCH
}

// SyscallCore (precompile = 0: the events are filtered as the chip filters them) / SyscallPrecompile (1) chips: SyscallEvents, 11 columns
int orc_tracegen_syscall(const void* events, size_t n_events, int precompile, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows,
                         uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_syscall((const tracegen::SyscallEvent*)events, n_events, precompile != 0, fixed_log2_rows, &h,
                                                byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// Poseidon2Permute precompile chip: flattened Poseidon2PermuteEvents (99 words), 973 columns
int orc_tracegen_poseidon2_permute(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows,
                                   uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_poseidon2_permute((const tracegen::Poseidon2PermuteEvent*)events, n_events, fixed_log2_rows, &h,
                                                          byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// KeccakSponge precompile chip: KeccakSpongeEvents cut into 36-word blocks (337 words each), 24 rows per block, 3531 columns
int orc_tracegen_keccak_sponge(const void* blocks, size_t n_blocks, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows,
                               uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_keccak_sponge((const tracegen::KeccakSpongeBlock*)blocks, n_blocks, fixed_log2_rows, &h,
                                                      byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// SHA-256 precompile chips: flattened ShaExtendEvents (1251 words, 48 rows each, 176 columns), ShaCompressEvents (412 words, 80 rows, 262)
int orc_tracegen_sha_extend(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows,
                            uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_sha_extend((const tracegen::ShaExtendEvent*)events, n_events, fixed_log2_rows, &h,
                                                   byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}
int orc_tracegen_sha_compress(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows,
                              uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_sha_compress((const tracegen::ShaCompressEvent*)events, n_events, fixed_log2_rows, &h,
                                                     byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// EdAddAssign precompile chip: flattened EllipticCurveAddEvents (180 words), one row each, 1861 columns
int orc_tracegen_ed_add(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_ed_add((const tracegen::EdAddEvent*)events, n_events, fixed_log2_rows, &h, byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// EdDecompress precompile chip: flattened EdDecompressEvents (92 words), one row each, 1566 columns
int orc_tracegen_ed_decompress(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows,
                               uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_ed_decompress((const tracegen::EdDecompressEvent*)events, n_events, fixed_log2_rows, &h,
                                                      byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// Short-Weierstrass add / double chips of any curve: flattened events (4 + 11 W or 3 + 6 W words, W = n_limbs / 2), one row each
int orc_tracegen_weierstrass(const uint32_t* events, size_t n_events, int is_double, int n_limbs, const uint8_t* modulus, const uint8_t* a,
                             uint32_t witness_offset, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_weierstrass(events, n_events, is_double != 0, n_limbs, modulus, a, witness_offset, fixed_log2_rows, &h,
                                                    byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// <Curve>Decompress chips: flattened events (4 + 11 W words, W = n_limbs / 4), one row each; the curve as bytes (modulus, a, b, generator x)
int orc_tracegen_weierstrass_decompress(const uint32_t* events, size_t n_events, int n_limbs, const uint8_t* modulus, const uint8_t* a, const uint8_t* b,
                                        const uint8_t* generator_x, uint32_t witness_offset, int lexicographic, int fixed_log2_rows, uint32_t* out,
                                        size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_weierstrass_decompress(events, n_events, n_limbs, modulus, a, b, generator_x, witness_offset, lexicographic != 0,
                                                               fixed_log2_rows, &h, byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// Uint256MulMod chip: flattened Uint256MulEvents (132 words), one row each
int orc_tracegen_uint256_mul(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_uint256_mul((const tracegen::Uint256MulEvent*)events, n_events, fixed_log2_rows, &h, byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// U256XU2048Mul chip: flattened U256xU2048MulEvents (808 words), one row each
int orc_tracegen_u256x2048_mul(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_u256x2048_mul((const tracegen::U256x2048MulEvent*)events, n_events, fixed_log2_rows, &h, byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// BooleanCircuitGarble chip: one 103-word row record per row (header rows and gate rows)
int orc_tracegen_boolean_circuit_garble(const void* rows_in, size_t n_rows, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_boolean_circuit_garble((const tracegen::GarbleRow*)rows_in, n_rows, fixed_log2_rows, &h, byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// SysLinux chip: flattened LinuxEvents (23 words), one row each
int orc_tracegen_sys_linux(const void* events, size_t n_events, int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_sys_linux((const tracegen::LinuxEvent*)events, n_events, fixed_log2_rows, &h, byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

// Field-tower chips (kind 0 FpOp, 1 Fp2AddSub, 2 Fp2Mul) over a base field given by its modulus bytes
int orc_tracegen_fp_tower(const uint32_t* events, size_t n_events, int kind, int n_limbs, const uint8_t* modulus, uint32_t witness_offset,
                          int fixed_log2_rows, uint32_t* out, size_t out_cap, size_t* rows, uint32_t* byte_counts) {
  ORC_TRY
  size_t h;
  std::vector<uint64_t> cnt(byte_counts ? tracegen::BYTE_ROWS * tracegen::NUM_BYTE_OPS : 0, 0);
  std::vector<F> t = tracegen::generate_fp_tower(events, n_events, kind, n_limbs, modulus, witness_offset, fixed_log2_rows, &h,
                                                 byte_counts && out ? cnt.data() : nullptr);
  *rows = h;
  if (out) {
    if (t.size() > out_cap) throw std::runtime_error("trace buffer too small");
    for (size_t i = 0; i < t.size(); i++) out[i] = to_monty(t[i]);
    for (size_t i = 0; i < cnt.size(); i++) byte_counts[i] += (uint32_t)cnt[i];
  }
  ORC_CATCH
}

}  // extern "C"
