// TEST INFRASTRUCTURE — CPU oracle. Never linked, imported or executed by the product path.
//
// The shard prover (commit + open) and the shard verifier of crates/stark, restated on the CPU
// over the oracle's PCS. Follows:
//   CpuProver::commit                      crates/stark/src/prover.rs:258-292
//   CpuProver::open (transcript order)     crates/stark/src/prover.rs:298-653
//   generate_permutation_trace (LogUp)     crates/stark/src/permutation.rs:29-69,102-196
//   quotient_values                        crates/stark/src/quotient.rs:19-171, folder.rs:79-102
//   Verifier::verify_shard                 crates/stark/src/verifier.rs:30-246,332-435
//   pk.observe_into                        crates/stark/src/machine.rs:79-86
// Chip constraints arrive as the bytecode of include/zkm_hip.h (in the real system the Rust
// shim records `Air::eval`, chip.rs:257-276, with a symbolic builder); the same bytecode is
// interpreted by the prover over the quotient coset and by the verifier at zeta.
#pragma once
#include "pcs.hpp"
#include "zkm_hip.h"
#include <array>
#include <string>
#include <stdexcept>

namespace orc {

struct PairCol {  // p3_air::VirtualPairCol
  std::vector<std::pair<uint32_t, F>> terms;  // ((is_main<<31)|col, weight)
  F constant = 0;
  template <class T, class FromF, class Add, class Mul>
  T apply(const T* prep, const T* main, FromF fromf, Add add, Mul mulf) const {
    T acc = fromf(constant);
    for (auto& t : terms) {
      uint32_t col = t.first & 0x7fffffffu;
      const T& v = (t.first >> 31) ? main[col] : prep[col];
      acc = add(acc, mulf(v, t.second));
    }
    return acc;
  }
  F apply_f(const F* prep, const F* main) const {
    F acc = constant;
    for (auto& t : terms) {
      uint32_t col = t.first & 0x7fffffffu;
      acc = fadd(acc, fmul((t.first >> 31) ? main[col] : prep[col], t.second));
    }
    return acc;
  }
};
struct Lookup { uint32_t kind; std::vector<PairCol> values; PairCol mult; bool is_send; };

struct Instr { uint8_t op, dst, a, b; uint32_t imm; };

struct Chip {
  std::string name;
  size_t main_width, prep_width;
  int prep_index;
  int lqd;
  bool local_only, global_scope;
  size_t num_constraints;
  std::vector<Lookup> lookups;  // sends then receives
  std::vector<Instr> program;
  size_t n_regs = 0, n_base_regs = 0;
  size_t batch() const { return (size_t)1 << lqd; }
  size_t perm_ext_width() const { return lookups.empty() ? 0 : (lookups.size() + batch() - 1) / batch() + 1; }
};

// Parse a zkm_chip_desc (Montgomery words) into canonical form.
static inline Chip parse_chip(const zkm_chip_desc& d) {
  Chip c;
  c.name = d.name;
  c.main_width = d.main_width; c.prep_width = d.prep_width; c.prep_index = d.prep_index;
  c.lqd = d.log_quotient_degree; c.local_only = d.local_only; c.global_scope = d.commit_scope_global;
  c.num_constraints = d.num_constraints;
  const uint32_t* w = d.lookups; size_t pos = 0;
  if (d.lookups_len) {
    uint32_t ns = w[pos++], nr = w[pos++];
    for (uint32_t i = 0; i < ns + nr; i++) {
      Lookup l; l.is_send = i < ns;
      l.kind = w[pos++];
      uint32_t nv = w[pos++];
      for (uint32_t v = 0; v <= nv; v++) {
        PairCol pc;
        uint32_t nt = w[pos++];
        pc.constant = from_monty(w[pos++]);
        for (uint32_t t = 0; t < nt; t++) { uint32_t col = w[pos++]; pc.terms.push_back({col, from_monty(w[pos++])}); }
        if (v < nv) l.values.push_back(pc); else l.mult = pc;
      }
      c.lookups.push_back(l);
    }
    if (pos != d.lookups_len) throw std::runtime_error("lookup blob length mismatch");
  }
  if (d.program_len) {
    const uint32_t* p = d.program;
    uint32_t ni = p[0]; c.n_regs = p[1]; c.n_base_regs = p[3];
    if (p[2] != d.num_constraints) throw std::runtime_error("program constraint count mismatch");
    if (d.program_len != 4 + 2 * (size_t)ni) throw std::runtime_error("program blob length mismatch");
    for (uint32_t i = 0; i < ni; i++) {
      uint32_t w0 = p[4 + 2 * i], w1 = p[5 + 2 * i];
      c.program.push_back(Instr{(uint8_t)(w0 & 0xff), (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24), w1});
    }
  }
  return c;
}

// Inputs of one constraint evaluation; everything is embedded in EF so the same interpreter
// serves the prover (base-field rows on the quotient coset) and the verifier (openings at zeta).
struct EvalInputs {
  const E* main[2]; const E* prep[2]; const E* perm[2];
  const F* public_values;
  E challenges[2];
  E local_sum;
  F global_sum[14];
  E is_first, is_last, is_trans;
  E alpha;
};

// Horner fold as in VerifierConstraintFolder; equal to the prover's sum_k alpha^(C-1-k) c_k
// (prover.rs:453-456, folder.rs:79-84).
static inline E eval_constraints(const Chip& chip, const EvalInputs& in) {
  std::vector<E> re(chip.n_regs ? chip.n_regs : 1), rb(chip.n_base_regs ? chip.n_base_regs : 1);
  E acc = ezero();
  size_t count = 0;
  for (const Instr& i : chip.program) {
    switch (i.op) {
      case ZKM_OP_LD_MAIN: rb[i.dst] = in.main[i.a][i.imm]; break;
      case ZKM_OP_LD_PREP: rb[i.dst] = in.prep[i.a][i.imm]; break;
      case ZKM_OP_LD_PERM: re[i.dst] = in.perm[i.a][i.imm]; break;
      case ZKM_OP_LD_CONST: rb[i.dst] = efrom(from_monty(i.imm)); break;
      case ZKM_OP_LD_PV: rb[i.dst] = efrom(in.public_values[i.imm]); break;
      case ZKM_OP_LD_CHALLENGE: re[i.dst] = in.challenges[i.imm]; break;
      case ZKM_OP_LD_LOCAL_SUM: re[i.dst] = in.local_sum; break;
      case ZKM_OP_LD_GLOBAL_SUM: rb[i.dst] = efrom(in.global_sum[i.imm]); break;
      case ZKM_OP_LD_IS_FIRST: rb[i.dst] = in.is_first; break;
      case ZKM_OP_LD_IS_LAST: rb[i.dst] = in.is_last; break;
      case ZKM_OP_LD_IS_TRANS: rb[i.dst] = in.is_trans; break;
      case ZKM_OP_ADD_B: rb[i.dst] = eadd(rb[i.a], rb[i.b]); break;
      case ZKM_OP_SUB_B: rb[i.dst] = esub(rb[i.a], rb[i.b]); break;
      case ZKM_OP_MUL_B: rb[i.dst] = emul(rb[i.a], rb[i.b]); break;
      case ZKM_OP_NEG_B: rb[i.dst] = eneg(rb[i.a]); break;
      case ZKM_OP_ADD_E: re[i.dst] = eadd(re[i.a], re[i.b]); break;
      case ZKM_OP_SUB_E: re[i.dst] = esub(re[i.a], re[i.b]); break;
      case ZKM_OP_MUL_E: re[i.dst] = emul(re[i.a], re[i.b]); break;
      case ZKM_OP_NEG_E: re[i.dst] = eneg(re[i.a]); break;
      case ZKM_OP_ADD_EB: re[i.dst] = eadd(re[i.a], rb[i.b]); break;
      case ZKM_OP_SUB_EB: re[i.dst] = esub(re[i.a], rb[i.b]); break;
      case ZKM_OP_MUL_EB: re[i.dst] = emul(re[i.a], rb[i.b]); break;
      case ZKM_OP_ASSERT_B: acc = eadd(emul(acc, in.alpha), rb[i.a]); count++; break;
      case ZKM_OP_ASSERT_E: acc = eadd(emul(acc, in.alpha), re[i.a]); count++; break;
      default: throw std::runtime_error("bad opcode");
    }
  }
  if (count != chip.num_constraints) throw std::runtime_error("constraint count mismatch");
  return acc;
}

// ---- keys, proof containers ------------------------------------------------------------------
struct ProvingKey {
  std::vector<Matrix> prep_traces;
  std::vector<bool> prep_local_only;
  PcsData data;  // empty when there are no preprocessed traces
  bool has_prep = false;
  Digest commit;
  F pc_start;
  F initial_global_cumulative_sum[14];
  void observe_into(Challenger& ch) const {  // machine.rs:79-86
    ch.observe_digest(commit);
    ch.observe(pc_start);
    ch.observe_slice(initial_global_cumulative_sum, 14);
    ch.observe(0);
  }
};

struct ChipOpenedValues {
  std::vector<E> prep_local, prep_next, main_local, main_next, perm_local, perm_next;  // perm: base-column openings (width*4)
  std::vector<std::vector<E>> quotient;  // [chunk][4]
  F global_sum[14];
  E local_sum;
  uint32_t log_degree;
};
struct ShardProof {
  Digest main_commit, perm_commit, quotient_commit;
  std::vector<ChipOpenedValues> chips;
  FriProof fri;
  std::vector<F> public_values;
  std::vector<size_t> order;  // sorted position -> caller index
};

static const uint32_t SEPTIC_START_X[7] = {637514027, 1595065213, 1998064738, 72333738, 1211544370, 822986770, 1518535784};
static const uint32_t SEPTIC_START_Y[7] = {1604177449, 90440090, 259343427, 140470264, 1162099742, 941559812, 1064053343};

// commit ordering: (Reverse(height), name)  prover.rs:264
static inline std::vector<size_t> chip_order(const std::vector<std::string>& names, const std::vector<size_t>& heights) {
  std::vector<size_t> o(names.size());
  std::iota(o.begin(), o.end(), 0);
  std::sort(o.begin(), o.end(), [&](size_t a, size_t b) {
    if (heights[a] != heights[b]) return heights[a] > heights[b];
    return names[a] < names[b];
  });
  return o;
}

// permutation.rs:102-196. Returns n x (perm_ext_width*4) base matrix (flatten_to_base) and the cumulative sum.
static inline Matrix generate_permutation_trace(const Chip& chip, const Matrix* prep, const Matrix& main,
                                                const E& alpha, const E& beta, E& local_sum) {
  size_t n = main.h, pw = chip.perm_ext_width();
  Matrix out(n, pw * 4);
  local_sum = ezero();
  if (pw == 0) return out;
  size_t bs = chip.batch();
  std::vector<E> rowsum(n);
#pragma omp parallel for
  for (size_t r = 0; r < n; r++) {
    const F* prow = prep ? prep->row(r) : nullptr;
    const F* mrow = main.row(r);
    E sum = ezero();
    for (size_t b = 0; b * bs < chip.lookups.size(); b++) {
      E val = ezero();
      for (size_t k = b * bs; k < std::min((b + 1) * bs, chip.lookups.size()); k++) {
        const Lookup& l = chip.lookups[k];
        E denom = eadd(alpha, efrom((F)l.kind));  // beta^0 * argument_index
        E bp = beta;
        for (auto& v : l.values) { denom = eadd(denom, escale(bp, v.apply_f(prow, mrow))); bp = emul(bp, beta); }
        F mult = l.mult.apply_f(prow, mrow);
        if (!l.is_send) mult = fneg(mult);
        val = eadd(val, escale(einv(denom), mult));
      }
      for (int c = 0; c < 4; c++) out.at(r, b * 4 + c) = val.c[c];
      sum = eadd(sum, val);
    }
    rowsum[r] = sum;
  }
  E run = ezero();
  for (size_t r = 0; r < n; r++) {
    run = eadd(run, rowsum[r]);
    for (int c = 0; c < 4; c++) out.at(r, (pw - 1) * 4 + c) = run.c[c];
  }
  local_sum = run;
  return out;
}

struct Selectors { E is_first, is_last, is_trans, inv_zerofier; };
// domain.rs:46-64 for the (unshifted) trace domain H_n at point x
static inline Selectors selectors_at(int log_n, const E& x) {
  E zh = esub(epow2k(x, log_n), eone());
  F ginv = finv(two_adic_generator(log_n));
  Selectors s;
  s.is_first = ediv(zh, esub(x, eone()));
  s.is_last = ediv(zh, esub(x, efrom(ginv)));
  s.is_trans = esub(x, efrom(ginv));
  s.inv_zerofier = einv(zh);
  return s;
}

// quotient.rs:19-171: returns Q = n << lqd extension values, natural order on 3 * <w_Q>.
static inline std::vector<E> quotient_values(const Chip& chip, int log_n, const Matrix* prep_lde, const Matrix& main_lde,
                                             const Matrix& perm_lde, const E& alpha, const E perm_ch[2],
                                             const E& local_sum, const F global_sum[14], const F* pv) {
  int lq = log_n + chip.lqd;
  size_t Q = (size_t)1 << lq;
  size_t step = (size_t)1 << chip.lqd;
  std::vector<E> out(Q);
  size_t pw = chip.perm_ext_width();
  F wq = two_adic_generator(lq);
#pragma omp parallel for
  for (size_t i = 0; i < Q; i++) {
    size_t rows[2] = {bitrev((uint32_t)i, lq), bitrev((uint32_t)((i + step) % Q), lq)};
    std::vector<E> mainv[2], prepv[2], permv[2];
    for (int k = 0; k < 2; k++) {
      mainv[k].resize(chip.main_width);
      for (size_t c = 0; c < chip.main_width; c++) mainv[k][c] = efrom(main_lde.at(rows[k], c));
      prepv[k].resize(chip.prep_width);
      for (size_t c = 0; c < chip.prep_width; c++) prepv[k][c] = efrom(prep_lde->at(rows[k], c));
      permv[k].resize(pw);
      for (size_t c = 0; c < pw; c++) for (int e = 0; e < 4; e++) permv[k][c].c[e] = perm_lde.at(rows[k], 4 * c + e);
    }
    E x = efrom(fmul(GENERATOR, fpow(wq, i)));
    Selectors s = selectors_at(log_n, x);
    EvalInputs in;
    for (int k = 0; k < 2; k++) { in.main[k] = mainv[k].data(); in.prep[k] = prepv[k].data(); in.perm[k] = permv[k].data(); }
    in.public_values = pv;
    in.challenges[0] = perm_ch[0]; in.challenges[1] = perm_ch[1];
    in.local_sum = local_sum;
    memcpy(in.global_sum, global_sum, sizeof in.global_sum);
    in.is_first = s.is_first; in.is_last = s.is_last; in.is_trans = s.is_trans;
    in.alpha = alpha;
    out[i] = emul(eval_constraints(chip, in), s.inv_zerofier);
  }
  return out;
}

struct MainData {
  std::vector<size_t> order;
  std::vector<Matrix> traces;  // sorted order
  PcsData data;
  std::vector<F> public_values;
};

static inline MainData shard_commit(const std::vector<std::string>& names, const std::vector<Matrix>& traces,
                                    const std::vector<F>& pv, int log_blowup) {
  MainData md;
  std::vector<size_t> heights;
  for (auto& t : traces) heights.push_back(t.h);
  md.order = chip_order(names, heights);
  for (size_t i : md.order) md.traces.push_back(traces[i]);
  md.data = pcs_commit(md.traces, {}, log_blowup);
  md.public_values = pv;
  return md;
}

// prover.rs:298-653. chips_in is in caller order; it is permuted with md.order.
static inline ShardProof shard_open(const ProvingKey& pk, MainData& md, const std::vector<Chip>& chips_in,
                                    const FriConfig& cfg, size_t num_pv_elts, Challenger& ch) {
  std::vector<Chip> chips;
  for (size_t i : md.order) chips.push_back(chips_in[i]);
  size_t nc = chips.size();
  ShardProof proof;
  proof.order = md.order;
  proof.public_values = md.public_values;
  proof.main_commit = md.data.tree.root();
  std::vector<int> logn(nc);
  for (size_t i = 0; i < nc; i++) logn[i] = log2_strict(md.traces[i].h);

  ch.observe_slice(md.public_values.data(), num_pv_elts);
  ch.observe_digest(proof.main_commit);
  E perm_ch[2] = {ch.sample_ext(), ch.sample_ext()};

  std::vector<Matrix> perm_traces(nc);
  std::vector<E> local_sums(nc);
  std::vector<std::array<F, 14>> global_sums(nc);
  for (size_t i = 0; i < nc; i++) {
    const Matrix* prep = chips[i].prep_index >= 0 ? &pk.prep_traces[chips[i].prep_index] : nullptr;
    perm_traces[i] = generate_permutation_trace(chips[i], prep, md.traces[i], perm_ch[0], perm_ch[1], local_sums[i]);
    if (chips[i].global_scope) {
      const Matrix& m = md.traces[i];
      for (int k = 0; k < 14; k++) global_sums[i][k] = m.v[m.h * m.w - 14 + k];
    } else {
      for (int k = 0; k < 7; k++) { global_sums[i][k] = SEPTIC_START_X[k]; global_sums[i][7 + k] = SEPTIC_START_Y[k]; }
    }
  }
  PcsData perm_data = pcs_commit(perm_traces, {}, cfg.log_blowup);
  proof.perm_commit = perm_data.tree.root();
  ch.observe_digest(proof.perm_commit);
  for (size_t i = 0; i < nc; i++) {
    ch.observe_ext(local_sums[i]);
    ch.observe_slice(global_sums[i].data(), 14);
  }
  E alpha = ch.sample_ext();

  std::vector<Matrix> qchunks;
  std::vector<F> qshifts;
  for (size_t i = 0; i < nc; i++) {
    if (chips[i].lqd > cfg.log_blowup) throw std::runtime_error("log_quotient_degree > log_blowup unsupported");
    const Matrix* prep_lde = chips[i].prep_index >= 0 ? &pk.data.tree.leaves[chips[i].prep_index] : nullptr;
    std::vector<E> q = quotient_values(chips[i], logn[i], prep_lde, md.data.tree.leaves[i], perm_data.tree.leaves[i],
                                       alpha, perm_ch, local_sums[i], global_sums[i].data(), md.public_values.data());
    size_t nchunks = chips[i].batch();
    size_t n = md.traces[i].h;
    F wq = two_adic_generator(logn[i] + chips[i].lqd);
    for (size_t c = 0; c < nchunks; c++) {
      Matrix m(n, 4);
      for (size_t r = 0; r < n; r++) for (int e = 0; e < 4; e++) m.at(r, e) = q[r * nchunks + c].c[e];
      qchunks.push_back(std::move(m));
      qshifts.push_back(fmul(GENERATOR, fpow(wq, c)));
    }
  }
  PcsData quot_data = pcs_commit(qchunks, qshifts, cfg.log_blowup);
  proof.quotient_commit = quot_data.tree.root();
  ch.observe_digest(proof.quotient_commit);
  E zeta = ch.sample_ext();

  std::vector<OpenRound> rounds;
  if (pk.has_prep) {
    OpenRound r; r.data = &pk.data;
    for (size_t j = 0; j < pk.prep_traces.size(); j++) {
      F g = two_adic_generator(log2_strict(pk.prep_traces[j].h));
      if (!pk.prep_local_only[j]) r.points.push_back({zeta, escale(zeta, g)}); else r.points.push_back({zeta});
    }
    rounds.push_back(r);
  }
  {
    OpenRound r; r.data = &md.data;
    for (size_t i = 0; i < nc; i++) {
      F g = two_adic_generator(logn[i]);
      if (!chips[i].local_only) r.points.push_back({zeta, escale(zeta, g)}); else r.points.push_back({zeta});
    }
    rounds.push_back(r);
  }
  {
    OpenRound r; r.data = &perm_data;
    for (size_t i = 0; i < nc; i++) r.points.push_back({zeta, escale(zeta, two_adic_generator(logn[i]))});
    rounds.push_back(r);
  }
  {
    OpenRound r; r.data = &quot_data;
    for (size_t i = 0; i < qchunks.size(); i++) r.points.push_back({zeta});
    rounds.push_back(r);
  }
  OpenedValues ov;
  pcs_open(rounds, cfg, ch, ov, proof.fri);

  size_t ri = 0;
  const std::vector<std::vector<std::vector<E>>>* prep_ov = pk.has_prep ? &ov[ri++] : nullptr;
  auto& main_ov = ov[ri++]; auto& perm_ov = ov[ri++]; auto& quot_ov = ov[ri++];
  size_t qpos = 0;
  for (size_t i = 0; i < nc; i++) {
    ChipOpenedValues c;
    if (chips[i].prep_index >= 0) {
      auto& p = (*prep_ov)[chips[i].prep_index];
      c.prep_local = p[0];
      c.prep_next = p.size() > 1 ? p[1] : std::vector<E>(p[0].size(), ezero());
    }
    c.main_local = main_ov[i][0];
    c.main_next = main_ov[i].size() > 1 ? main_ov[i][1] : std::vector<E>(c.main_local.size(), ezero());
    c.perm_local = perm_ov[i][0]; c.perm_next = perm_ov[i][1];
    for (size_t k = 0; k < chips[i].batch(); k++) c.quotient.push_back(quot_ov[qpos++][0]);
    memcpy(c.global_sum, global_sums[i].data(), sizeof c.global_sum);
    c.local_sum = local_sums[i];
    c.log_degree = logn[i];
    proof.chips.push_back(std::move(c));
  }
  return proof;
}

// ---- verifier (verifier.rs:30-435) ------------------------------------------------------------
struct VerifyingKey {
  Digest commit; F pc_start; F initial_global_cumulative_sum[14];
  bool has_prep = false;
  std::vector<int> prep_log_heights;     // vk.chip_information domains, pk trace order
  std::vector<std::string> prep_names;
  void observe_into(Challenger& ch) const {
    ch.observe_digest(commit); ch.observe(pc_start); ch.observe_slice(initial_global_cumulative_sum, 14); ch.observe(0);
  }
};

// chips_in: caller order; proof.order maps sorted position -> caller index.
// `ch` must already hold vk.observe_into (machine.rs:630). Returns 0 on accept.
static inline int verify_shard(const VerifyingKey& vk, const std::vector<Chip>& chips_in, const FriConfig& cfg,
                               size_t num_pv_elts, Challenger& ch, const ShardProof& proof) {
  std::vector<Chip> chips;
  for (size_t i : proof.order) chips.push_back(chips_in[i]);
  size_t nc = chips.size();
  if (proof.chips.size() != nc) return 1;
  ch.observe_slice(proof.public_values.data(), num_pv_elts);  // machine.rs:646-647
  ch.observe_digest(proof.main_commit);
  E perm_ch[2] = {ch.sample_ext(), ch.sample_ext()};
  ch.observe_digest(proof.perm_commit);
  for (size_t i = 0; i < nc; i++) {
    const ChipOpenedValues& o = proof.chips[i];
    ch.observe_ext(o.local_sum);
    ch.observe_slice(o.global_sum, 14);
    bool gzero = true;
    for (int k = 0; k < 7; k++) gzero = gzero && o.global_sum[k] == SEPTIC_START_X[k] && o.global_sum[7 + k] == SEPTIC_START_Y[k];
    if (!chips[i].global_scope && !gzero) return 2;
    if (chips[i].lookups.empty() && !eis_zero(o.local_sum)) return 3;
  }
  E alpha = ch.sample_ext();
  ch.observe_digest(proof.quotient_commit);
  E zeta = ch.sample_ext();

  std::vector<VerifyRound> rounds;
  if (vk.has_prep) {
    VerifyRound r; r.commit = vk.commit;
    for (size_t j = 0; j < vk.prep_log_heights.size(); j++) {
      // locate the chip that owns preprocessed trace j
      size_t ci = nc;
      for (size_t i = 0; i < nc; i++) if (chips[i].prep_index == (int)j) ci = i;
      if (ci == nc) return 4;
      VerifyMat m; m.log_height = vk.prep_log_heights[j]; m.domain_shift = 1;
      m.points.push_back(zeta); m.values.push_back(proof.chips[ci].prep_local);
      if (!chips[ci].local_only) {
        m.points.push_back(escale(zeta, two_adic_generator(m.log_height)));
        m.values.push_back(proof.chips[ci].prep_next);
      }
      r.mats.push_back(m);
    }
    rounds.push_back(r);
  }
  VerifyRound rm, rp, rq;
  rm.commit = proof.main_commit; rp.commit = proof.perm_commit; rq.commit = proof.quotient_commit;
  for (size_t i = 0; i < nc; i++) {
    const ChipOpenedValues& o = proof.chips[i];
    int ln = o.log_degree;
    E zg = escale(zeta, two_adic_generator(ln));
    VerifyMat m; m.log_height = ln; m.domain_shift = 1;
    m.points.push_back(zeta); m.values.push_back(o.main_local);
    if (!chips[i].local_only) { m.points.push_back(zg); m.values.push_back(o.main_next); }
    rm.mats.push_back(m);
    VerifyMat p; p.log_height = ln; p.domain_shift = 1;
    p.points = {zeta, zg}; p.values = {o.perm_local, o.perm_next};
    rp.mats.push_back(p);
    if (o.quotient.size() != chips[i].batch()) return 5;
    for (size_t c = 0; c < o.quotient.size(); c++) {
      VerifyMat q; q.log_height = ln;
      q.domain_shift = fmul(GENERATOR, fpow(two_adic_generator(ln + chips[i].lqd), c));
      q.points = {zeta}; q.values = {o.quotient[c]};
      rq.mats.push_back(q);
    }
  }
  rounds.push_back(rm); rounds.push_back(rp); rounds.push_back(rq);
  int rc = pcs_verify(rounds, cfg, proof.fri, ch);
  if (rc) return rc;

  E total_local = ezero();
  for (size_t i = 0; i < nc; i++) {
    const Chip& chip = chips[i];
    const ChipOpenedValues& o = proof.chips[i];
    // verify_opening_shape (verifier.rs:248-312)
    if (o.prep_local.size() != chip.prep_width || o.prep_next.size() != chip.prep_width) return 20;
    if (o.main_local.size() != chip.main_width || o.main_next.size() != chip.main_width) return 21;
    if (o.perm_local.size() != chip.perm_ext_width() * 4 || o.perm_next.size() != chip.perm_ext_width() * 4) return 22;
    for (auto& q : o.quotient) if (q.size() != 4) return 23;
    int ln = o.log_degree;
    Selectors s = selectors_at(ln, zeta);
    // recompute_quotient (verifier.rs:400-435)
    size_t nch = chip.batch();
    std::vector<F> shifts(nch);
    for (size_t c = 0; c < nch; c++) shifts[c] = fmul(GENERATOR, fpow(two_adic_generator(ln + chip.lqd), c));
    auto zp_at = [&](F shift, const E& pt) { return esub(epow2k(escale(pt, finv(shift)), ln), eone()); };
    E quotient = ezero();
    for (size_t c = 0; c < nch; c++) {
      E zps = eone();
      for (size_t j = 0; j < nch; j++) if (j != c)
        zps = emul(zps, emul(zp_at(shifts[j], zeta), einv(zp_at(shifts[j], efrom(shifts[c])))));
      for (int e = 0; e < 4; e++) {
        E mono = ezero(); mono.c[e] = 1;
        quotient = eadd(quotient, emul(emul(zps, mono), o.quotient[c][e]));
      }
    }
    // eval_constraints (verifier.rs:352-398)
    std::vector<E> permv[2];
    for (int k = 0; k < 2; k++) {
      const std::vector<E>& src = k ? o.perm_next : o.perm_local;
      permv[k].assign(chip.perm_ext_width(), ezero());
      for (size_t c = 0; c < chip.perm_ext_width(); c++)
        for (int e = 0; e < 4; e++) { E mono = ezero(); mono.c[e] = 1; permv[k][c] = eadd(permv[k][c], emul(mono, src[4 * c + e])); }
    }
    EvalInputs in;
    in.main[0] = o.main_local.data(); in.main[1] = o.main_next.data();
    in.prep[0] = o.prep_local.data(); in.prep[1] = o.prep_next.data();
    in.perm[0] = permv[0].data(); in.perm[1] = permv[1].data();
    in.public_values = proof.public_values.data();
    in.challenges[0] = perm_ch[0]; in.challenges[1] = perm_ch[1];
    in.local_sum = o.local_sum;
    memcpy(in.global_sum, o.global_sum, sizeof in.global_sum);
    in.is_first = s.is_first; in.is_last = s.is_last; in.is_trans = s.is_trans;
    in.alpha = alpha;
    E folded = eval_constraints(chip, in);
    if (emul(folded, s.inv_zerofier) != quotient) return 30 + (int)i * 0 + 0;
    total_local = eadd(total_local, o.local_sum);
  }
  if (!eis_zero(total_local)) return 40;
  return 0;
}

}  // namespace orc
