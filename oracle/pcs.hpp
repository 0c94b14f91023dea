// TEST INFRASTRUCTURE — CPU oracle. Never linked, imported or executed by the product path.
//
// Two-adic FRI PCS over KoalaBear with a mixed-height Poseidon2 Merkle MMCS, restated on the
// CPU. The arithmetic of this layer lives in the un-vendored dependency
//     ProjectZKM/Plonky3 @ faa24ca4597eebeecbf71b194b71c7d1a99b3f01   (Cargo.lock:4014-4320)
// (crates p3-dft, p3-merkle-tree, p3-commit, p3-fri); its source is absent from
// /root/reference, so this file restates the published algorithm and anchors every choice on
// the reference's in-tree mirrors of it:
//     MMCS verify_batch                  crates/recursion/circuit/src/fri.rs:363-405
//     reduced openings / x formula       crates/recursion/circuit/src/fri.rs:71-218
//     FRI fold, beta^2 * ro, final poly  crates/recursion/circuit/src/fri.rs:34-69,220-361
//     domains / selectors                crates/recursion/circuit/src/domain.rs:36-88
//     call sites of Pcs::commit/open     crates/stark/src/prover.rs:277,403,497,546-556
// PARITY UNPINNED at the byte level: the reference holds no golden vectors for this layer
// (SURVEY.md F6); it is pinned structurally — proofs produced here are accepted by
// verify_* below, which follow the in-tree verifier line by line.
#pragma once
#include <omp.h>
#include "poseidon2.hpp"
#include <algorithm>
#include <numeric>

namespace orc {

struct Matrix {
  size_t h = 0, w = 0;
  std::vector<F> v;  // row-major
  Matrix() {}
  Matrix(size_t h_, size_t w_) : h(h_), w(w_), v(h_ * w_, 0) {}
  F& at(size_t r, size_t c) { return v[r * w + c]; }
  F at(size_t r, size_t c) const { return v[r * w + c]; }
  const F* row(size_t r) const { return v.data() + r * w; }
};

// ---- textbook radix-2 NTT, natural order in and out ---------------------------------------
static inline void ntt_inplace(std::vector<F>& a, bool inverse) {
  size_t n = a.size();
  int k = log2_strict(n);
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev((uint32_t)i, k);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int s = 1; s <= k; s++) {
    size_t m = (size_t)1 << s;
    F wm = two_adic_generator(s);
    if (inverse) wm = finv(wm);
    for (size_t blk = 0; blk < n; blk += m) {
      F w = 1;
      for (size_t j = 0; j < m / 2; j++) {
        F t = fmul(w, a[blk + j + m / 2]);
        F u = a[blk + j];
        a[blk + j] = fadd(u, t);
        a[blk + j + m / 2] = fsub(u, t);
        w = fmul(w, wm);
      }
    }
  }
  if (inverse) {
    F ninv = finv((F)(n % P));
    for (auto& x : a) x = fmul(x, ninv);
  }
}

// Coefficients of the degree<n interpolant P with P(domain_shift * w_n^i) = evals[i].
static inline std::vector<F> interpolate_coset(std::vector<F> evals, F domain_shift) {
  ntt_inplace(evals, true);  // coefficients of P(domain_shift * x)
  F sinv = finv(domain_shift), p = 1;
  for (auto& c : evals) { c = fmul(c, p); p = fmul(p, sinv); }
  return evals;
}

// Radix2Dit::coset_lde_batch semantics: evals over H_n -> evaluations of the
// interpolant over lde_shift * K_{n << added_bits}, natural order.
static inline std::vector<F> coset_lde(std::vector<F> evals, int added_bits, F lde_shift) {
  size_t n = evals.size();
  ntt_inplace(evals, true);
  evals.resize(n << added_bits, 0);
  F p = 1;
  for (size_t i = 0; i < n; i++) { evals[i] = fmul(evals[i], p); p = fmul(p, lde_shift); }
  ntt_inplace(evals, false);
  return evals;
}

// wall-clock seconds spent in coset LDEs since the last reset (the one n log n phase: bench.py scales it separately)
static inline double& lde_seconds() { static double t = 0; return t; }
static inline Matrix coset_lde_matrix_bitrev(const Matrix& m, int added_bits, F lde_shift) {
  struct Timer { double t0 = omp_get_wtime(); ~Timer() { lde_seconds() += omp_get_wtime() - t0; } } timer;
  size_t H = m.h << added_bits;
  int logH = log2_strict(H);
  Matrix out(H, m.w);
  // columns are extended side by side, each into a vector of its own; the row-major result is then written row by row (a thread owns
  // whole rows: no two threads share a cache line of `out`, which is what kept this loop from scaling past a few threads before)
  std::vector<std::vector<F>> cols(m.w);
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < m.w; c++) {
    std::vector<F> col(m.h);
    for (size_t r = 0; r < m.h; r++) col[r] = m.at(r, c);
    cols[c] = coset_lde(std::move(col), added_bits, lde_shift);
  }
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < H; r++) {
    const size_t j = bitrev((uint32_t)r, logH);
    for (size_t c = 0; c < m.w; c++) out.at(r, c) = cols[c][j];
  }
  return out;
}

// ---- MerkleTreeMmcs ------------------------------------------------------------------------
struct MerkleTree {
  std::vector<Matrix> leaves;               // committed matrices (for PCS: bit-reversed LDEs)
  std::vector<std::vector<Digest>> layers;  // layers[0] has max_height digests, last has 1
  Digest root() const { return layers.back()[0]; }
  size_t max_height() const { return layers[0].size(); }
};

static inline MerkleTree mmcs_commit(std::vector<Matrix> mats) {
  MerkleTree t;
  t.leaves = std::move(mats);
  size_t nm = t.leaves.size();
  std::vector<size_t> order(nm);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](size_t a, size_t b) { return t.leaves[a].h > t.leaves[b].h; });
  size_t maxh = t.leaves[order[0]].h;
  auto hash_rows_of_height = [&](size_t height, size_t r) {
    std::vector<F> buf;
    for (size_t idx : order)
      if (t.leaves[idx].h == height)
        buf.insert(buf.end(), t.leaves[idx].row(r), t.leaves[idx].row(r) + t.leaves[idx].w);
    return hash_slice(buf.data(), buf.size());
  };
  auto has_height = [&](size_t height) {
    for (auto& m : t.leaves) if (m.h == height) return true;
    return false;
  };
  std::vector<Digest> cur(maxh);
#pragma omp parallel for
  for (size_t r = 0; r < maxh; r++) cur[r] = hash_rows_of_height(maxh, r);
  t.layers.push_back(cur);
  while (cur.size() > 1) {
    size_t len = cur.size() / 2;
    std::vector<Digest> next(len);
    bool inject = has_height(len);
#pragma omp parallel for
    for (size_t i = 0; i < len; i++) {
      Digest d = compress(cur[2 * i], cur[2 * i + 1]);
      if (inject) d = compress(d, hash_rows_of_height(len, i));
      next[i] = d;
    }
    t.layers.push_back(next);
    cur.swap(next);
  }
  return t;
}

struct BatchOpening {
  std::vector<std::vector<F>> opened_values;  // one row per matrix, caller order
  std::vector<Digest> proof;                  // siblings bottom-up
};

static inline BatchOpening mmcs_open_batch(const MerkleTree& t, size_t index) {
  BatchOpening o;
  int log_max = log2_strict(t.max_height());
  for (auto& m : t.leaves) {
    size_t r = index >> (log_max - log2_strict(m.h));
    o.opened_values.emplace_back(m.row(r), m.row(r) + m.w);
  }
  for (int l = 0; l < log_max; l++) o.proof.push_back(t.layers[l][(index >> l) ^ 1]);
  return o;
}

// fri.rs:363-405
static inline bool mmcs_verify_batch(const Digest& commit, const std::vector<size_t>& heights,
                                     size_t index, const std::vector<std::vector<F>>& opened,
                                     const std::vector<Digest>& proof) {
  size_t nm = heights.size();
  if (opened.size() != nm) return false;
  std::vector<size_t> order(nm);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return heights[a] > heights[b]; });
  size_t pos = 0;
  size_t cur_h = heights[order[0]];
  if (proof.size() != (size_t)log2_strict(cur_h)) return false;
  std::vector<F> buf;
  while (pos < nm && heights[order[pos]] == cur_h) {
    buf.insert(buf.end(), opened[order[pos]].begin(), opened[order[pos]].end());
    pos++;
  }
  Digest root = hash_slice(buf.data(), buf.size());
  for (size_t l = 0; l < proof.size(); l++) {
    bool bit = (index >> l) & 1;
    root = bit ? compress(proof[l], root) : compress(root, proof[l]);
    cur_h >>= 1;
    if (pos < nm && heights[order[pos]] == cur_h) {
      buf.clear();
      while (pos < nm && heights[order[pos]] == cur_h) {
        buf.insert(buf.end(), opened[order[pos]].begin(), opened[order[pos]].end());
        pos++;
      }
      root = compress(root, hash_slice(buf.data(), buf.size()));
    }
  }
  return pos == nm && root == commit;
}

// ---- TwoAdicFriPcs::commit -----------------------------------------------------------------
struct PcsData {
  std::vector<Matrix> evals;       // original evaluations
  std::vector<F> domain_shifts;    // evals[i] are over domain_shifts[i] * H
  MerkleTree tree;                 // leaves = bit-reversed LDEs on 3 * K
  int log_blowup = 1;
};

static inline PcsData pcs_commit(std::vector<Matrix> mats, std::vector<F> shifts, int log_blowup) {
  PcsData d;
  d.log_blowup = log_blowup;
  if (shifts.empty()) shifts.assign(mats.size(), 1);
  std::vector<Matrix> ldes;
  for (size_t i = 0; i < mats.size(); i++)
    ldes.push_back(coset_lde_matrix_bitrev(mats[i], log_blowup, fmul(GENERATOR, finv(shifts[i]))));
  d.evals = std::move(mats);
  d.domain_shifts = std::move(shifts);
  d.tree = mmcs_commit(std::move(ldes));
  return d;
}

// ---- Pcs::open / FRI prover ----------------------------------------------------------------
struct FriConfig { int log_blowup, num_queries, pow_bits; };

struct CommitPhaseStep { E sibling_value; std::vector<Digest> proof; };
struct QueryProof {
  std::vector<BatchOpening> input_proof;  // one per round
  std::vector<CommitPhaseStep> steps;
};
struct FriProof {
  std::vector<Digest> commit_phase_commits;
  std::vector<QueryProof> queries;
  E final_poly;
  F pow_witness;
};

struct OpenRound {
  const PcsData* data;
  std::vector<std::vector<E>> points;  // per matrix
};
typedef std::vector<std::vector<std::vector<std::vector<E>>>> OpenedValues;  // [round][mat][point][col]

static inline E eval_poly_ext(const std::vector<F>& coeffs, const E& z) {
  E acc = ezero();
  for (size_t i = coeffs.size(); i-- > 0;) acc = eadd(emul(acc, z), efrom(coeffs[i]));
  return acc;
}

static inline Matrix fri_layer_matrix(const std::vector<E>& f) {
  Matrix m(f.size() / 2, 8);
  for (size_t j = 0; j < f.size() / 2; j++)
    for (int c = 0; c < 4; c++) { m.at(j, c) = f[2 * j].c[c]; m.at(j, 4 + c) = f[2 * j + 1].c[c]; }
  return m;
}

static inline void pcs_open(const std::vector<OpenRound>& rounds, const FriConfig& cfg, Challenger& ch,
                            OpenedValues& opened, FriProof& proof) {
  // (i) evaluate every column at every point (Horner on interpolated coefficients).
  opened.assign(rounds.size(), {});
  for (size_t r = 0; r < rounds.size(); r++) {
    const PcsData& d = *rounds[r].data;
    opened[r].resize(d.evals.size());
    for (size_t m = 0; m < d.evals.size(); m++) {
      const Matrix& M = d.evals[m];
      size_t np = rounds[r].points[m].size();
      opened[r][m].assign(np, std::vector<E>(M.w));
#pragma omp parallel for schedule(dynamic)
      for (size_t c = 0; c < M.w; c++) {
        std::vector<F> col(M.h);
        for (size_t i = 0; i < M.h; i++) col[i] = M.at(i, c);
        std::vector<F> coeffs = interpolate_coset(std::move(col), d.domain_shifts[m]);
        for (size_t p = 0; p < np; p++) opened[r][m][p][c] = eval_poly_ext(coeffs, rounds[r].points[m][p]);
      }
    }
  }
  // (ii) alpha — the opened values are not observed (fri.rs:78; recursion/circuit/src/stark.rs:325->432)
  E alpha = ch.sample_ext();
  // (iii) reduced openings per LDE log-height
  std::vector<std::vector<E>> ro(32);
  E alpha_pow[32];
  for (int i = 0; i < 32; i++) alpha_pow[i] = eone();
  int log_max = 0;
  for (size_t r = 0; r < rounds.size(); r++) {
    const PcsData& d = *rounds[r].data;
    for (size_t m = 0; m < d.evals.size(); m++) {
      const Matrix& L = d.tree.leaves[m];
      int lh = log2_strict(L.h);
      log_max = std::max(log_max, lh);
      if (ro[lh].empty()) ro[lh].assign(L.h, ezero());
      F wh = two_adic_generator(lh);
      for (size_t p = 0; p < rounds[r].points[m].size(); p++) {
        const E z = rounds[r].points[m][p];
        const std::vector<E>& ys = opened[r][m][p];
        std::vector<E> apows(L.w);
        for (size_t c = 0; c < L.w; c++) { apows[c] = alpha_pow[lh]; alpha_pow[lh] = emul(alpha_pow[lh], alpha); }
#pragma omp parallel for
        for (size_t row = 0; row < L.h; row++) {
          F x = fmul(GENERATOR, fpow(wh, bitrev((uint32_t)row, lh)));
          E acc = ezero();
          for (size_t c = 0; c < L.w; c++)
            acc = eadd(acc, emul(apows[c], esub(ys[c], efrom(L.at(row, c)))));
          E denom = esub(z, efrom(x));
          ro[lh][row] = eadd(ro[lh][row], ediv(acc, denom));
        }
      }
    }
  }
  // (iv) FRI commit phase (fri.rs:257-358 mirrored)
  std::vector<E> f = ro[log_max];
  std::vector<MerkleTree> trees;
  std::vector<std::vector<E>> layers;
  size_t blowup = (size_t)1 << cfg.log_blowup;
  while (f.size() > blowup) {
    int lf = log2_strict(f.size());
    MerkleTree t = mmcs_commit({fri_layer_matrix(f)});
    ch.observe_digest(t.root());
    proof.commit_phase_commits.push_back(t.root());
    E beta = ch.sample_ext();
    std::vector<E> g(f.size() / 2);
    F wl = two_adic_generator(lf);
    F inv2 = finv(2);
#pragma omp parallel for
    for (size_t j = 0; j < g.size(); j++) {
      F x = fpow(wl, bitrev((uint32_t)(2 * j), lf));
      // e0 + (beta - x)(e1 - e0)/(-2x)
      E t1 = esub(f[2 * j + 1], f[2 * j]);
      E t2 = esub(beta, efrom(x));
      F dinv = fneg(fmul(inv2, finv(x)));
      g[j] = eadd(f[2 * j], escale(emul(t2, t1), dinv));
    }
    if (!ro[lf - 1].empty()) {
      E b2 = emul(beta, beta);
      for (size_t j = 0; j < g.size(); j++) g[j] = eadd(g[j], emul(b2, ro[lf - 1][j]));
    }
    trees.push_back(std::move(t));
    layers.push_back(f);
    f.swap(g);
  }
  for (size_t i = 1; i < f.size(); i++) assert(f[i] == f[0]);
  proof.final_poly = f[0];
  ch.observe_ext(proof.final_poly);
  proof.pow_witness = ch.grind(cfg.pow_bits);
  for (int q = 0; q < cfg.num_queries; q++) {
    size_t index = ch.sample_bits(log_max);
    QueryProof qp;
    for (auto& rd : rounds) {
      int lr = log2_strict(rd.data->tree.max_height());
      qp.input_proof.push_back(mmcs_open_batch(rd.data->tree, index >> (log_max - lr)));
    }
    for (size_t t = 0; t < trees.size(); t++) {
      CommitPhaseStep st;
      size_t i = index >> t;
      st.sibling_value = layers[t][i ^ 1];
      st.proof = mmcs_open_batch(trees[t], i >> 1).proof;
      qp.steps.push_back(st);
    }
    proof.queries.push_back(std::move(qp));
  }
}

// ---- Pcs::verify (fri.rs:34-361) -----------------------------------------------------------
struct VerifyMat { int log_height; F domain_shift; std::vector<E> points; std::vector<std::vector<E>> values; };
struct VerifyRound { Digest commit; std::vector<VerifyMat> mats; };

// returns 0 on accept, otherwise a distinct error code
static inline int pcs_verify(const std::vector<VerifyRound>& rounds, const FriConfig& cfg,
                             const FriProof& proof, Challenger& ch) {
  E alpha = ch.sample_ext();
  std::vector<E> betas;
  for (auto& c : proof.commit_phase_commits) { ch.observe_digest(c); betas.push_back(ch.sample_ext()); }
  ch.observe_ext(proof.final_poly);
  if ((int)proof.queries.size() != cfg.num_queries) return 10;
  if (!ch.check_witness(cfg.pow_bits, proof.pow_witness)) return 11;
  int log_max = (int)proof.commit_phase_commits.size() + cfg.log_blowup;
  for (auto& qp : proof.queries) {
    size_t index = ch.sample_bits(log_max);
    E ro[32]; int pow_cnt[32];
    std::vector<E> alpha_pows{eone()};
    for (int i = 0; i < 32; i++) { ro[i] = ezero(); pow_cnt[i] = 0; }
    if (qp.input_proof.size() != rounds.size()) return 12;
    for (size_t r = 0; r < rounds.size(); r++) {
      const BatchOpening& bo = qp.input_proof[r];
      std::vector<size_t> dims;
      int lbm = 0;
      for (auto& m : rounds[r].mats) { dims.push_back((size_t)1 << (m.log_height + cfg.log_blowup)); lbm = std::max(lbm, m.log_height + cfg.log_blowup); }
      size_t red = index >> (log_max - lbm);
      if (!mmcs_verify_batch(rounds[r].commit, dims, red, bo.opened_values, bo.proof)) return 13;
      for (size_t m = 0; m < rounds[r].mats.size(); m++) {
        const VerifyMat& vm = rounds[r].mats[m];
        int lh = vm.log_height + cfg.log_blowup;
        size_t idx = (index >> (log_max - lh));
        F x = fmul(GENERATOR, fpow(two_adic_generator(lh), bitrev((uint32_t)idx, lh)));
        for (size_t p = 0; p < vm.points.size(); p++) {
          if (vm.values[p].size() != bo.opened_values[m].size()) return 14;
          E acc = ezero();
          for (size_t c = 0; c < vm.values[p].size(); c++) {
            while ((int)alpha_pows.size() <= pow_cnt[lh]) alpha_pows.push_back(emul(alpha_pows.back(), alpha));
            acc = eadd(acc, emul(alpha_pows[pow_cnt[lh]], esub(vm.values[p][c], efrom(bo.opened_values[m][c]))));
            pow_cnt[lh]++;
          }
          ro[lh] = eadd(ro[lh], ediv(acc, esub(vm.points[p], efrom(x))));
        }
      }
    }
    if (!eis_zero(ro[cfg.log_blowup])) return 15;
    // verify_query
    if (qp.steps.size() != proof.commit_phase_commits.size()) return 16;
    E folded = ro[log_max];
    F x = fpow(two_adic_generator(log_max), bitrev((uint32_t)index, log_max));
    for (size_t t = 0; t < qp.steps.size(); t++) {
      int lfh = log_max - 1 - (int)t;
      bool bit = (index >> t) & 1;
      size_t pair = index >> (t + 1);
      E e0 = bit ? qp.steps[t].sibling_value : folded;
      E e1 = bit ? folded : qp.steps[t].sibling_value;
      std::vector<F> row(8);
      for (int c = 0; c < 4; c++) { row[c] = e0.c[c]; row[4 + c] = e1.c[c]; }
      if (!mmcs_verify_batch(proof.commit_phase_commits[t], {(size_t)1 << lfh}, pair, {row}, qp.steps[t].proof)) return 17;
      F xs_new = fmul(x, two_adic_generator(1));
      F x0 = bit ? xs_new : x, x1 = bit ? x : xs_new;
      E t3 = escale(esub(e1, e0), finv(fsub(x1, x0)));
      folded = eadd(e0, emul(esub(betas[t], efrom(x0)), t3));
      folded = eadd(folded, emul(emul(betas[t], betas[t]), ro[lfh]));
      x = fmul(x, x);
    }
    if (folded != proof.final_poly) return 18;
  }
  return 0;
}

}  // namespace orc
