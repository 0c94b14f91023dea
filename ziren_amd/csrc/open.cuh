// Pcs::open on gfx950: out-of-domain evaluation, reduced openings, FRI folding, query gathers.
//
// Replaces p3-fri TwoAdicFriPcs::open + p3-fri prover::prove as called at
// crates/stark/src/prover.rs:546-556; semantics mirrored in-tree by
// crates/recursion/circuit/src/fri.rs:71-218 (reduced openings) and :220-361 (fold, queries).
#pragma once
#include "kb31.cuh"
#include "gptr.cuh"

namespace open {

constexpr int THREADS = 256;

// Barycentric weights for evaluating, at the extension point z, a polynomial of degree < n
// given by its values on the coset s * H_n in natural order:
//   p(z) = sum_i v_i * w_i,   w_i = (u^n - 1)/n * w^i / (u - w^i),   u = z / s.
// `c` = (u^n - 1)/n is computed on the host. p(z * g) = sum_i v_{i+1} w_i (rotation), so one
// weight vector serves both opening points of a trace.
// Four consecutive points per thread: one power of w_n, one extension inverse for the four denominators
// (Montgomery's trick: 9 products instead of 3 more inverses). grid = ceil(n / 4 / THREADS).
// The opening phase of a shard evaluates ~4 matrices per chip: one launch per kernel covers all of them (round 3; before, each matrix
// had its own bary_weights / eval_columns / reduce_partials launches: ~80 dispatches per proof). A flat blockIdx.x is mapped to
// (job, block inside the job) through the jobs' cumulative block counts; the job tables live in device memory, indices are wave-uniform.
// Column tables (round 5): the opening kernels address a matrix's columns through a pointer and a row mask per column. A column that
// never changes (lde::Mat::cflag, kept with the commitment) gets mask 0 and a pointer at eight copies of its canonical word, so every row
// (and every 16-byte load) of it reads one cache line instead of the column: reduce_openings and eval_columns_batch run at the HBM read
// ceiling, and a fifth to a quarter of the benchmarked shard's columns are constant. Any other column: the column itself, mask ~0.
struct ColJob { const uint32_t* lde; const uint32_t* evals; const uint32_t* cflag; size_t N, n; uint32_t col0, width; };
__global__ __launch_bounds__(256) void build_col_tables(const ColJob* __restrict__ jobs, int n_jobs, uint32_t total, const uint32_t** __restrict__ lde_ptr,
                                                        const uint32_t** __restrict__ eval_ptr, uint32_t* __restrict__ mask, uint32_t* __restrict__ cword) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int j = 0;
  while (j + 1 < n_jobs && t >= jobs[j + 1].col0) j++;
  const ColJob J = jobs[j];
  const uint32_t c = t - J.col0;
  if (J.cflag && J.cflag[2 * c] == 0) {
    const uint32_t w0 = J.cflag[2 * c + 1];
    const uint32_t w = kb::umin32(w0, w0 - kb::P);   // what lde_cols<true> wrote into the LDE; the evaluations hold a word congruent to it
#pragma unroll
    for (int k = 0; k < 8; k++) cword[8 * (size_t)t + k] = w;
    lde_ptr[t] = cword + 8 * (size_t)t;
    eval_ptr[t] = cword + 8 * (size_t)t;
    mask[t] = 0;
  } else {
    lde_ptr[t] = J.lde + (size_t)c * J.N;
    eval_ptr[t] = J.evals + (size_t)c * J.n;
    mask[t] = 0xffffffffu;
  }
}

struct WeightJob { kb::E4 u, c; uint32_t w_n, blk_end; size_t n; kb::E4* out; };
struct EvalJob {
  const uint32_t* mat; const kb::E4* weights; kb::E4* partials;
  const uint32_t* const* colptr; const uint32_t* colmask;   // the matrix's window of the column tables
  size_t n;
  int width, groups, split, kind;   // kind 0: one point, 1: two points, 2: small one point, 3: small two points
  uint32_t blk_end;
  uint32_t lead;                    // XCD_AWARE | leading blocks of the job's range that do nothing (they align its first block to a multiple of 8)
};
constexpr uint32_t EVAL_XCD_AWARE = 0x100;
struct SumJob { const kb::E4* partials; int split, count; uint32_t out0, blk_end; };
template <class J>
__device__ __forceinline__ const J& find_job(const J* __restrict__ jobs, uint32_t& local) {
  uint32_t j = 0, start = 0;
  const uint32_t b = blockIdx.x;
  while (b >= jobs[j].blk_end) { start = jobs[j].blk_end; j++; }
  local = b - start;
  return jobs[j];
}

__device__ __forceinline__ void bary_weights_body(kb::E4 u, kb::E4 c, uint32_t w_n, size_t n, kb::E4* __restrict__ out, uint32_t bx) {
  const size_t i0 = ((size_t)bx * blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  uint32_t wi[4];
  wi[0] = kb::pow(w_n, (uint64_t)i0);
#pragma unroll
  for (int k = 1; k < 4; k++) wi[k] = kb::mul(wi[k - 1], w_n);
  kb::E4 d[4];
#pragma unroll
  for (int k = 0; k < 4; k++) d[k] = i0 + k < n ? kb::esub_base(u, wi[k]) : kb::eone();
  const kb::E4 p1 = kb::emul(d[0], d[1]), p2 = kb::emul(p1, d[2]), p3 = kb::emul(p2, d[3]);
  kb::E4 inv = kb::einv(p3), di[4];
  di[3] = kb::emul(inv, p2);
  inv = kb::emul(inv, d[3]);
  di[2] = kb::emul(inv, p1);
  inv = kb::emul(inv, d[2]);
  di[1] = kb::emul(inv, d[0]);
  di[0] = kb::emul(inv, d[1]);
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (i0 + k < n) out[i0 + k] = kb::emul(kb::escale(di[k], wi[k]), c);
}
__global__ __launch_bounds__(THREADS) void bary_weights(kb::E4 u, kb::E4 c, uint32_t w_n, size_t n, kb::E4* __restrict__ out) {
  bary_weights_body(u, c, w_n, n, out, blockIdx.x);
}
__global__ __launch_bounds__(THREADS) void bary_weights_batch(const WeightJob* __restrict__ jobs) {
  uint32_t bx;
  const WeightJob& j = find_job(jobs, bx);
  bary_weights_body(j.u, j.c, j.w_n, j.n, j.out, bx);
}

// Column evaluations. Block (bx, by) owns EVAL_COLS columns starting at EVAL_COLS*bx and the rows
//   r = 4 * (by * THREADS + tid) + k * 4 * gridDim.y * THREADS,   r .. r+3 per iteration,
// accumulating  acc0[c] += v[c][r] * w[r]  and (TWO)  acc1[c] += v[c][(r+1) mod n] * w[r].
// Every load of an iteration (4 weight quads, one 16-byte column quad + 1 word per column) is issued
// before the first multiply, so a wavefront keeps ~6 KiB in flight. Requires n % 4 == 0.
// One partial per block: partials[(by * width + col) * 2 + {0,1}].
constexpr int EVAL_COLS = 4;

// Block-wide sum of the 8 * EVAL_COLS accumulator words: every thread parks its words in LDS
// (word-major, conflict-free), 8 threads per word add 32-entry segments, one thread per word finishes.
constexpr int EVAL_WORDS = EVAL_COLS * 8;
__device__ __forceinline__ void block_reduce_store(kb::E4 (&acc)[EVAL_COLS][2], int c0, int width, kb::E4* __restrict__ partials,
                                                   uint32_t* lds /* EVAL_WORDS * THREADS + EVAL_WORDS * 8 */, uint32_t by) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int c = 0; c < EVAL_COLS; c++)
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int e = 0; e < 4; e++) lds[((c * 2 + pt) * 4 + e) * THREADS + tid] = acc[c][pt].c[e];
  __syncthreads();
  uint32_t* seg = lds + EVAL_WORDS * THREADS;
  {
    const int word = tid >> 3, part = tid & 7;  // THREADS / 8 == EVAL_WORDS
    const uint32_t* src = lds + word * THREADS + part * (THREADS / 8);
    uint32_t t = 0;
#pragma unroll 8
    for (int k = 0; k < THREADS / 8; k++) t = kb::add(t, src[k]);
    seg[word * 8 + part] = t;
  }
  __syncthreads();
  if (tid < EVAL_WORDS) {
    uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) t = kb::add(t, seg[tid * 8 + k]);
    const int c = tid >> 3, pt = (tid >> 2) & 1, e = tid & 3;
    if (c0 + c < width) partials[((size_t)by * width + c0 + c) * 2 + pt].c[e] = t;
  }
}

template <bool TWO>
__device__ __forceinline__ void eval_columns_body(const uint32_t* const* __restrict__ colptr, const uint32_t* __restrict__ colmask, size_t n, int width,
                                                  const kb::E4* __restrict__ weights,
                                                  kb::E4* __restrict__ partials, uint32_t* red, uint32_t bx, uint32_t by, uint32_t ny) {
  const int c0 = bx * EVAL_COLS;
  const uint32_t* cols[EVAL_COLS];
  uint32_t msk[EVAL_COLS];
#pragma unroll
  for (int c = 0; c < EVAL_COLS; c++) {   // clamp: duplicates are not stored
    cols[c] = gp::load(colptr + min(c0 + c, width - 1));
    msk[c] = gp::load(colmask + min(c0 + c, width - 1));
  }
  kb::E4 acc[EVAL_COLS][2];
#pragma unroll
  for (int c = 0; c < EVAL_COLS; c++) { acc[c][0] = kb::ezero(); acc[c][1] = kb::ezero(); }
  const size_t stride = (size_t)ny * THREADS * 4;
  for (size_t r = ((size_t)by * THREADS + threadIdx.x) * 4; r < n; r += stride) {
    kb::E4 w[4];
    uint4 v[EVAL_COLS];
    uint32_t vnext[EVAL_COLS];
    const size_t rn = r + 4 == n ? 0 : r + 4;
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = gp::load_e4(weights + r + k);
#pragma unroll
    for (int c = 0; c < EVAL_COLS; c++) {
      v[c] = gp::load(reinterpret_cast<const uint4*>(cols[c] + ((uint32_t)r & msk[c])));
      if (TWO) vnext[c] = gp::load(cols[c] + ((uint32_t)rn & msk[c]));
    }
#pragma unroll
    for (int c = 0; c < EVAL_COLS; c++) {
      const uint32_t x[5] = {v[c].x, v[c].y, v[c].z, v[c].w, TWO ? vnext[c] : 0u};
      // the iteration's four products and the running value (as one more product, by R mod p) share ONE reduction: 4 p^2 + 2^25 p < 2^64
      // (round 5: fifteen instructions per coefficient and iteration instead of twenty with a reduction every second product)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acc[c][0].c[e] = kb::reduce96_bounded(0, (uint64_t)w[0].c[e] * x[0] + (uint64_t)w[1].c[e] * x[1] + (uint64_t)w[2].c[e] * x[2] +
                                                     (uint64_t)w[3].c[e] * x[3] + (uint64_t)acc[c][0].c[e] * kb::ONE);
        if (TWO)
          acc[c][1].c[e] = kb::reduce96_bounded(0, (uint64_t)w[0].c[e] * x[1] + (uint64_t)w[1].c[e] * x[2] + (uint64_t)w[2].c[e] * x[3] +
                                                       (uint64_t)w[3].c[e] * x[4] + (uint64_t)acc[c][1].c[e] * kb::ONE);
      }
    }
  }
  block_reduce_store(acc, c0, width, partials, red, by);
}

// Small matrices (n < 4 * THREADS): one block per column group, scalar loads.
__device__ __forceinline__ void eval_columns_small_body(const uint32_t* const* __restrict__ colptr, const uint32_t* __restrict__ colmask, size_t n, int width,
                                                        const kb::E4* __restrict__ weights,
                                                        int two_points, kb::E4* __restrict__ partials, uint32_t* red, uint32_t bx) {
  const int c0 = bx * EVAL_COLS;
  kb::E4 acc[EVAL_COLS][2];
#pragma unroll
  for (int c = 0; c < EVAL_COLS; c++) { acc[c][0] = kb::ezero(); acc[c][1] = kb::ezero(); }
  for (size_t r = threadIdx.x; r < n; r += THREADS) {
    kb::E4 w = weights[r];
    size_t rn = r + 1 == n ? 0 : r + 1;
#pragma unroll
    for (int c = 0; c < EVAL_COLS; c++) {
      const uint32_t* col = colptr[min(c0 + c, width - 1)];
      const uint32_t mk = colmask[min(c0 + c, width - 1)];
      acc[c][0] = kb::eadd(acc[c][0], kb::escale(w, col[(uint32_t)r & mk]));
      if (two_points) acc[c][1] = kb::eadd(acc[c][1], kb::escale(w, col[(uint32_t)rn & mk]));
    }
  }
  block_reduce_store(acc, c0, width, partials, red, 0);
}
// every matrix of an opening in one launch: block -> (job, column group, row split)
__global__ __launch_bounds__(THREADS) void eval_columns_batch(const EvalJob* __restrict__ jobs) {
  __shared__ uint32_t red[EVAL_WORDS * THREADS + EVAL_WORDS * 8];
  uint32_t local;
  const EvalJob& j = find_job(jobs, local);
  uint32_t bx = local % (uint32_t)j.groups, by = local / (uint32_t)j.groups;
  if (j.lead & EVAL_XCD_AWARE) {
    // Every column group of a row slice reads the same weights (16 bytes per row against 4 per column and row): the `groups` blocks of a
    // slice are dealt to ONE XCD, back to back in its queue, so that the slice's weights come through HBM once and out of that XCD's L2
    // for the other groups. Workgroups go round-robin to the 8 XCDs: the job's first block is aligned to a multiple of 8 (`lead` idle
    // blocks), split is a multiple of 8, and block 8 s + x of the job is the s-th block of XCD x: slice (s / groups) * 8 + x, group s % groups.
    const uint32_t lead = j.lead & 0xff;
    if (local < lead) return;
    local -= lead;
    const uint32_t x = local & 7, s = local >> 3;
    bx = s % (uint32_t)j.groups;
    by = (s / (uint32_t)j.groups) * 8 + x;
  }
  if (j.kind == 0) eval_columns_body<false>(j.colptr, j.colmask, j.n, j.width, j.weights, j.partials, red, bx, by, (uint32_t)j.split);
  else if (j.kind == 1) eval_columns_body<true>(j.colptr, j.colmask, j.n, j.width, j.weights, j.partials, red, bx, by, (uint32_t)j.split);
  else eval_columns_small_body(j.colptr, j.colmask, j.n, j.width, j.weights, j.kind == 3, j.partials, red, bx);
}

// out[i] = sum_s partials[s * count + i]; one 64-lane block per output element
__device__ __forceinline__ void reduce_partials_body(const kb::E4* __restrict__ partials, int split, int count, kb::E4* __restrict__ out, int i) {
  kb::E4 acc = kb::ezero();
  for (int s = threadIdx.x; s < split; s += 64) acc = kb::eadd(acc, partials[(size_t)s * count + i]);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    uint32_t v = acc.c[e];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = kb::add(v, __shfl_xor(v, d));
    acc.c[e] = v;
  }
  if (threadIdx.x == 0) out[i] = acc;
}
__global__ __launch_bounds__(64) void reduce_partials_batch(const SumJob* __restrict__ jobs, kb::E4* __restrict__ out) {
  uint32_t i;
  const SumJob& j = find_job(jobs, i);
  reduce_partials_body(j.partials, j.split, j.count, out + j.out0, (int)i);
}

// Reduced openings for one LDE height N (fri.rs:103-204). One thread per stored row r (bit-reversed: x_r = 3 * w_N^bitrev(r)).
//   ro[r] = sum_m sum_pt (Yc[m][pt] - alpha^off(m,pt) * S_m[r]) / (z_pt - x_r),   S_m[r] = sum_col alpha^col L_m[r][col],
// off(m, 0) = the height's running column count before matrix m, off(m, 1) = off(m, 0) + width (fri.rs counts per (point, column)).
// Rearranged so that a row's fixed cost does not grow with the number of matrices (round 3: the kernel was 4 269 instructions per
// row of which ~1 200 were the columns themselves): the alpha powers come from ONE table indexed from off(m, 0), so
// S'_m = alpha^off(m,0) S_m falls out of the dot products directly (no product by A); the point-0 terms are summed as T0 = sum S'_m,
// the point-1 terms as T1 = sum alpha^width S'_m (one extension product per two-point matrix), and
//   ro[r] = (Y0 - T0) / (z0 - x) + (Y1 - T1) / (z1 - x)     with Y_pt = sum_m Yc[m][pt] (host).
// x_r comes from the two-level power table of w_N (two loads and a product instead of a 23-step square-and-multiply).
// word at byte offset `off` (a lane's 32-bit value) of a column whose base is wave-uniform: global_load with a scalar base and a vector offset
__device__ __forceinline__ uint32_t load_row(const uint32_t* base, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const uint32_t __attribute__((address_space(1)))*)((const char __attribute__((address_space(1)))*)base + off);
#else
  return *(const uint32_t*)((const char*)base + off);
#endif
}
struct ReduceMat {
  const uint32_t* lde;  // column-major, height N
  const uint32_t* const* colptr; const uint32_t* colmask;   // the matrix's window of the column tables (row r of column c: colptr[c][r & colmask[c]])
  int width;
  int n_points;         // 1: zeta only, 2: zeta and zeta*g
  uint32_t apow_off;    // off(m, 0)
  uint32_t pad;
  kb::E4 A1;            // alpha^width
};

__global__ __launch_bounds__(THREADS) void reduce_openings(const ReduceMat* __restrict__ mats, int n_mats, int log_N,
                                                           const kb::E4* __restrict__ alpha_pows, kb::E4 Y0, kb::E4 Y1, kb::E4 z0, kb::E4 z1,
                                                           uint32_t w_N, const uint32_t* __restrict__ pw_lo, const uint32_t* __restrict__ pw_hi,
                                                           kb::E4* __restrict__ ro) {
  size_t N = (size_t)1 << log_N;
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const uint32_t ex = kb::bitrev((uint32_t)r, log_N);
  const uint32_t wp = pw_lo ? kb::mul(pw_lo[ex & 1023], pw_hi[ex >> 10]) : kb::pow(w_N, (uint64_t)ex);
  uint32_t x = kb::mul(kb::GEN, wp);
  // 1 / (z0 - x) and 1 / (z1 - x) from one inverse
  const kb::E4 e0 = kb::esub_base(z0, x), e1 = kb::esub_base(z1, x);
  const kb::E4 e01 = kb::einv(kb::emul(e0, e1));
  kb::E4 d0 = kb::emul(e01, e1);
  kb::E4 d1 = kb::emul(e01, e0);
  kb::E4 T0 = kb::ezero(), T1 = kb::ezero();
  bool two = false;
  const uint32_t roff = (uint32_t)r * 4u;   // the row's byte offset in a column (N <= 2^30): one v_and with the column's mask, then a load with a scalar base
  for (int m = 0; m < n_mats; m++) {
    const ReduceMat& M = mats[m];
    const kb::E4* __restrict__ ap = alpha_pows + M.apow_off;
    // S' = sum_c alpha^(off + c) * L[r][c]: four base-field dot products over the matrix's columns, accumulated in 96 bits
    kb::Acc96 s0 = kb::acc96_zero(), s1 = kb::acc96_zero(), s2 = kb::acc96_zero(), s3 = kb::acc96_zero();
    // the tables through the constant address space: a wave-uniform index then is a scalar load (as global memory the compiler cannot
    // prove them unwritten and loads them per lane, one after the other: 2.55 ms instead of 1.75)
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t* __attribute__((address_space(4))) const* PtrTable;
    typedef const uint32_t __attribute__((address_space(4)))* MaskTable;
    const PtrTable cp = (PtrTable)(uint64_t)M.colptr;
    const MaskTable cm = (MaskTable)(uint64_t)M.colmask;
#else
    const uint32_t* const* cp = M.colptr;
    const uint32_t* cm = M.colmask;
#endif
    int c = 0;
    for (; c + 8 <= M.width; c += 8) {
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = load_row(cp[c + k], roff & cm[c + k]);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const kb::E4 a = ap[c + k];
        kb::acc96_fma_uniform(s0, a.c[0], v[k]);
        kb::acc96_fma_uniform(s1, a.c[1], v[k]);
        kb::acc96_fma_uniform(s2, a.c[2], v[k]);
        kb::acc96_fma_uniform(s3, a.c[3], v[k]);
      }
    }
    for (; c < M.width; c++) {
      const kb::E4 a = ap[c];
      const uint32_t v = load_row(cp[c], roff & cm[c]);
      kb::acc96_fma_uniform(s0, a.c[0], v);
      kb::acc96_fma_uniform(s1, a.c[1], v);
      kb::acc96_fma_uniform(s2, a.c[2], v);
      kb::acc96_fma_uniform(s3, a.c[3], v);
    }
    // up to 126 columns (every core and recursion chip; KeccakSponge has 3531): the sum is below reduce96_bounded's 127 * 2^63
    const kb::E4 S = M.width <= 126
                         ? kb::E4{{kb::reduce96_bounded(s0.hi, s0.lo), kb::reduce96_bounded(s1.hi, s1.lo), kb::reduce96_bounded(s2.hi, s2.lo),
                                   kb::reduce96_bounded(s3.hi, s3.lo)}}
                         : kb::E4{{kb::acc96_reduce(s0), kb::acc96_reduce(s1), kb::acc96_reduce(s2), kb::acc96_reduce(s3)}};
    T0 = kb::eadd(T0, S);
    if (M.n_points > 1) { T1 = kb::eadd(T1, kb::emul(M.A1, S)); two = true; }
  }
  kb::E4 acc = kb::emul(kb::esub(Y0, T0), d0);
  if (two) acc = kb::eadd(acc, kb::emul(kb::esub(Y1, T1), d1));
  ro[r] = acc;
}


// FRI fold (fri.rs:257-358): g[j] = e0 + (beta - x)(e1 - e0)/(-2x) [+ beta^2 * ro_next[j]],
// (e0, e1) = (f[2j], f[2j+1]), x = w_len^bitrev(2j).
// beta and beta^2 are read from memory (betas[0], betas[1]): the launch that finished this layer's tree sampled them on the device
// (merkle::observe_root_sample_beta), so the fold is queued right behind it without the host seeing the root first.
__global__ __launch_bounds__(THREADS) void fri_fold(const kb::E4* __restrict__ f, int log_len, const kb::E4* __restrict__ betas,
                                                    uint32_t w_len, uint32_t w_len_inv, uint32_t neg_half,
                                                    const kb::E4* __restrict__ ro_next,
                                                    kb::E4* __restrict__ g) {
  size_t half = (size_t)1 << (log_len - 1);
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= half) return;
  const kb::E4 beta = betas[0], beta_sq = betas[1];
  kb::E4 e0 = f[2 * j], e1 = f[2 * j + 1];
  uint32_t e = kb::bitrev((uint32_t)(2 * j), log_len);
  uint32_t xinv = kb::pow(w_len_inv, (uint64_t)e);
  uint32_t x = kb::pow(w_len, (uint64_t)e);
  // (beta - x) * (e1 - e0) * (-1/2) * x^-1
  kb::E4 t = kb::emul(kb::esub_base(beta, x), kb::esub(e1, e0));
  kb::E4 r = kb::eadd(e0, kb::escale(t, kb::mul(neg_half, xinv)));
  if (ro_next) r = kb::eadd(r, kb::emul(beta_sq, ro_next[j]));
  g[j] = r;
}

// The query phase's gather (fri.rs:110-128, 279-306): every query reads the same *shape* of words — the opened row of every committed
// matrix, the sibling digests up each input tree, the FRI sibling values and their paths — at positions that depend on the query index only
// through shifts. One template entry per word of a query: the word is base[(((q >> shift) ^ flip) * step) + k]. The host uploads the
// template once (a few thousand entries) instead of one pointer per word per query (84 x as many).
struct QueryWord {
  const uint32_t* base;
  uint32_t shift, flip, step, k;
};
__global__ __launch_bounds__(THREADS) void gather_queries(const QueryWord* __restrict__ tmpl, size_t per_query, const uint32_t* __restrict__ indices,
                                                          size_t n_queries, uint32_t* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_query * n_queries) return;
  const size_t qi = i / per_query;
  const QueryWord w = tmpl[i - qi * per_query];
  const size_t q = indices[qi];
  dst[i] = w.base[(((q >> w.shift) ^ w.flip) * w.step) + w.k];
}

// dst[i] = *src[i]
__global__ __launch_bounds__(THREADS) void gather_words(const uint32_t* const* __restrict__ src, size_t count, uint32_t* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = *src[i];
}

// 32x32 tile transpose between row-major [h][w] and column-major [w][h]
__global__ void transpose(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t rows, size_t cols) {
  __shared__ uint32_t tile[32][33];
  size_t bx = (size_t)blockIdx.x * 32, by = (size_t)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    size_t r = by + j, c = bx + threadIdx.x;
    if (r < rows && c < cols) tile[j][threadIdx.x] = in[r * cols + c];
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    size_t c = bx + j, r = by + threadIdx.x;
    if (r < rows && c < cols) out[c * rows + r] = tile[threadIdx.x][j];
  }
}

// rows [r0, r0 + rows) of a row-major [.][cols] slab -> column-major matrix of `height` rows. This is where host words enter the
// prover (zkm_matrix_upload, zkm_matrix_upload_async, zkm_tracegen_flat): every kernel behind it sizes its accumulators for canonical
// Montgomery words (< p: what a RowMajorMatrix<KoalaBear> holds — MontyField31 keeps its value reduced), so a word >= p is reduced here
// (3p > 2^32: two conditional subtractions cover every u32) instead of silently computing with a wrong bound.
__global__ void transpose_slab(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t rows, size_t cols, size_t r0,
                               size_t height) {
  __shared__ uint32_t tile[32][33];
  size_t bx = (size_t)blockIdx.x * 32, by = (size_t)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    size_t r = by + j, c = bx + threadIdx.x;
    if (r < rows && c < cols) {
      uint32_t w = in[r * cols + c];
      w = w >= kb::P ? w - kb::P : w;
      w = w >= kb::P ? w - kb::P : w;
      tile[j][threadIdx.x] = w;
    }
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    size_t c = bx + j, r = by + threadIdx.x;
    if (r < rows && c < cols) out[c * height + r0 + r] = tile[threadIdx.x][j];
  }
}

// zkm_tracegen_flat's rows: `in` holds n_words words, records end to end (row-major, `cols` per row); rows behind them are zero. The
// same tile transpose as transpose_slab, reading past the records as zero: no staging buffer, no memset, and the source may be an
// address zkm_events_upload_async returned. Host words are reduced to canonical form on the way in, as in transpose_slab.
__global__ void flat_rows(const uint32_t* __restrict__ in, size_t n_words, uint32_t* __restrict__ out, size_t cols, size_t height) {
  __shared__ uint32_t tile[32][33];
  size_t bx = (size_t)blockIdx.x * 32, by = (size_t)blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    size_t r = by + j, c = bx + threadIdx.x;
    uint32_t w = 0;
    if (r < height && c < cols && r * cols + c < n_words) {
      w = in[r * cols + c];
      w = w >= kb::P ? w - kb::P : w;
      w = w >= kb::P ? w - kb::P : w;
    }
    tile[j][threadIdx.x] = w;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    size_t c = bx + j, r = by + threadIdx.x;
    if (r < height && c < cols) out[c * height + r] = tile[threadIdx.x][j];
  }
}

}  // namespace open
