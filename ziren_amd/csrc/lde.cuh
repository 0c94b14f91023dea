// Coset low-degree extension of column-major trace matrices on gfx950.
//
// Replaces p3-dft Radix2DitParallel::coset_lde_batch + bit_reverse_rows as called by
// TwoAdicFriPcs::commit (call sites crates/stark/src/prover.rs:277,403,497; semantics pinned by
// the verifier's x formula, crates/recursion/circuit/src/fri.rs:140-150):
//     out[bitrev(j)] = P(shift * w_N^j),  N = n << log_blowup,  P = interpolant of the column on H_n.
//
// Layout: a matrix is column-major in HBM, column c at base + c * height, so every kernel here
// streams contiguous 4-byte words along a column (coalesced across a wavefront).
//
// Algorithm (per column): the 2^bl cosets of H_n inside shift*K_N are 2^bl independent size-n
// transforms, and coset j lands in the contiguous output block [bitrev_bl(j) * n, +n) in
// bit-reversed row order. With n = A * B (A = 2^la strided, B = 2^lb <= 8192 contiguous):
//   1. lde_cols<false>  : A-point inverse DIF down the strided dimension, tile [A][T+1] in LDS; row pr of the
//        result holds frequency k1 = bitrev_la(pr)
//   2. lde_rows         : per contiguous row of B words, all in LDS:
//        twiddle w_n^(-i0 k1) on load, B-point inverse DIF -> coefficients (bit-reversed),
//        then for each coset: scale by shift_j^k / n, B-point forward DIT, twiddle w_n^(j0 k1);
//        the row is stored at row k1 (natural frequency order) of a per-coset scratch matrix
//   3. lde_cols<true>   : A-point forward DIF down the strided dimension (natural rows in, bit-reversed
//        rows out), then the padded tile is read out transposed so each tile column lands as one contiguous,
//        already bit-reversed A-word segment of the output.
// For n <= 8192 (A = 1) step 2 alone reads the column once and writes the LDE once.
#pragma once
#include "kb31.cuh"
#include "gptr.cuh"

namespace lde {

constexpr int LOG_ROW_MAX = 13;        // B <= 8192 words = 32 KiB of LDS per buffer
constexpr int THREADS = 512;  // 8 waves per block: two blocks per CU keep 4 waves per SIMD over the barriers

// ---- in-LDS transforms ---------------------------------------------------------------------------
// A transform of 2^logn points runs as passes of up to 4 radix-2 stages. In one pass a thread owns a
// group of 2^R points (R <= 4) that only interact with each other during those stages, keeps them in
// VGPRs, and touches LDS once to read and once to write them: 13 stages cost 4 LDS round trips and
// 4 barriers instead of 13.
//
// Sequence layout: element i of sequence t lives at buf[phys(i) * istride + t]; 2^lognb sequences are
// interleaved (strided kernels: t = column inside the tile). PAD inserts one word per 32 so that the
// power-of-two strides of the late stages spread over the LDS banks (contiguous kernels only).
//
// Twiddles are stage-major: for stage s (butterfly span m = n >> (s+1)) the m factors w_{2m}^off sit
// contiguously at tw[n - (n >> s) + off], so lanes with consecutive `off` read consecutive words.
// DIF: natural in -> bit-reversed out (stages 0..logn-1).  DIT: bit-reversed in -> natural out.
template <bool PAD>
__device__ __forceinline__ uint32_t phys(uint32_t i) { return PAD ? i + (i >> 5) : i; }

// All index arithmetic is unsigned 32-bit so that LDS addresses and the twiddle loads (scalar base +
// 32-bit lane offset) need no sign extension or 64-bit address math.
// R radix-2 stages (s0 .. s0+R-1) on the 2^R points a thread holds in registers; `lo` is the group's offset
// inside its butterfly span (the low logm2 bits of the group index).
// LAZY (DIT only): the points are int32 words congruent to the field elements, |v| < 2^31, and stay that way from
// stage to stage: a + w b and a - w b are the signed Montgomery reductions of a R + b w and a R - b w (R mod p = ONE),
// two multiply-adds and two three-instruction reductions per butterfly instead of a full multiply and two modular
// additions (10 instructions instead of 12). Bound: |out| <= (|a| ONE + |b| w) / 2^32 + p / 2, i.e. in units of 2^31
// X <- 0.50390625 X + 0.49609375, which stays below 1 for any number of stages when the inputs are reduced.
template <bool DIF, int R, bool LAZY = false>
__device__ __forceinline__ void butterflies(uint32_t (&x)[1 << R], uint32_t n, uint32_t s0, uint32_t lo, uint32_t logm2,
                                            const uint32_t* __restrict__ tw) {
  static_assert(!(DIF && LAZY), "the lazy form exists for the DIT butterfly only");
#pragma unroll
  for (int qq = 0; qq < R; qq++) {
    const uint32_t q = DIF ? qq : R - 1 - qq;
    const uint32_t s = s0 + q;
    const uint32_t half = 1u << (R - 1 - q);
    const uint32_t tbase = (n - (n >> s)) + lo;
#pragma unroll
    for (uint32_t j0 = 0; j0 < (1u << R); j0++) {
      if (j0 & half) continue;
      const uint32_t j1 = j0 + half;
      const uint32_t w = gp::load(tw + tbase + ((j0 & (half - 1)) << logm2));
      uint32_t a = x[j0], b = x[j1];
      if (DIF) {
        x[j0] = kb::add(a, b);
        x[j1] = kb::mul_signed(a - b, w);
      } else if (LAZY) {
        const int64_t ar = kb::mad_i64_i32_uniform((int32_t)a, kb::ONE, 0);
        x[j0] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32((int32_t)b, (int32_t)w, ar));
        x[j1] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32((int32_t)b, -(int32_t)w, ar));  // -w, not -b: shared by the butterflies of a twiddle
      } else {
        b = kb::mul(b, w);
        x[j0] = kb::add(a, b);
        x[j1] = kb::sub(a, b);
      }
    }
  }
}

template <bool DIF, int R, bool PAD, bool LAZY = false>
__device__ __forceinline__ void ntt_pass(uint32_t* buf, uint32_t logn, uint32_t lognb, uint32_t istride, uint32_t s0,
                                         const uint32_t* __restrict__ tw, uint32_t tid = threadIdx.x) {
  const uint32_t n = 1u << logn;
  const uint32_t logm2 = logn - s0 - R;  // points of a group are m2 = 2^logm2 apart
  const uint32_t total = (n >> R) << lognb;
  for (uint32_t u = tid; u < total; u += blockDim.x) {
    const uint32_t t = u & ((1u << lognb) - 1), g = u >> lognb;
    const uint32_t lo = g & ((1u << logm2) - 1), hi = g >> logm2;
    const uint32_t base = (hi << (R + logm2)) + lo;
    uint32_t x[1 << R];
#pragma unroll
    for (uint32_t j = 0; j < (1u << R); j++) x[j] = buf[phys<PAD>(base + (j << logm2)) * istride + t];
    butterflies<DIF, R, LAZY>(x, n, s0, lo, logm2, tw);
#pragma unroll
    for (uint32_t j = 0; j < (1u << R); j++) buf[phys<PAD>(base + (j << logm2)) * istride + t] = x[j];
  }
  __syncthreads();
}

template <bool DIF, bool PAD>
__device__ __forceinline__ void lds_ntt(uint32_t* buf, int logn, int lognb, int istride, const uint32_t* __restrict__ tw) {
  // passes of 4 stages (16 points per thread), then one short pass for the remaining 1-3 stages
  const int rem = logn & 3;
  if (DIF) {
    int s0 = 0;
    for (; s0 + 4 <= logn; s0 += 4) ntt_pass<true, 4, PAD>(buf, logn, lognb, istride, s0, tw);
    if (rem == 3) ntt_pass<true, 3, PAD>(buf, logn, lognb, istride, s0, tw);
    if (rem == 2) ntt_pass<true, 2, PAD>(buf, logn, lognb, istride, s0, tw);
    if (rem == 1) ntt_pass<true, 1, PAD>(buf, logn, lognb, istride, s0, tw);
  } else {
    int s0 = logn - rem;
    if (rem == 3) ntt_pass<false, 3, PAD>(buf, logn, lognb, istride, s0, tw);
    if (rem == 2) ntt_pass<false, 2, PAD>(buf, logn, lognb, istride, s0, tw);
    if (rem == 1) ntt_pass<false, 1, PAD>(buf, logn, lognb, istride, s0, tw);
    for (s0 -= 4; s0 >= 0; s0 -= 4) ntt_pass<false, 4, PAD>(buf, logn, lognb, istride, s0, tw);
  }
}

// ---- natural in -> bit-reversed out with the twiddle BEFORE the add (Cooley-Tukey butterflies on a decimation-in-frequency geometry) ----
// Stage s splits sub-problem r (positions [r M, (r + 1) M), M = n >> s) into its halves: a' = a + c b, b' = a - c b for the pairs
// (p, p + M/2), with ONE factor c per sub-problem: c = shift(s, r)^(M/2), shift(s, r) = g w^bitrev_s(r) the coset the sub-problem is
// evaluated on (even outputs keep the shift, odd outputs multiply it by the current root). Two things follow. (i) A pre-scaling of the
// inputs by g^i is absorbed: the table tw[(1 << s) - 1 + r] = (g w^bitrev_s(r))^(M/2) simply starts from g instead of 1 — this is how
// lde_rows_big's inverse row transform takes its w_n^(-i0 k1) twiddle without one multiplication (round 3; the table is per row k1:
// n words per height). (ii) The butterfly has the lazy form of the forward passes (ten instructions, signed unreduced words).
// `hi` = the group's sub-problem index at stage s0 (its position >> (logn - s0)).
template <int R, bool UNIFORM>
__device__ __forceinline__ void butterflies_ct(uint32_t (&x)[1 << R], uint32_t s0, uint32_t hi, const uint32_t* __restrict__ tw) {
#pragma unroll
  for (int q = 0; q < R; q++) {
    const uint32_t half = 1u << (R - 1 - q);
    const uint32_t tbase = (1u << (s0 + q)) - 1 + (hi << q);
#pragma unroll
    for (uint32_t j0 = 0; j0 < (1u << R); j0++) {
      if (j0 & half) continue;
      const uint32_t j1 = j0 + half;
      const uint32_t w = gp::load(tw + tbase + (j0 >> (R - q)));
      const int32_t a = (int32_t)x[j0], b = (int32_t)x[j1];
      const int64_t ar = kb::mad_i64_i32_uniform(a, kb::ONE, 0);
      if (UNIFORM) {
        x[j0] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32_uniform(b, w, ar));
        x[j1] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32_uniform(b, 0u - w, ar));  // the negation is scalar work
      } else {
        x[j0] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(b, (int32_t)w, ar));
        x[j1] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(b, -(int32_t)w, ar));  // one negation per twiddle (15 per 32 butterflies)
      }
    }
  }
}
// one pass of R such stages over an LDS buffer of 2^logn points (PAD as in ntt_pass); a thread's 2^R points are 2^logm2 apart
// (2^lognb sequences interleaved, element i of sequence t at buf[phys(i) * istride + t], as in ntt_pass)
template <int R, bool PAD>
__device__ __forceinline__ void ntt_pass_ct(uint32_t* buf, uint32_t logn, uint32_t s0, const uint32_t* __restrict__ tw, uint32_t tid = threadIdx.x,
                                            uint32_t lognb = 0, uint32_t istride = 1) {
  const uint32_t logm2 = logn - s0 - R;
  const uint32_t total = ((1u << logn) >> R) << lognb;
  for (uint32_t u = tid; u < total; u += blockDim.x) {
    const uint32_t t = u & ((1u << lognb) - 1), g = u >> lognb;
    const uint32_t lo = g & ((1u << logm2) - 1), hi = g >> logm2;
    const uint32_t base = (hi << (R + logm2)) + lo;
    uint32_t x[1 << R];
#pragma unroll
    for (uint32_t j = 0; j < (1u << R); j++) x[j] = buf[phys<PAD>(base + (j << logm2)) * istride + t];
    butterflies_ct<R, false>(x, s0, hi, tw);
#pragma unroll
    for (uint32_t j = 0; j < (1u << R); j++) buf[phys<PAD>(base + (j << logm2)) * istride + t] = x[j];
  }
  __syncthreads();
}
// a whole transform of 2^logn points that way: passes of four stages from stage 0 up, one shorter pass for the last 1-3 stages. In: reduced
// or lazy words; out: lazy signed words (|v| < 2^31, congruent to the result), bit-reversed order.
template <bool PAD>
__device__ __forceinline__ void lds_ntt_ct(uint32_t* buf, int logn, int lognb, int istride, const uint32_t* __restrict__ tw, bool leave_single = false) {
  int s0 = 0;
  for (; s0 + 4 <= logn; s0 += 4) ntt_pass_ct<4, PAD>(buf, logn, s0, tw, threadIdx.x, lognb, istride);
  const int rem = logn - s0;
  if (rem == 3) ntt_pass_ct<3, PAD>(buf, logn, s0, tw, threadIdx.x, lognb, istride);
  if (rem == 2) ntt_pass_ct<2, PAD>(buf, logn, s0, tw, threadIdx.x, lognb, istride);
  if (rem == 1 && !leave_single) ntt_pass_ct<1, PAD>(buf, logn, s0, tw, threadIdx.x, lognb, istride);  // leave_single: the caller's read-out does it
}
// the last stage of such a transform (pairs of adjacent positions 2 g, 2 g + 1; one twiddle per pair) on values already in registers
__device__ __forceinline__ void last_stage_ct(uint32_t& x0, uint32_t& x1, uint32_t w) {
  const int32_t a = (int32_t)x0, b = (int32_t)x1;
  const int64_t ar = kb::mad_i64_i32_uniform(a, kb::ONE, 0);
  x0 = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(b, (int32_t)w, ar));
  x1 = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(b, -(int32_t)w, ar));
}
// a lazy word back to [0, p): v in (-2^31, 2^31) -> v + 2p if negative (in (p - 2^24, 2p)), then one conditional subtraction
__device__ __forceinline__ uint32_t canonical(uint32_t lazy) {
  const uint32_t w = lazy + ((uint32_t)((int32_t)lazy >> 31) & (2u * kb::P));
  return kb::umin32(w, w - kb::P);
}

// Two-level power table of g in LDS: lo[i] = g^i (i < 64), hi[i] = g^(64 i) (i < nhi).
// `scale` is folded into the low table, so pow_lookup() returns scale * g^e.
__device__ __forceinline__ void build_pow_table(uint32_t g, uint32_t* lo, uint32_t* hi, int nhi, uint32_t scale = kb::ONE) {
  uint32_t g64 = g;
  for (int i = 0; i < 6; i++) g64 = kb::sqr(g64);
  for (int i = threadIdx.x; i < 64 + nhi; i += blockDim.x) {
    if (i < 64) lo[i] = kb::mul(scale, kb::pow(g, (uint64_t)i));
    else hi[i - 64] = kb::pow(g64, (uint64_t)(i - 64));
  }
}
__device__ __forceinline__ uint32_t pow_lookup(const uint32_t* lo, const uint32_t* hi, uint32_t e) {
  return kb::mul(lo[e & 63], hi[e >> 6]);
}

// ---- batch descriptors ------------------------------------------------------------------------------
// One launch per kernel covers every matrix of a commit (round 3; before, each matrix had its own three launches and a
// SYN-22 proof spent 96 dispatches and their ~8 us gaps on the LDEs alone). The matrices are grouped by height; a flat
// blockIdx.x is mapped to (group, block inside the group) through the groups' cumulative block counts and then to
// (matrix, column) through the matrices' first-column indices. The descriptor lives in device memory (one small upload per
// commit); every index into it is wave-uniform, so the reads are scalar loads.
constexpr int MAX_GROUPS = 24, MAX_MATS = 64;
enum { K_COLS_INV = 0, K_ROWS_BIG = 1, K_COLS_FWD = 2, K_ROWS_SMALL = 3 };

struct Mat {
  const uint32_t* in;   // n x w evaluations, column-major
  uint32_t* out;        // N x w LDE, column-major, rows bit-reversed
  uint32_t* tmp1;       // n x w            (la > 0): the column after the strided inverse pass
  uint32_t* tmp2;       // cosets x n x w   (la > 0): coset j's rows, natural frequency order, at tmp2 + j n w
  const uint32_t* twf;  // la > 0: per coset the forward stage twiddles of the B-point row transform with the coset scaling folded in
                        //         (cosets x B words, see fill_scaled_stage_twiddles); la == 0: unused
  const uint32_t* cs;   // la > 0: cs[j A + k1] = shift_j^k1 / n
  uint32_t col0, w;     // first column inside the group, width
  uint32_t shift;       // shift of coset 0 (shift_j = shift w_N^j)
  uint32_t uniform;     // != 0: the producer says every row of `in` is the same (a trace generator without events, zkm_matrix::uniform_rows):
                        //       every column is constant and lde_cols<false> only records its first word
  uint32_t* cflag;      // la > 0: per column two words, zeroed before the batch: [2c] != 0 once some tile of column c was seen to hold two
                        //         different values (lde_cols<false>), [2c + 1] = the column's first word. A column whose words are all equal
                        //         is the constant polynomial: its LDE is that word on every row of every coset, so the two later passes do
                        //         not transform it (lde_rows_big returns, lde_cols<true> fills its output tile). The shape step pads every
                        //         shard with chips that have no events at all (constant rows: 14 % of the benchmarked shard's cells),
                        //         and real traces hold columns that never change (unused selectors)
};
struct Group {
  int la, lb, logT, pad;
  uint32_t n_cols, first_mat, n_mats, pad2;
  uint32_t blk_end[4];                     // cumulative number of blocks up to and including this group, per kernel
  const uint32_t *twb_fwd, *twb_inv;       // stage-major twiddles of the B-point transform (la == 0 uses both, la > 0 the inverse)
  const uint32_t *twa_fwd, *twa_inv;       // of the A-point transform, by sub-problem (fill_ct_twiddles)
  const uint32_t *pw_lo, *pw_hi;           // w_n^e = pw_lo[e & 1023] * pw_hi[e >> 10]
  const uint32_t* tw_rows;                 // la > 0: per row k1 the B - 1 sub-problem twiddles of the inverse row transform (fill_row_twiddles)
  uint32_t w_N, n_inv, pad3[2];
};
struct Batch {
  int n_groups, log_blowup;
  Group g[MAX_GROUPS];
  Mat m[MAX_MATS];
};

__device__ __forceinline__ const Group& find_group(const Batch* __restrict__ d, int which, uint32_t& local) {
  uint32_t gi = 0, start = 0;
  const uint32_t b = blockIdx.x;
  while (b >= d->g[gi].blk_end[which]) { start = d->g[gi].blk_end[which]; gi++; }
  local = b - start;
  return d->g[gi];
}
__device__ __forceinline__ const Mat& find_mat(const Batch* __restrict__ d, const Group& g, uint32_t col) {
  uint32_t mi = g.first_mat;
  while (col >= d->m[mi].col0 + d->m[mi].w) mi++;
  return d->m[mi];
}

// Step 1 / 3: A-point transforms down the strided dimension of each column.
// Blocks of a group: (B / T) x columns (x cosets for FORWARD), x fastest. in/out are column-major.
// The LDS tile is [A][T + 1] (one pad word per row: the transposed read-out below walks down a column).
// FORWARD == false: inverse DIF over rows i1 (natural) -> rows in bit-reversed k1 order, written to tmp1.
// FORWARD == true : rows arrive in natural k1 order (lde_rows_big stores row k1 = bitrev(pr)); a forward DIF
//                   leaves row q holding L[B * bitrev_la(q) + j0], which belongs at
//                   out[c * N + out_block(z) * n + bitrev_lb(j0) * A + q]:
//                   one contiguous A-word segment per tile column.
template <bool FORWARD>
__global__ __launch_bounds__(THREADS) void lde_cols(const Batch* __restrict__ d) {
  extern __shared__ uint32_t lds[];
  uint32_t local;
  const Group& g = find_group(d, FORWARD ? K_COLS_FWD : K_COLS_INV, local);
  const int la = g.la, lb = g.lb, logT = g.logT, log_blowup = d->log_blowup;
  const int A = 1 << la, T = 1 << logT, TP = T + 1;
  const size_t B = (size_t)1 << lb, n = (size_t)A << lb;
  const uint32_t xb = local & ((1u << (lb - logT)) - 1);
  const uint32_t rest = local >> (lb - logT);
  const uint32_t col = FORWARD ? rest % g.n_cols : rest;
  const int z = FORWARD ? (int)(rest / g.n_cols) : 0;
  const Mat& m = find_mat(d, g, col);
  const size_t c = col - m.col0;
  const size_t t0 = (size_t)xb << logT;
  const uint32_t* src = (FORWARD ? m.tmp2 + (size_t)z * n * m.w : m.in) + c * n + t0;
  const uint32_t* __restrict__ tw = FORWARD ? g.twa_fwd : g.twa_inv;   // sub-problem twiddles (fill_ct_twiddles)
  // 16-byte global accesses (T >= 8), four in flight per thread
  const int logTq = logT - 2;
  const int quads = (A << logT) >> 2;
  if (!FORWARD && m.uniform) {
    // nothing to find out: the column's word for the two later passes, the "differs" word stays zero
    if (xb == 0 && threadIdx.x == 0) gp::store(m.cflag + 2 * c + 1, gp::load(m.in + c * n));
    return;
  }
  if (FORWARD && gp::load(m.cflag + 2 * c) == 0) {
    // a constant column (see Mat::cflag): every word of this block's output tile is the column's word
    const uint32_t v0 = gp::load(m.cflag + 2 * c + 1);
    const uint32_t v = kb::umin32(v0, v0 - kb::P);
    uint32_t* dst = m.out + c * (n << log_blowup) + (size_t)kb::bitrev(z, log_blowup) * n;
    if (la < 2) {
      for (int u = threadIdx.x; u < (A << logT); u += blockDim.x)
        gp::store(dst + (size_t)kb::bitrev((uint32_t)(t0 + (u >> la)), lb) * A + (u & (A - 1)), v);
      return;
    }
    const int logAq = la - 2;
    for (int u0 = threadIdx.x; u0 < quads; u0 += blockDim.x) {
      const int t = u0 >> logAq, q = (u0 & ((1 << logAq) - 1)) << 2;
      gp::store(reinterpret_cast<uint4*>(dst + (size_t)kb::bitrev((uint32_t)(t0 + t), lb) * A + q), make_uint4(v, v, v, v));
    }
    return;
  }
  const uint32_t first_word = FORWARD ? 0u : gp::load(m.in + c * n);   // the column's first word: what a constant column holds everywhere
  uint32_t differs = 0;
  for (int u0 = threadIdx.x; u0 < quads; u0 += 4 * blockDim.x) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int u = u0 + k * blockDim.x;
      if (u < quads) v[k] = gp::load(reinterpret_cast<const uint4*>(src + (size_t)(u >> logTq) * B + ((u & ((1 << logTq) - 1)) << 2)));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int u = u0 + k * blockDim.x;
      if (u < quads) {
        uint32_t* dd = lds + (u >> logTq) * TP + ((u & ((1 << logTq) - 1)) << 2);
        dd[0] = v[k].x; dd[1] = v[k].y; dd[2] = v[k].z; dd[3] = v[k].w;
        if (!FORWARD) differs |= (v[k].x ^ first_word) | (v[k].y ^ first_word) | (v[k].z ^ first_word) | (v[k].w ^ first_word);
      }
    }
  }
  if (!FORWARD) {
    // a wave that saw a word different from the column's first one marks the column (every writer writes the same word)
    if (__any(differs != 0) && (threadIdx.x & 63) == 0) gp::store(m.cflag + 2 * c, 1u);
    if (xb == 0 && threadIdx.x == 0) gp::store(m.cflag + 2 * c + 1, first_word);
  }
  __syncthreads();
  // natural rows in, bit-reversed rows out either way, lazy Cooley-Tukey butterflies (ten instructions instead of twelve). The inverse
  // pass hands its lazy words on as they are (lde_rows_big's first butterflies take signed words); the forward pass writes the
  // committed LDE and brings them back to [0, p) on the way out.
  // When the stage count leaves a single last stage (la = 5, 9, 13) it is not a pass of its own over LDS: it pairs adjacent rows, and
  // the read-out below has both rows of a pair in hand (round 3: one LDS round trip and barrier fewer).
  const bool fused_last = la >= 2 && (la & 3) == 1;
  lds_ntt_ct<false>(lds, la, logT, TP, tw, fused_last);
  const uint32_t* __restrict__ tw_last = tw + ((1u << (la - 1)) - 1);  // stage la - 1: sub-problem g = pair g
  if (!FORWARD) {
    uint32_t* dst = m.tmp1 + c * n + t0;
    if (fused_last) {
      for (int u0 = threadIdx.x; u0 < (quads >> 1); u0 += blockDim.x) {
        const int rp = u0 >> logTq, qd = (u0 & ((1 << logTq) - 1)) << 2;
        const uint32_t* sp = lds + (2 * rp) * TP + qd;
        uint32_t x[4] = {sp[0], sp[1], sp[2], sp[3]}, y[4] = {sp[TP], sp[TP + 1], sp[TP + 2], sp[TP + 3]};
        const uint32_t w = gp::load(tw_last + rp);
#pragma unroll
        for (int k = 0; k < 4; k++) last_stage_ct(x[k], y[k], w);
        gp::store(reinterpret_cast<uint4*>(dst + (size_t)(2 * rp) * B + qd), make_uint4(x[0], x[1], x[2], x[3]));
        gp::store(reinterpret_cast<uint4*>(dst + (size_t)(2 * rp + 1) * B + qd), make_uint4(y[0], y[1], y[2], y[3]));
      }
      return;
    }
    for (int u0 = threadIdx.x; u0 < quads; u0 += blockDim.x) {
      const uint32_t* sp = lds + (u0 >> logTq) * TP + ((u0 & ((1 << logTq) - 1)) << 2);
      gp::store(reinterpret_cast<uint4*>(dst + (size_t)(u0 >> logTq) * B + ((u0 & ((1 << logTq) - 1)) << 2)), make_uint4(sp[0], sp[1], sp[2], sp[3]));
    }
  } else {
    uint32_t* dst = m.out + c * (n << log_blowup) + (size_t)kb::bitrev(z, log_blowup) * n;
    if (la < 2) {  // A = 2: scalar stores
      for (int u = threadIdx.x; u < (A << logT); u += blockDim.x) {
        const int t = u >> la, q = u & (A - 1);
        gp::store(dst + (size_t)kb::bitrev((uint32_t)(t0 + t), lb) * A + q, canonical(lds[q * TP + t]));
      }
      return;
    }
    const int logAq = la - 2;
    for (int u0 = threadIdx.x; u0 < quads; u0 += blockDim.x) {
      const int t = u0 >> logAq, q = (u0 & ((1 << logAq) - 1)) << 2;
      const uint32_t* sp = lds + q * TP + t;
      const size_t j0 = t0 + t;
      uint32_t x[4] = {sp[0], sp[TP], sp[2 * TP], sp[3 * TP]};
      if (fused_last) {
        last_stage_ct(x[0], x[1], gp::load(tw_last + (q >> 1)));
        last_stage_ct(x[2], x[3], gp::load(tw_last + (q >> 1) + 1));
      }
      gp::store(reinterpret_cast<uint4*>(dst + (size_t)kb::bitrev((uint32_t)j0, lb) * A + q),
                make_uint4(canonical(x[0]), canonical(x[1]), canonical(x[2]), canonical(x[3])));
    }
  }
}

// Step 2 when the whole column fits one LDS row (n = B <= 8192, A = 1): one block per column; reads the column once,
// writes its LDE once, rows already bit-reversed. Taller columns go through lde_rows_big below.
__global__ __launch_bounds__(THREADS) void lde_rows(const Batch* __restrict__ d) {
  extern __shared__ uint32_t lds[];
  uint32_t col;
  const Group& g = find_group(d, K_ROWS_SMALL, col);
  const Mat& m = find_mat(d, g, col);
  const int lb = g.lb, log_blowup = d->log_blowup;
  const int B = 1 << lb;
  const int nhi = B > 64 ? (B >> 6) : 1;
  const int BP = B + (B >> 5);     // padded length, see phys<>
  uint32_t* coef = lds;            // BP
  uint32_t* work = lds + BP;       // BP
  uint32_t* lo2 = work + BP;       // 64   powers of shift_j   (coset scaling)
  uint32_t* hi2 = lo2 + 64;        // nhi
  const size_t c = col - m.col0;
  const uint32_t* src = m.in + c * B;
  const uint32_t* __restrict__ tw_fwd = g.twb_fwd;
  const uint32_t* __restrict__ tw_inv = g.twb_inv;
  for (int i = threadIdx.x; i < B; i += blockDim.x) coef[phys<true>(i)] = gp::load(src + i);
  __syncthreads();
  if (lb > 0) lds_ntt<true, true>(coef, lb, 0, 1, tw_inv);
  // coef[pc] = n * c_k with k = bitrev_lb(pc)
  const int ncosets = 1 << log_blowup;
  uint32_t sj = m.shift;
  for (int j = 0; j < ncosets; j++) {
    // shift_j = shift * w_N^j
    __syncthreads();
    build_pow_table(sj, lo2, hi2, nhi, g.n_inv);  // 1/n folded in
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x)
      work[phys<true>(i)] = kb::mul(coef[phys<true>(i)], pow_lookup(lo2, hi2, kb::bitrev(i, lb)));
    __syncthreads();
    if (lb > 0) lds_ntt<false, true>(work, lb, 0, 1, tw_fwd);
    uint32_t* dst = m.out + c * ((size_t)B << log_blowup) + (size_t)kb::bitrev(j, log_blowup) * B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) gp::store(dst + i, work[phys<true>(kb::bitrev(i, lb))]);
    sj = kb::mul(sj, g.w_N);
  }
}

// Step 2 for n > 8192 (la > 0, B = 8192, 512 threads), blocks of a group = A x columns (row position fastest): the data stays
// in registers wherever two neighbouring steps touch the same points, so one 33 KiB LDS buffer and 9 LDS round trips per row
// (3 for the inverse transform, 3 per coset) instead of one per four stages plus one for every twiddle / scaling step:
//   * the load (with its w_n^(-i0 k1) twiddle) feeds the first inverse pass directly: thread g owns i0 = g + 512 j;
//   * the last inverse stage (pairs 2g, 2g+1) leaves the 16 coefficients of a thread in VGPRs, where they stay
//     for every coset;
//   * per coset the scaling by shift_j^k is NOT a step of its own (round 3): a forward DIT whose stage twiddles are
//     (shift_j^A w_B^off)^(2^s) instead of w_B^(off 2^s) evaluates on the shifted coset directly (P(z) = Pe(z^2) + z Po(z^2)
//     with z = shift_j^A w^off), so the first forward stage works straight from the kept coefficients as a lazy butterfly
//     with one wave-uniform twiddle; what is left of the scaling, the row's constant shift_j^k1 / n, rides on the store twiddle;
//   * the last forward pass (again i = g + 512 j) goes from registers through the w_n^(j0 k1) twiddle to HBM.
// Everything a block needs that depends only on its row k1 — w_n^(+-k1 tid), w_n^(+-512 k1), shift_j^k1 / n — comes from
// small tables (two multiplications per lookup) instead of square-and-multiply chains run redundantly by every thread:
// those chains were ~76 of the kernel's 336 VALU instructions per point.
__global__ __launch_bounds__(THREADS, 6) void lde_rows_big(const Batch* __restrict__ d) {
  constexpr uint32_t LB = LOG_ROW_MAX, B = 1u << LB, BP = B + (B >> 5);
  constexpr uint32_t G = B / THREADS;  // points per thread (16)
  static_assert(G == 16 && LB == 13, "pass structure below is written for 8192 points on 512 threads");
  extern __shared__ uint32_t lds[];
  uint32_t* work = lds;          // BP
  uint32_t local;
  const Group& g = find_group(d, K_ROWS_BIG, local);
  const uint32_t la = g.la;
  const uint32_t tid = threadIdx.x;
  // column fastest: the blocks in flight at one time share their row position, hence the row's twiddle table (32 KiB, L2-resident)
  const uint32_t pr = local / g.n_cols;
  const uint32_t col = local - pr * g.n_cols;
  const Mat& m = find_mat(d, g, col);
  const size_t c = col - m.col0;
  const size_t n = (size_t)B << la;
  const uint32_t nmask = (uint32_t)n - 1;
  const uint32_t k1 = kb::bitrev(pr, la);
  if (gp::load(m.cflag + 2 * c) == 0) return;   // a constant column: nothing to transform (lde_cols<true> fills the output), whole block
  const uint32_t* src = m.tmp1 + c * n + (size_t)pr * B;
  const uint32_t* __restrict__ pw_lo = g.pw_lo;
  const uint32_t* __restrict__ pw_hi = g.pw_hi;
  auto pw = [&](uint32_t e) { return kb::mul(gp::load(pw_lo + (e & 1023)), gp::load(pw_hi + (e >> 10))); };  // w_n^e, e < n

  // inverse row transform, natural in -> bit-reversed out, with the row's table of sub-problem twiddles (butterflies_ct): the
  // w_n^(-i0 k1) twiddle of the four-step decomposition is inside the table. First four stages in registers (thread g owns i0 = g + 512 j:
  // the sub-problem index of those stages is made of j's top bits, so their fifteen twiddles are the same for the whole block: scalar
  // loads), two passes through LDS, last stage (pairs 2g, 2g+1) into the registers where the coefficients stay for every coset.
  const uint32_t* __restrict__ twr = g.tw_rows + (size_t)k1 * B;
  uint32_t x[16];
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) x[j] = gp::load(src + tid + (j << 9));
  butterflies_ct<4, true>(x, 0, 0, twr);
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) work[phys<true>(tid + (j << 9))] = x[j];
  __syncthreads();
  ntt_pass_ct<4, true>(work, LB, 4, twr);
  ntt_pass_ct<4, true>(work, LB, 8, twr);
  // coef[pc] = n * c_kk with kk = bitrev_13(pc) * A + k1, as signed unreduced words (the forward passes are lazy as well)
  uint32_t keep[16];
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t gg = tid + (k << 9);
    const int32_t a = (int32_t)work[phys<true>(2 * gg)], b = (int32_t)work[phys<true>(2 * gg + 1)];
    const uint32_t w = gp::load(twr + (1u << 12) - 1 + gg);
    const int64_t ar = kb::mad_i64_i32_uniform(a, kb::ONE, 0);
    keep[2 * k] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(b, (int32_t)w, ar));
    keep[2 * k + 1] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(-b, (int32_t)w, ar));
  }
  // store twiddle w_n^(k1 (tid + 512 j)): first factor and step
  const uint32_t st0 = pw((k1 * tid) & nmask), st_step = pw((k1 << 9) & nmask);
  const uint32_t ncosets = 1u << d->log_blowup;
  for (uint32_t j = 0; j < ncosets; j++) {
    const uint32_t* __restrict__ tw_fwd = m.twf + (size_t)j * B;  // stage twiddles (shift_j^A w_B^off)^(2^s)
    const uint32_t t_first = gp::load(tw_fwd + B - 2);                       // stage 12 (span 1): shift_j^(A B / 2)
    const uint32_t cj = gp::load(m.cs + (j << la) + k1);                     // shift_j^k1 / n
    __syncthreads();  // every reader of `work` from the previous step is done
    // The LDS/global addresses below are the same for every coset; left alone the optimiser hoists all of them out
    // of this loop and the kernel needs 160 VGPRs. An opaque copy of the thread index keeps them loop-local.
    uint32_t lt = tid;
    asm volatile("" : "+v"(lt));
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t gg = lt + (k << 9);
      // first forward stage (span 1) as a lazy butterfly: a + t b, a - t b, kept as signed words from here to the store
      const int64_t ar = kb::mad_i64_i32_uniform((int32_t)keep[2 * k], kb::ONE, 0);
      work[phys<true>(2 * gg)] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32_uniform((int32_t)keep[2 * k + 1], t_first, ar));
      work[phys<true>(2 * gg + 1)] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32_uniform((int32_t)keep[2 * k + 1], 0u - t_first, ar));
    }
    __syncthreads();
    ntt_pass<false, 4, true, true>(work, LB, 0, 1, 8, tw_fwd, lt);
    ntt_pass<false, 4, true, true>(work, LB, 0, 1, 4, tw_fwd, lt);
#pragma unroll
    for (uint32_t q = 0; q < 16; q++) x[q] = work[phys<true>(lt + (q << 9))];
    butterflies<false, 4, true>(x, B, 0, lt, 9, tw_fwd);
    uint32_t* dst = m.tmp2 + (size_t)j * n * m.w + c * n + (size_t)k1 * B;  // row k1: natural order for lde_cols<true>
    uint32_t t = kb::mul(st0, cj);
#pragma unroll
    for (uint32_t q = 0; q < 16; q++) {
      {  // x t with both factors signed words; the one correction brings the product back to [0, p)
        const uint32_t r = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32((int32_t)x[q], (int32_t)t, 0));
        gp::store(dst + lt + (q << 9), min(r, r + kb::P));
      }
      // the running twiddle stays a signed unreduced word (four instructions, st_step is block-uniform): in units of 2^31
      // T <- 0.4961 T + 0.4961 <= 0.985, so |x t| / 2^32 + p / 2 < p and one correction is enough
      t = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32_uniform((int32_t)t, st_step, 0));
    }
  }
}

// Stage-major twiddles of the size-2^logn transform with root w: tw[n - (n >> s) + off] = w^(off << s)
// for stage s = blockIdx.y and off < n >> (s + 1).
__global__ void fill_stage_twiddles(uint32_t* tw, uint32_t w, int logn) {
  const size_t n = (size_t)1 << logn;
  const int s = blockIdx.y;
  size_t off = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (off < (n >> (s + 1))) tw[n - (n >> s) + off] = kb::pow(w, (uint64_t)off << s);
}

// The same table with a coset shift folded in: tw[n - (n >> s) + off] = (g w^off)^(2^s) = g^(2^s) w^(off << s). A forward DIT run
// with it evaluates the polynomial on g <w> instead of <w>: the sub-transform of stage s works in the variable y = z^(2^s), and
// P(y) = Pe(y^2) + y Po(y^2) makes y itself the butterfly's twiddle.
__global__ void fill_scaled_stage_twiddles(uint32_t* tw, uint32_t w, int logn, uint32_t g) {
  const size_t n = (size_t)1 << logn;
  const int s = blockIdx.y;
  size_t off = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (off < (n >> (s + 1))) {
    uint32_t gs = g;
    for (int i = 0; i < s; i++) gs = kb::sqr(gs);
    tw[n - (n >> s) + off] = kb::mul(gs, kb::pow(w, (uint64_t)off << s));
  }
}
// Sub-problem twiddles of a plain size-2^logn transform run as butterflies_ct: tw[(1 << s) - 1 + r] = w^(bitrev_s(r) * (n >> (s + 1)))
__global__ void fill_ct_twiddles(uint32_t* tw, uint32_t w, int logn) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = 1u << logn;
  if (e + 1 >= n) { if (e + 1 == n) tw[e] = 0; return; }
  const int s = 31 - __clz(e + 1);
  const uint32_t r = e + 1 - (1u << s);
  tw[e] = kb::pow(w, (uint64_t)kb::bitrev(r, s) * (n >> (s + 1)));
}
// Per row k1 < A of the four-step decomposition, the sub-problem twiddles of the inverse B-point row transform with the w_n^(-i0 k1)
// twiddle folded in (butterflies_ct): tw[k1 B + (1 << s) - 1 + r] = (g w^bitrev_s(r))^(B >> (s + 1)) with g = w_n^-k1, w = w_B^-1 = w_n^-A,
// i.e. w_n^(-(B >> (s + 1)) (k1 + A bitrev_s(r))). One thread per entry; w_n^e through the two-level power table.
__global__ void fill_row_twiddles(uint32_t* tw, int la, int lb, const uint32_t* __restrict__ pw_lo, const uint32_t* __restrict__ pw_hi) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t B = (size_t)1 << lb, n = B << la;
  if (idx >= n) return;
  const uint32_t k1 = (uint32_t)(idx >> lb), e = (uint32_t)(idx & (B - 1));
  if (e == B - 1) { tw[idx] = 0; return; }        // B - 1 entries per row
  const int s = 31 - __clz(e + 1);
  const uint32_t r = e + 1 - (1u << s);
  const uint64_t ex = ((uint64_t)(B >> (s + 1)) * ((uint64_t)k1 + ((uint64_t)kb::bitrev(r, s) << la))) & (n - 1);
  const uint32_t en = (uint32_t)((n - ex) & (n - 1));
  tw[idx] = kb::mul(pw_lo[en & 1023], pw_hi[en >> 10]);
}
// lo[i] = w^i (i < 1024), hi[i] = w^(1024 i) (i < n_hi): w^e = lo[e & 1023] * hi[e >> 10]
__global__ void fill_pow_tables(uint32_t* lo, uint32_t* hi, uint32_t w, uint32_t n_hi) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 1024) lo[i] = kb::pow(w, (uint64_t)i);
  if (i < n_hi) hi[i] = kb::pow(w, (uint64_t)i << 10);
}
// cs[j A + k1] = (shift w_N^j)^k1 / n for j < cosets, k1 < A
__global__ void fill_row_scales(uint32_t* cs, uint32_t shift, uint32_t w_N, uint32_t n_inv, uint32_t A, uint32_t cosets) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A * cosets) return;
  const uint32_t j = i / A, k1 = i % A;
  cs[i] = kb::mul(n_inv, kb::pow(kb::mul(shift, kb::pow(w_N, (uint64_t)j)), (uint64_t)k1));
}

}  // namespace lde
