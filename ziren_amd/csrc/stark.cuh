// LogUp permutation trace and quotient (constraint) evaluation on gfx950.
//
// Replaces, for one chip:
//   generate_permutation_trace / populate_local_permutation_row
//       crates/stark/src/permutation.rs:29-69,102-196   (caller prover.rs:337-365)
//   quotient_values + ProverConstraintFolder
//       crates/stark/src/quotient.rs:19-171, crates/stark/src/folder.rs:19-149
//
// Both are one-thread-per-row streaming kernels over column-major matrices: lane l of a
// wavefront touches row r0 + l of each column, so each column access is one coalesced
// transaction. Per-chip metadata (lookup linear forms, constraint bytecode, alpha powers) is
// wave-uniform and comes through the scalar/constant path.
//
// quotient_kernel below is the generic bytecode interpreter (register files in LDS). For chips whose
// program has been specialised (ziren_amd/codegen.py -> zkm_quotient_specialized, registered through
// zkm_ctx_register_quotient_kernel) the host launches the generated kernel instead; both share the
// prologue/epilogue of quotient_args.cuh and compute the same values.
#pragma once
#include "kb31.cuh"
#include "quotient_args.cuh"
#include "perm_args.cuh"

namespace stark {

constexpr int THREADS = 256;

// sum_i weight_i * (prep|main)[col_i][row] + constant; advances `pos` past the VirtualPairCol.
__device__ __forceinline__ uint32_t apply_pair_col(const uint32_t* __restrict__ blob, int& pos, const uint32_t* __restrict__ main,
                                                   size_t main_stride, const uint32_t* __restrict__ prep, size_t prep_stride,
                                                   size_t row) {
  int nt = blob[pos++];
  uint32_t acc = blob[pos++];
  for (int t = 0; t < nt; t++) {
    uint32_t cw = blob[pos++], weight = blob[pos++];
    uint32_t col = cw & 0x7fffffffu;
    uint32_t v = (cw >> 31) ? main[col * main_stride + row] : prep[col * prep_stride + row];
    acc = kb::add(acc, kb::mul(v, weight));
  }
  return acc;
}

// One thread per trace row. Writes the (perm_w - 1) batched columns and, in the last ext
// column, the row sum (the inclusive scan over rows runs afterwards, see scan kernels).
// perm is column-major with 4 base columns per ext column (flatten_to_base order).
__global__ __launch_bounds__(THREADS) void perm_rows(const uint32_t* __restrict__ blob, int n_lookups, int n_sends, int batch,
                                                     const uint32_t* __restrict__ main, const uint32_t* __restrict__ prep,
                                                     size_t n, kb::E4 alpha, const kb::E4* __restrict__ beta_pows,
                                                     uint32_t* __restrict__ perm, int perm_ext_w) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int pos = 2;
  kb::E4 rowsum = kb::ezero();
  int k = 0;
  // Columns go in groups of PERM_GROUP: each column's batch of lookups is summed as one fraction num / den (two extension products per
  // further term instead of an inverse), and the group's denominators are inverted with ONE base-field inversion between them
  // (einv_norm / inv_batch / einv_finish: the base inversion is 54 of an extension inverse's 74 multiplications; round 3 — a Cpu row
  // has ten of them).
  constexpr int PERM_GROUP = 2;   // 4 halves the inversions again but takes 98 registers (3 waves per SIMD: the kernel turns latency-bound and gets slower); 2 takes 50
  for (int b0 = 0; b0 < perm_ext_w - 1; b0 += PERM_GROUP) {
    // Rows whose lookups all have multiplicity zero — the padding rows behind a chip's events (62 % of the benchmarked shard's Cpu rows),
    // every row of a chip the shape step added without events, table rows nobody looked up — contribute the fraction 0 / den = 0 to
    // every column: when that holds for all 64 rows of a wavefront, the group's denominators and inversion are not computed at all
    // (round 4). The multiplicities are evaluated first (their linear forms are one or two terms); the walk over the blob is scalar.
    {
      int p2 = pos, k2 = k;
      uint32_t any_mult = 0;
      for (int g = 0; g < PERM_GROUP; g++)
        if (b0 + g < perm_ext_w - 1)
          for (int q = 0; q < batch && k2 < n_lookups; q++, k2++) {
            const int nv = blob[p2 + 1];
            p2 += 2;
            for (int v = 0; v < nv; v++) p2 += 2 + 2 * (int)blob[p2];     // skip the value forms: {terms, constant, terms x (column, weight)}
            any_mult |= apply_pair_col(blob, p2, main, n, prep, n, r);
          }
      if (__all(any_mult == 0)) {
#pragma unroll
        for (int g = 0; g < PERM_GROUP; g++)
          if (b0 + g < perm_ext_w - 1)
#pragma unroll
            for (int e = 0; e < 4; e++) perm[(size_t)(4 * (b0 + g) + e) * n + r] = 0;
        pos = p2;
        k = k2;
        continue;
      }
    }
    kb::E4 num[PERM_GROUP], den[PERM_GROUP];
    uint32_t d0[PERM_GROUP], d1[PERM_GROUP], nrm[PERM_GROUP];
#pragma unroll
    for (int g = 0; g < PERM_GROUP; g++) {
      num[g] = kb::ezero();
      den[g] = kb::eone();
      if (b0 + g < perm_ext_w - 1) {
        bool first = true;
        for (int q = 0; q < batch && k < n_lookups; q++, k++) {
          uint32_t kind = blob[pos++];
          int nv = blob[pos++];
          // alpha + beta^0 * argument_index + sum_v beta^(v+1) * value_v: the sum as four 96-bit dot products with the (row-uniform)
          // powers of beta, reduced once (8 instructions per value instead of a Montgomery product and a modular addition per coefficient)
          // (pays from four values on: the four reductions at the end cost as much as three terms the other way)
          kb::E4 denom = kb::eadd_base(alpha, kb::to_monty(kind));
          if (nv >= 4) {
            kb::FoldAcc dsum = kb::fold_zero();
            for (int v = 0; v < nv; v++) {
              uint32_t lin = apply_pair_col(blob, pos, main, n, prep, n, r);
              kb::fold_base(dsum, beta_pows[v + 1], lin);
            }
            denom = kb::eadd(denom, kb::fold_finish(dsum));
          } else {
            for (int v = 0; v < nv; v++) {
              uint32_t lin = apply_pair_col(blob, pos, main, n, prep, n, r);
              denom = kb::eadd(denom, kb::escale(beta_pows[v + 1], lin));
            }
          }
          uint32_t mult = apply_pair_col(blob, pos, main, n, prep, n, r);
          if (k >= n_sends) mult = kb::neg(mult);
          if (first) {
            num[g] = kb::efrom(mult);
            den[g] = denom;
            first = false;
          } else {
            num[g] = kb::eadd(kb::emul(num[g], denom), kb::escale(den[g], mult));
            den[g] = kb::emul(den[g], denom);
          }
        }
      }
      nrm[g] = kb::einv_norm(den[g], d0[g], d1[g]);
    }
    kb::inv_batch<PERM_GROUP>(nrm);
#pragma unroll
    for (int g = 0; g < PERM_GROUP; g++) {
      if (b0 + g < perm_ext_w - 1) {
        const kb::E4 val = kb::emul(num[g], kb::einv_finish(den[g], d0[g], d1[g], nrm[g]));
#pragma unroll
        for (int e = 0; e < 4; e++) perm[(size_t)(4 * (b0 + g) + e) * n + r] = val.c[e];
        rowsum = kb::eadd(rowsum, val);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) perm[(size_t)(4 * (perm_ext_w - 1) + e) * n + r] = rowsum.c[e];
}

// ---- inclusive prefix sum mod p over each of `ncols` columns of length n --------------------
constexpr int SCAN_BLOCK = 1024;  // elements per block (256 threads x 4)

__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* lds /*[THREADS]*/) {
  // Hillis-Steele over the block's per-thread totals; returns the inclusive prefix for this thread
  int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int d = 1; d < THREADS; d <<= 1) {
    uint32_t add = t >= d ? lds[t - d] : 0;
    __syncthreads();
    lds[t] = kb::add(lds[t], add);
    __syncthreads();
  }
  return lds[t];
}

// The running sums of all chips of a shard are scanned by the same three launches (round 3; before: three launches per chip). A job is
// one chip's last extension column (four base columns of n words, `data + e n`); blocks are dealt through the jobs' cumulative counts.
struct ScanJob { uint32_t* data; uint32_t* totals; size_t n, nchunks; uint32_t blk_end, pad; };
__device__ __forceinline__ const ScanJob& find_scan_job(const ScanJob* __restrict__ jobs, uint32_t& local) {
  uint32_t j = 0, start = 0;
  const uint32_t b = blockIdx.x;
  while (b >= jobs[j].blk_end) { start = jobs[j].blk_end; j++; }
  local = b - start;
  return jobs[j];
}
// phase 1: scan each 1024-chunk in place, emit chunk totals. blocks of a job = chunks x 4 columns
__global__ __launch_bounds__(THREADS) void scan_chunks(const ScanJob* __restrict__ jobs) {
  __shared__ uint32_t lds[THREADS];
  uint32_t local;
  const ScanJob& j = find_scan_job(jobs, local);
  const size_t n = j.n, nchunks = j.nchunks;
  const uint32_t bx = local % (uint32_t)nchunks, by = local / (uint32_t)nchunks;
  uint32_t* col = j.data + (size_t)by * n;
  size_t base = (size_t)bx * SCAN_BLOCK + (size_t)threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = base + i < n ? col[base + i] : 0;
  v[1] = kb::add(v[1], v[0]); v[2] = kb::add(v[2], v[1]); v[3] = kb::add(v[3], v[2]);
  uint32_t incl = block_inclusive_scan(v[3], lds);
  uint32_t excl = kb::sub(incl, v[3]);
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (base + i < n) col[base + i] = kb::add(v[i], excl);
  if (threadIdx.x == THREADS - 1) j.totals[(size_t)by * nchunks + bx] = incl;
}
// phase 2: exclusive scan of the chunk totals, one block per (job, column) (serial over 256-wide slabs)
__global__ __launch_bounds__(THREADS) void scan_totals(const ScanJob* __restrict__ jobs) {
  __shared__ uint32_t lds[THREADS];
  const ScanJob& j = jobs[blockIdx.x >> 2];
  const size_t nchunks = j.nchunks;
  uint32_t* t = j.totals + (size_t)(blockIdx.x & 3) * nchunks;
  uint32_t carry = 0;
  for (size_t base = 0; base < nchunks; base += THREADS) {
    size_t i = base + threadIdx.x;
    uint32_t v = i < nchunks ? t[i] : 0;
    uint32_t incl = block_inclusive_scan(v, lds);
    if (i < nchunks) t[i] = kb::add(carry, kb::sub(incl, v));
    uint32_t slab = lds[THREADS - 1];
    __syncthreads();
    carry = kb::add(carry, slab);
  }
}
// phase 3: add each chunk's offset
__global__ __launch_bounds__(THREADS) void scan_add_offsets(const ScanJob* __restrict__ jobs) {
  uint32_t local;
  const ScanJob& j = find_scan_job(jobs, local);
  const size_t n = j.n, nchunks = j.nchunks;
  const uint32_t bx = local % (uint32_t)nchunks, by = local / (uint32_t)nchunks;
  uint32_t off = j.totals[(size_t)by * nchunks + bx];
  uint32_t* col = j.data + (size_t)by * n;
  size_t base = (size_t)bx * SCAN_BLOCK + (size_t)threadIdx.x * 4;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (base + i < n) col[base + i] = kb::add(col[base + i], off);
}

// ---- quotient -------------------------------------------------------------------------------

// The table quotient_point reads: for stored row p (i = bitrev(p)), x = 3 w_Q^i:
//   is_first = Z_H(x) / (x - 1), is_last = Z_H(x) / (x - g^-1), is_transition = x - g^-1,  Z_H(x) = zh[i mod 2^lqd] (zerofier_coset.rs:22-51)
__global__ void fill_selectors(uint32_t* __restrict__ out, int lq, int lqd, uint32_t w_q, uint32_t g_inv, const uint32_t* __restrict__ consts) {
  const size_t Q = (size_t)1 << lq;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Q) return;
  const uint32_t i = kb::bitrev((uint32_t)p, lq);
  const uint32_t x = kb::mul(kb::GEN, kb::pow(w_q, (uint64_t)i));
  const uint32_t zh = consts[16 + (i & ((1u << lqd) - 1))];
  // both inverses from one
  const uint32_t xm1 = kb::sub(x, kb::ONE);
  const uint32_t is_trans = kb::sub(x, g_inv);
  const uint32_t both = kb::mul(zh, kb::inv(kb::mul(xm1, is_trans)));
  out[p] = kb::mul(both, is_trans);
  out[Q + p] = kb::mul(both, xm1);
  out[2 * Q + p] = is_trans;
}


// Register files in LDS, one slot per thread: extension registers as 16-byte words at
// regs_e[r * blockDim + t] (ds_read/write_b128, conflict-free), base registers at regs_b[r * blockDim + t].
// One thread per stored LDE row p (bit-reversed order): i = bitrev(p) is the natural index on
// the quotient coset 3 * <w_Q>; local row = p, next row = bitrev(i + 2^lqd) (quotient.rs:44-45,61).
__global__ void quotient_kernel(QuotientArgs a) {
  extern __shared__ uint4 lds4[];
  const int bd = blockDim.x, tid = threadIdx.x;
  kb::E4* regs_e = reinterpret_cast<kb::E4*>(lds4);
  uint32_t* regs_b = reinterpret_cast<uint32_t*>(lds4 + (size_t)a.n_regs * bd);
#define RE(r) regs_e[(r) * bd + tid]
#define RB(r) regs_b[(r) * bd + tid]
  QuotientPoint qp;
  if (!quotient_point(a, quotient_row(a), qp)) return;
  const size_t p = qp.p, pn = qp.pn;
  const uint32_t is_first = qp.is_first, is_last = qp.is_last, is_trans = qp.is_trans;

  kb::E4 acc = kb::ezero();
  int cidx = 0;
  // the next instruction's two words are fetched (scalar loads) while the current one executes
  uint32_t nw0 = a.n_instr ? a.program[0] : 0, nimm = a.n_instr ? a.program[1] : 0;
  for (int pc = 0; pc < a.n_instr; pc++) {
    const uint32_t w0 = nw0, imm = nimm;
    if (pc + 1 < a.n_instr) { nw0 = a.program[2 * pc + 2]; nimm = a.program[2 * pc + 3]; }
    int op = w0 & 0xff, dst = (w0 >> 8) & 0xff, ra = (w0 >> 16) & 0xff, rb = w0 >> 24;
    switch (op) {
      case 1: RB(dst) = a.main_lde[(size_t)imm * a.main_stride + (ra ? pn : p)]; break;
      case 2: RB(dst) = a.prep_lde[(size_t)imm * a.prep_stride + (ra ? pn : p)]; break;
      case 3: {
        const uint32_t* q = a.perm_lde + (size_t)(4 * imm) * a.perm_stride + (ra ? pn : p);
        RE(dst) = kb::E4{{q[0], q[a.perm_stride], q[2 * a.perm_stride], q[3 * a.perm_stride]}};
        break;
      }
      case 4: RB(dst) = imm; break;
      case 5: RB(dst) = a.public_values[imm]; break;
      case 6: RE(dst) = imm ? a.perm_beta : a.perm_alpha; break;
      case 7: RE(dst) = a.local_sum; break;
      case 8: RB(dst) = a.consts[imm]; break;
      case 9: RB(dst) = is_first; break;
      case 10: RB(dst) = is_last; break;
      case 11: RB(dst) = is_trans; break;
      case 16: RB(dst) = kb::add(RB(ra), RB(rb)); break;
      case 17: RB(dst) = kb::sub(RB(ra), RB(rb)); break;
      case 18: RB(dst) = kb::mul(RB(ra), RB(rb)); break;
      case 19: RB(dst) = kb::neg(RB(ra)); break;
      case 20: RE(dst) = kb::eadd(RE(ra), RE(rb)); break;
      case 21: RE(dst) = kb::esub(RE(ra), RE(rb)); break;
      case 22: RE(dst) = kb::emul(RE(ra), RE(rb)); break;
      case 23: RE(dst) = kb::eneg(RE(ra)); break;
      case 24: RE(dst) = kb::eadd_base(RE(ra), RB(rb)); break;
      case 25: RE(dst) = kb::esub_base(RE(ra), RB(rb)); break;
      case 26: RE(dst) = kb::escale(RE(ra), RB(rb)); break;
      case 32: acc = kb::eadd(acc, kb::escale(a.alpha_pows[cidx++], RB(ra))); break;
      case 33: acc = kb::eadd(acc, kb::emul(a.alpha_pows[cidx++], RE(ra))); break;
      default: break;
    }
  }
#undef RE
#undef RB
  quotient_store(a, qp, acc);
}

}  // namespace stark
