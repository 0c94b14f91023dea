// Mixed-height Poseidon2 Merkle commitment (p3 MerkleTreeMmcs semantics) on gfx950.
//
// Replaces MerkleTreeMmcs::commit as used by TwoAdicFriPcs::commit and the FRI commit phase
// (semantics mirrored in-tree by crates/recursion/circuit/src/fri.rs:363-405 and
// crates/recursion/circuit/src/hash.rs:40-49,75-80):
//   leaf r      = PaddingFreeSponge(rate 8, overwrite) over row r of every tallest matrix, concatenated
//   parent      = Poseidon2(left || right)[0..8]
//   when a layer's length equals a shorter matrix's height: node = compress(node, hash(rows))
//
// One thread owns one row / one node: the 16-word sponge state stays in VGPRs — as sixteen doubles, the permutation
// runs on the FP64 vector pipe (poseidon2_f64.cuh) — and because the matrices are column-major, lane l of a wavefront
// reads word (column, r0 + l): every column read is one coalesced 256-byte transaction. Words are converted on the way in
// (Montgomery word -> canonical double) and digests on the way out; they are stored as 8 consecutive Montgomery words.
#pragma once
#include "poseidon2_f64.cuh"
#include "gptr.cuh"

namespace merkle {

constexpr int THREADS = 256;

// Absorb `width` columns (colptrs[g][row]) into the sponge state, 8 per permutation.
__device__ __forceinline__ void absorb_row(double s[16], const uint32_t* const* __restrict__ colptrs, int width, size_t row) {
  // the next group's eight words are requested before the current group is permuted, so their latency sits under ~3400
  // instructions of arithmetic instead of in front of them
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (i < width) w[i] = gp::load(colptrs[i] + row);
  for (int g0 = 0; g0 < width; g0 += 8) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (g0 + i < width) s[i] = p2f::load_monty(w[i]);
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (g0 + 8 + i < width) w[i] = gp::load(colptrs[g0 + 8 + i] + row);
    p2f::permute(s);
  }
}

__device__ __forceinline__ void store_digest(uint32_t* dst, const double s[16]) {
  uint4* d = reinterpret_cast<uint4*>(dst);
  d[0] = make_uint4(p2f::store_monty(s[0]), p2f::store_monty(s[1]), p2f::store_monty(s[2]), p2f::store_monty(s[3]));
  d[1] = make_uint4(p2f::store_monty(s[4]), p2f::store_monty(s[5]), p2f::store_monty(s[6]), p2f::store_monty(s[7]));
}
__device__ __forceinline__ void load_digest(double s[8], const uint32_t* src) {
  const uint4* p = reinterpret_cast<const uint4*>(src);
  uint4 a = p[0], b = p[1];
  s[0] = p2f::load_monty(a.x); s[1] = p2f::load_monty(a.y); s[2] = p2f::load_monty(a.z); s[3] = p2f::load_monty(a.w);
  s[4] = p2f::load_monty(b.x); s[5] = p2f::load_monty(b.y); s[6] = p2f::load_monty(b.z); s[7] = p2f::load_monty(b.w);
}

// layer 0: digests[r] = hash(row r of all tallest matrices)
__global__ __launch_bounds__(THREADS) void hash_leaves(const uint32_t* const* __restrict__ colptrs, int width, size_t height,
                                                       uint32_t* __restrict__ digests) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= height) return;
  double s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = 0.0;
  absorb_row(s, colptrs, width, r);
  store_digest(digests + r * 8, s);
}

// next[i] = compress(prev[2i], prev[2i+1]); optionally inject the rows of matrices of height m:
// node = compress(node, hash(row i)). All of a node's permutations (1, or 1 + ceil(w/8) + 1) go through ONE call site in a
// uniform loop — three inlined copies of the permutation would be 64 KiB of code, the size of the instruction cache.
// `prefix` (may be null; written by sponge_prefix below): prefix[16] = P, the number of leading eight-column groups of the injected
// row whose columns are constant over all rows, prefix[0..16) = the sponge state after absorbing them (Montgomery words) — the same
// for every row, so a node's sponge starts from it and absorbs the groups from P on.
__global__ __launch_bounds__(THREADS) void compress_layer(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next, size_t m,
                                                          const uint32_t* const* __restrict__ inject_cols, int inject_width,
                                                          const uint32_t* __restrict__ prefix) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double s[16], node[8];
  load_digest(s, prev + 16 * i);
  load_digest(s + 8, prev + 16 * i + 8);
  const int skip = prefix ? (int)gp::load(prefix + 16) : 0;
  const int groups = (inject_width + 7) / 8 - skip;   // the groups this node absorbs itself
  const int last = inject_width > 0 ? groups + 1 : 0;
  inject_cols += 8 * skip;
  inject_width -= 8 * skip;
  uint32_t w[8];  // the injected row's next eight words, requested one permutation ahead
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k < inject_width) w[k] = gp::load(inject_cols[k] + i);
  for (int ph = 0;; ph++) {
    p2f::permute(s);
    if (ph == last) break;
    if (ph == 0) {  // the node's digest is set aside, the state becomes the sponge of the injected row
#pragma unroll
      for (int k = 0; k < 8; k++) node[k] = s[k];
      if (skip) {
#pragma unroll
        for (int k = 0; k < 16; k++) s[k] = p2f::load_monty(gp::load(prefix + k));
      } else {
#pragma unroll
        for (int k = 0; k < 16; k++) s[k] = 0.0;
      }
    }
    if (ph < groups) {
      const int g0 = 8 * ph;
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (g0 + k < inject_width) s[k] = p2f::load_monty(w[k]);
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (g0 + 8 + k < inject_width) w[k] = gp::load(inject_cols[g0 + 8 + k] + i);
    } else {  // compress(node, row hash): both halves stay unreduced doubles (|.| < 2^35.3: permute's input bound)
#pragma unroll
      for (int k = 0; k < 8; k++) { s[8 + k] = s[k]; s[k] = node[k]; }
    }
  }
  store_digest(next + 8 * i, s);
}

// ---- the rows of every shorter height hashed up front (round 5) --------------------------------------------------------------------
// compress_layer with injection runs a node's whole sponge (up to ~60 permutations for the benchmarked shard's 2^19 layer) in the thread
// that also compresses the node: 128 registers, four waves per SIMD, and a layer of 2^18 - 2^19 nodes is one or two rounds of waves that
// live a millisecond — its launches run at 86 - 93 % VALU busy where hash_leaves (36 registers, 6.6 waves) runs at 96.5 %
// (profiles/r05_compress_layer_dispatches.csv). The sponge of an injected row does not depend on the tree below it, only the last
// compression does: hash_rows hashes the rows of ALL shorter heights of a commit in one launch (groups ordered longest row first, so
// the launch's tail is made of the short rows), compress_layer_rowdig then takes two permutations per node.
struct RowGroup {
  const uint32_t* const* cols;  // the height's column pointers (all matrices of that height, in commit order)
  const uint32_t* prefix;       // sponge_prefix's output for that height, or null
  uint32_t* out;                // [height][8] row digests (Montgomery words)
  uint32_t height;
  int width;
  uint32_t first_block;         // of this group in the launch's grid
};
__global__ __launch_bounds__(THREADS) void hash_rows(const RowGroup* __restrict__ groups, int n_groups) {
  int gi = 0;
  while (gi + 1 < n_groups && blockIdx.x >= gp::load(&groups[gi + 1].first_block)) gi++;
  const RowGroup* g = groups + gi;
  const size_t r = (size_t)(blockIdx.x - gp::load(&g->first_block)) * blockDim.x + threadIdx.x;
  if (r >= gp::load(&g->height)) return;
  const uint32_t* prefix = gp::load(&g->prefix);
  const uint32_t* const* cols = gp::load(&g->cols);
  int width = gp::load(&g->width);
  double s[16];
  if (prefix) {
    const int skip = (int)gp::load(prefix + 16);
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = p2f::load_monty(gp::load(prefix + k));
    cols += 8 * skip;
    width -= 8 * skip;   // <= 0: every column of the row is constant, the prefix state is the row's sponge
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = 0.0;
  }
  absorb_row(s, cols, width, r);
  store_digest(gp::load(&g->out) + r * 8, s);
}
// next[i] = compress(compress(prev[2i], prev[2i+1]), rowdig[i]); one call site of the permutation
__global__ __launch_bounds__(THREADS) void compress_layer_rowdig(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next, size_t m,
                                                                 const uint32_t* __restrict__ rowdig) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double s[16];
  load_digest(s, prev + 16 * i);
  load_digest(s + 8, prev + 16 * i + 8);
  const uint4* rd = reinterpret_cast<const uint4*>(rowdig + 8 * i);
  const uint4 a = rd[0], b = rd[1];   // requested a permutation ahead of their use
  for (int ph = 0;; ph++) {
    p2f::permute(s);
    if (ph == 1) break;
    s[8] = p2f::load_monty(a.x); s[9] = p2f::load_monty(a.y); s[10] = p2f::load_monty(a.z); s[11] = p2f::load_monty(a.w);
    s[12] = p2f::load_monty(b.x); s[13] = p2f::load_monty(b.y); s[14] = p2f::load_monty(b.z); s[15] = p2f::load_monty(b.w);
  }
  store_digest(next + 8 * i, s);
}

// Top of the tree in one launch: starting from a layer of 2*len0 digests at `prev`, compress down to the
// root; the layers are contiguous in memory (2*len0, len0, len0/2, ..., 1 digests). One block; a
// barrier between levels makes the freshly written layer visible to the block.
__global__ __launch_bounds__(THREADS) void compress_tail(uint32_t* __restrict__ prev, size_t len0) {
  uint32_t* p = prev;
  uint32_t* nx = prev + 16 * len0;
  for (size_t len = len0; len >= 1; len >>= 1) {
    for (size_t i = threadIdx.x; i < len; i += blockDim.x) {
      double s[16];
      load_digest(s, p + 16 * i);
      load_digest(s + 8, p + 16 * i + 8);
      p2f::permute(s);
      store_digest(nx + 8 * i, s);
    }
    __syncthreads();
    p = nx;
    nx += 8 * len;
    if (len == 1) break;
  }
}

// ---- fused leaf hashing + in-block tree (north_star: "Poseidon2 Merkle-tree commitment as a fused permutation+tree kernel") ----------
// One block hashes FUSE_LEAVES = 1024 consecutive leaves (one per thread, sponge state in VGPRs) and then reduces them through up to
// FUSE_MAX_LEVELS = 4 tree levels without going back to HBM for its inputs: a level's digests are handed over in LDS (two 32 KiB
// buffers), every level is also written to its place in the tree (openings read all of them later). With 1024 leaves the four levels
// have 512, 256, 128 and 64 nodes — whole wavefronts, so no lane idles inside a permutation; deeper in-block levels would run
// part-filled waves on an issue-bound kernel and are left to compress_layer / the lane-parallel kernels. Applies where no shorter
// matrix is injected into those levels: every FRI commit-phase tree, and commits whose next matrix is at least 2^levels shorter.
constexpr int FUSE_LEAVES = 1024, FUSE_MAX_LEVELS = 4;
// The FRI commit-phase trees (one permutation per leaf) use smaller blocks: a 1024-thread block owns its CU (70 registers x 16 waves),
// and while it walks its four tree levels with 8, 4, 2, 1 waves the CU's issue slots idle — 61 % of the permutation rate over the
// kernel. With 256 leaves per block (levels of 128 and 64 nodes: whole wavefronts again) seven blocks share a CU and one block's
// sparse levels run under the other blocks' leaves; the levels below go to compress_layer, which fills the chip.
#ifndef ZKM_FRI_FUSE_LEAVES
#define ZKM_FRI_FUSE_LEAVES 256
#endif
#ifndef ZKM_FRI_FUSE_LEVELS
#define ZKM_FRI_FUSE_LEVELS 2
#endif
constexpr int FRI_FUSE_LEAVES = ZKM_FRI_FUSE_LEAVES, FRI_FUSE_MAX_LEVELS = ZKM_FRI_FUSE_LEVELS;

template <int LEAVES>
__device__ __forceinline__ void tree_levels_in_block(double s[16], uint32_t* lds, size_t leaf0, size_t n_leaves, uint32_t* __restrict__ tree, int levels) {
  // s[0..8): this thread's leaf digest (doubles, unreduced). Layer l of the tree starts at digest index n_leaves * (2 - 2^(1-l)).
  uint32_t* cur = lds;                       // [LEAVES][8] words
  uint32_t* nxt = lds + LEAVES * 8;          // [LEAVES / 2][8]
  const int t = threadIdx.x;
  {
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = p2f::store_monty(s[k]);
    uint4* g = reinterpret_cast<uint4*>(tree + (leaf0 + t) * 8);
    g[0] = make_uint4(w[0], w[1], w[2], w[3]);
    g[1] = make_uint4(w[4], w[5], w[6], w[7]);
    uint4* l = reinterpret_cast<uint4*>(cur + t * 8);
    l[0] = make_uint4(w[0], w[1], w[2], w[3]);
    l[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  size_t layer_off = n_leaves;               // digest index where layer 1 starts
  size_t layer_len = n_leaves >> 1;
  size_t node0 = leaf0 >> 1;
  int nodes = LEAVES >> 1;
  for (int lvl = 1; lvl <= levels; lvl++) {
    __syncthreads();
    if (t < nodes) {
      const uint4* c = reinterpret_cast<const uint4*>(cur + 16 * t);
      const uint4 a0 = c[0], a1 = c[1], b0 = c[2], b1 = c[3];
      s[0] = p2f::load_monty(a0.x); s[1] = p2f::load_monty(a0.y); s[2] = p2f::load_monty(a0.z); s[3] = p2f::load_monty(a0.w);
      s[4] = p2f::load_monty(a1.x); s[5] = p2f::load_monty(a1.y); s[6] = p2f::load_monty(a1.z); s[7] = p2f::load_monty(a1.w);
      s[8] = p2f::load_monty(b0.x); s[9] = p2f::load_monty(b0.y); s[10] = p2f::load_monty(b0.z); s[11] = p2f::load_monty(b0.w);
      s[12] = p2f::load_monty(b1.x); s[13] = p2f::load_monty(b1.y); s[14] = p2f::load_monty(b1.z); s[15] = p2f::load_monty(b1.w);
      p2f::permute(s);
      uint32_t w[8];
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = p2f::store_monty(s[k]);
      uint4* g = reinterpret_cast<uint4*>(tree + (layer_off + node0 + t) * 8);
      g[0] = make_uint4(w[0], w[1], w[2], w[3]);
      g[1] = make_uint4(w[4], w[5], w[6], w[7]);
      uint4* l = reinterpret_cast<uint4*>(nxt + t * 8);
      l[0] = make_uint4(w[0], w[1], w[2], w[3]);
      l[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
    uint32_t* tmp = cur; cur = nxt; nxt = tmp;
    layer_off += layer_len;
    layer_len >>= 1;
    node0 >>= 1;
    nodes >>= 1;
  }
}

// leaves = rows of the tallest matrices (hash_leaves) + `levels` tree levels; height is a multiple of FUSE_LEAVES
__global__ __launch_bounds__(FUSE_LEAVES) void hash_leaves_tree(const uint32_t* const* __restrict__ colptrs, int width, size_t height,
                                                                uint32_t* __restrict__ tree, int levels) {
  extern __shared__ uint32_t fuse_lds[];
  const size_t leaf0 = (size_t)blockIdx.x * FUSE_LEAVES;
  double s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = 0.0;
  absorb_row(s, colptrs, width, leaf0 + threadIdx.x);
  tree_levels_in_block<FUSE_LEAVES>(s, fuse_lds, leaf0, height, tree, levels);
}

// FRI commit-phase leaves (row j = (f[2j], f[2j+1]), as hash_fri_leaves) + `levels` tree levels; m is a multiple of FRI_FUSE_LEAVES
__global__ __launch_bounds__(FRI_FUSE_LEAVES) void hash_fri_leaves_tree(const kb::E4* __restrict__ f, size_t m, uint32_t* __restrict__ tree, int levels) {
  extern __shared__ uint32_t fuse_lds[];
  const size_t leaf0 = (size_t)blockIdx.x * FRI_FUSE_LEAVES;
  const size_t j = leaf0 + threadIdx.x;
  double s[16];
  const kb::E4 a = f[2 * j], b = f[2 * j + 1];
#pragma unroll
  for (int k = 0; k < 4; k++) { s[k] = p2f::load_monty(a.c[k]); s[4 + k] = p2f::load_monty(b.c[k]); }
#pragma unroll
  for (int k = 8; k < 16; k++) s[k] = 0.0;
  p2f::permute(s);
  tree_levels_in_block<FRI_FUSE_LEAVES>(s, fuse_lds, leaf0, m, tree, levels);
}

// ---- lane-parallel Poseidon2 for the small layers near the root ------------------------------------
// A layer with few nodes cannot fill the chip with one thread per node and pays the full ~11 us latency
// of a serial permutation per level. Here 16 lanes (one DPP row) share one permutation, lane e holding
// state word e: the S-box runs on all lanes, the 4x4 MDS and the column/lane sums are DPP quad-permutes
// and row rotations. ~4x lower latency per level for ~2.3x the instructions — used only where latency rules.
namespace lanes {
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
constexpr int QUAD_ROT1 = 0x39, QUAD_ROT2 = 0x4E, QUAD_ROT3 = 0x93;  // lane q reads q+1, q+2, q+3 (mod 4)
constexpr int ROW_ROR1 = 0x121, ROW_ROR2 = 0x122, ROW_ROR4 = 0x124, ROW_ROR8 = 0x128;

__device__ __forceinline__ uint32_t external_layer(uint32_t a) {
  // circulant M4 row: 2 s_q + 3 s_{q+1} + s_{q+2} + s_{q+3}
  uint32_t b = dpp<QUAD_ROT1>(a), c = dpp<QUAD_ROT2>(a), d = dpp<QUAD_ROT3>(a);
  uint32_t t1 = kb::add(a, b), t2 = kb::add(c, d);
  uint32_t out = kb::add(kb::add(kb::add(t1, t2), t1), b);
  // + sum of the same position over the four quads
  uint32_t cs = kb::add(out, dpp<ROW_ROR4>(out));
  cs = kb::add(cs, dpp<ROW_ROR8>(cs));
  return kb::add(out, cs);
}

struct LaneConsts {
  uint32_t rc[8];  // external round constants of this lane's state word
  uint32_t diag;   // internal-layer diagonal entry
  bool lane0;
};
__device__ __forceinline__ LaneConsts load_consts(int e) {
  LaneConsts k;
#pragma unroll
  for (int r = 0; r < 8; r++) k.rc[r] = p2::d_rc_ext[r][e];
  k.diag = p2::d_diag[e];
  k.lane0 = e == 0;
  return k;
}

__device__ __forceinline__ uint32_t permute(uint32_t x, const LaneConsts& k) {
  x = external_layer(x);
#pragma unroll
  for (int r = 0; r < 4; r++) x = external_layer(p2::sbox_rc(x, k.rc[r]));
#pragma unroll 1
  for (int r = 0; r < 13; r++) {
    uint32_t y = p2::sbox_rc(x, p2::d_rc_int[r]);
    x = k.lane0 ? y : x;
    uint32_t sum = kb::add(x, dpp<ROW_ROR1>(x));
    sum = kb::add(sum, dpp<ROW_ROR2>(sum));
    sum = kb::add(sum, dpp<ROW_ROR4>(sum));
    sum = kb::add(sum, dpp<ROW_ROR8>(sum));
    x = kb::monty_reduce((uint64_t)x * k.diag + (uint64_t)sum * kb::ONE);
  }
#pragma unroll
  for (int r = 4; r < 8; r++) x = external_layer(p2::sbox_rc(x, k.rc[r]));
  return x;
}
}  // namespace lanes

// next[i] = compress(prev[2i], prev[2i+1]) with 16 lanes per node (no injection); m >= 1 nodes
__global__ __launch_bounds__(THREADS) void compress_layer_lanes(const uint32_t* __restrict__ prev, uint32_t* __restrict__ next, size_t m) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t node = t >> 4;
  const int e = threadIdx.x & 15;
  if (node >= m) return;
  lanes::LaneConsts k = lanes::load_consts(e);
  uint32_t x = lanes::permute(prev[16 * node + e], k);
  if (e < 8) next[8 * node + e] = x;
}

// ---- the duplex challenger on the device: the FRI commit phase without host round trips ---------------------------------------------
// DuplexChallenger<KoalaBear, Perm, 16, 8> (crates/recursion/circuit/src/challenger.rs:90-114,201-233; host twin: chal:: in host_ctx.hpp)
// over the same state the ABI carries (zkm_challenger), kept in device memory while the commit phase runs: the launch that finishes a
// layer's tree observes the root and samples beta right there (fri.rs:34-69: observe the commitment, sample_ext), the fold reads beta from
// memory, and the host queues all layers back to back — it replays the transcript from the roots afterwards and stays the source of truth
// (host_open.hpp fri_commit_phase). Sixteen lanes of one DPP row: lane e holds sponge word e, input word e, output word e.
struct DevChallenger {
  uint32_t sponge_state[16];
  uint32_t num_inputs;
  uint32_t input_buffer[16];
  uint32_t num_outputs;
  uint32_t output_buffer[16];
};
// rw: lane e < 8 holds root word e. Writes beta_out[0] = beta, beta_out[1] = beta^2, and the advanced state back to *ch.
__device__ __forceinline__ void observe_root_sample_beta(uint32_t rw, int e, const lanes::LaneConsts& k, DevChallenger* ch, kb::E4* beta_out) {
  uint32_t x = ch->sponge_state[e];
  uint32_t inb = ch->input_buffer[e], outb = ch->output_buffer[e];
  uint32_t ni = ch->num_inputs, no = ch->num_outputs;      // the same in every lane
  auto duplexing = [&]() {                                 // challenger.rs:90-103: overwrite the first num_inputs words, permute, eight outputs
    x = (uint32_t)e < ni ? inb : x;
    ni = 0;
    x = lanes::permute(x, k);
    outb = x;
    no = 8;
  };
#pragma unroll 1
  for (int i = 0; i < 8; i++) {                            // observe(root[i]) (:105-114)
    const uint32_t ri = (uint32_t)__shfl((int)rw, i, 16);
    no = 0;
    inb = (uint32_t)e == ni ? ri : inb;
    ni++;
    if (ni == 8) duplexing();
  }
  uint32_t b[4];
#pragma unroll 1
  for (int j = 0; j < 4; j++) {                            // sample_ext = four samples, each popped from the back (:201-233)
    if (ni != 0 || no == 0) duplexing();
    no--;
    b[j] = (uint32_t)__shfl((int)outb, (int)no, 16);
  }
  ch->sponge_state[e] = x;
  ch->input_buffer[e] = inb;
  ch->output_buffer[e] = outb;
  if (e == 0) {
    ch->num_inputs = ni;
    ch->num_outputs = no;
    const kb::E4 beta{{b[0], b[1], b[2], b[3]}};
    beta_out[0] = beta;
    beta_out[1] = kb::esqr(beta);
  }
}

// FRI commit-phase leaves with 16 lanes per leaf (row j = (f[2j], f[2j+1]): the eight words at f + 8 j; the capacity lanes start at zero): a
// layer of a few thousand leaves cannot fill the chip with one thread per leaf and pays a serial permutation's ~11 us; this way ~4 us.
__global__ __launch_bounds__(THREADS) void hash_fri_leaves_lanes(const kb::E4* __restrict__ f, size_t m, uint32_t* __restrict__ digests) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j = t >> 4;
  const int e = threadIdx.x & 15;
  if (j >= m) return;
  lanes::LaneConsts k = lanes::load_consts(e);
  const uint32_t x = lanes::permute(e < 8 ? reinterpret_cast<const uint32_t*>(f)[8 * j + e] : 0u, k);
  if (e < 8) digests[8 * j + e] = x;
}

// Root of the tree in one launch, 16 lanes per node: from 2*len0 digests at `prev` (len0 <= 64) down to 1. The root also goes straight
// to `root_host` (page-locked host memory, may be null): the transcript reads it after the stream synchronisation, no copy dispatch.
// With `ch` (an FRI commit-phase tree): the root is observed into the device challenger and beta sampled here (observe_root_sample_beta).
// (Tried and removed, round 3: ONE launch from 4096 nodes down, len0 / 64 blocks with grid barriers between levels — 72 dispatches fewer
// per proof, bit-exact, but 0.8 ms slower per SYN-22 proof: an agent-scope release is a whole-L2 write-back on gfx950, and a level is
// latency-bound at ~3 us either way.)
__global__ __launch_bounds__(1024) void compress_tail_lanes(uint32_t* __restrict__ prev, size_t len0, uint32_t* root_host, DevChallenger* ch,
                                                            kb::E4* beta_out) {
  const int e = threadIdx.x & 15;
  lanes::LaneConsts k = lanes::load_consts(e);
  uint32_t* p = prev;
  uint32_t* nx = prev + 16 * len0;
  uint32_t root_word = 0;
  for (size_t len = len0; len >= 1; len >>= 1) {
    const size_t node = threadIdx.x >> 4;
    if (node < len) {
      uint32_t x = lanes::permute(p[16 * node + e], k);
      if (e < 8) {
        nx[8 * node + e] = x;
        if (len == 1 && root_host) root_host[e] = x;
      }
      if (len == 1) root_word = x;
    }
    __syncthreads();
    p = nx;
    nx += 8 * len;
    if (len == 1) break;
  }
  if (ch && threadIdx.x < 16) observe_root_sample_beta(root_word, e, k, ch, beta_out);
}

// The same step for a tree whose root was produced by another launch (a layer of one or two leaves has no tail launch): one row of lanes.
__global__ __launch_bounds__(64) void fri_root_challenge(const uint32_t* __restrict__ root_dev, uint32_t* root_host, DevChallenger* ch, kb::E4* beta_out) {
  if (threadIdx.x >= 16) return;
  const int e = threadIdx.x;
  lanes::LaneConsts k = lanes::load_consts(e);
  const uint32_t rw = e < 8 ? root_dev[e] : 0;
  if (e < 8 && root_host) root_host[e] = rw;
  observe_root_sample_beta(rw, e, k, ch, beta_out);
}

// The part of an injected row's sponge that is the same for every row (round 4). The shape step pads a shard with chips that have no
// events: every row of such a trace is the same, its LDE is that row again, and when such matrices come first among the matrices of a
// height, every row's sponge spends its first permutations on the same words. flags[c] -> the two words lde::Mat::cflag keeps for column
// c of the concatenated row ([0] != 0: the column is not constant, [1]: its first word), or null where no flags were kept. One
// wavefront, sixteen lanes per permutation (lanes::permute): out[16] = P, the number of leading whole groups of eight constant columns
// (all groups when every column is constant, the last one possibly short), out[0..16) = the sponge state after them.
__global__ __launch_bounds__(64) void sponge_prefix(const uint32_t* const* __restrict__ flags, int width, uint32_t* __restrict__ out) {
  const int e = threadIdx.x & 15;
  int nc = 0;
  while (nc < width) {
    const uint32_t* f = flags[nc];
    if (!f || f[0] != 0) break;
    nc++;
  }
  const int P = nc == width ? (width + 7) / 8 : nc / 8;
  const lanes::LaneConsts k = lanes::load_consts(e);
  uint32_t x = 0;
  for (int g = 0; g < P; g++) {
    // the canonical word, as lde_cols<true> writes it into the LDE (an input word may be any u32 congruent to the value)
    if (e < 8 && 8 * g + e < width) { const uint32_t v = flags[8 * g + e][1]; x = kb::umin32(v, v - kb::P); }
    x = lanes::permute(x, k);
  }
  if (threadIdx.x < 16) out[e] = x;
  if (threadIdx.x == 0) out[16] = (uint32_t)P;
}

// FRI commit-phase leaves: row j = (f[2j], f[2j+1]) as 8 base words (fri.rs:279-306)
__global__ __launch_bounds__(THREADS) void hash_fri_leaves(const kb::E4* __restrict__ f, size_t m, uint32_t* __restrict__ digests) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double s[16];
  kb::E4 a = f[2 * j], b = f[2 * j + 1];
#pragma unroll
  for (int k = 0; k < 4; k++) { s[k] = p2f::load_monty(a.c[k]); s[4 + k] = p2f::load_monty(b.c[k]); }
#pragma unroll
  for (int k = 8; k < 16; k++) s[k] = 0.0;
  p2f::permute(s);
  store_digest(digests + 8 * j, s);
}

// n independent permutations (parity / micro-benchmark entry point)
__global__ __launch_bounds__(THREADS) void permute_batch(uint32_t* __restrict__ states, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
  uint4* p = reinterpret_cast<uint4*>(states + 16 * i);
  uint4 v[4] = {p[0], p[1], p[2], p[3]};
#pragma unroll
  for (int k = 0; k < 4; k++) { w[4 * k] = v[k].x; w[4 * k + 1] = v[k].y; w[4 * k + 2] = v[k].z; w[4 * k + 3] = v[k].w; }
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = p2f::load_monty(w[k]);
  p2f::permute(s);
#pragma unroll
  for (int k = 0; k < 16; k++) w[k] = p2f::store_monty(s[k]);
#pragma unroll
  for (int k = 0; k < 4; k++) p[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
// the same through the integer-pipe formulation (poseidon2.cuh): what the lane-parallel kernels near a tree's root and the
// host transcript compute with; kept as a parity entry point
__global__ __launch_bounds__(THREADS) void permute_batch_int(uint32_t* __restrict__ states, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[16];
  uint4* p = reinterpret_cast<uint4*>(states + 16 * i);
  uint4 v[4] = {p[0], p[1], p[2], p[3]};
#pragma unroll
  for (int k = 0; k < 4; k++) { s[4 * k] = v[k].x; s[4 * k + 1] = v[k].y; s[4 * k + 2] = v[k].z; s[4 * k + 3] = v[k].w; }
  p2::permute(s);
#pragma unroll
  for (int k = 0; k < 4; k++) p[k] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
}

// Proof-of-work search (DuplexChallenger::grind): the challenger state after observing the
// witness is fully determined: state[0..n_in) = pending inputs, state[n_in] = witness.
// Finds the smallest canonical witness in [base, base + total) whose sample has `bits` low zero bits.
__global__ __launch_bounds__(THREADS) void grind(const uint32_t* __restrict__ sponge_state, const uint32_t* __restrict__ inputs, int n_in,
                                                 int bits, uint32_t base, uint32_t total, unsigned int* __restrict__ best) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  uint32_t w = base + i;  // canonical witness value
  if (w >= kb::P) return;
  double s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = p2f::load_monty(sponge_state[k]);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k < n_in) s[k] = p2f::load_monty(inputs[k]);
    else if (k == n_in) s[k] = (double)w;  // the witness is given in canonical form
  }
  p2f::permute(s);
  // the sample is the last element of the rate (output buffer popped from the back)
  uint32_t v = kb::from_monty(p2f::store_monty(s[7]));
  if ((v & ((1u << bits) - 1)) == 0) atomicMin(best, w);
}

}  // namespace merkle
