// Experiment (VERDICT x1 / north_star "MFMA only for the dense MDS-matrix state mix inside Poseidon2"):
// the external MDS layer of the FP64 Poseidon2 on the matrix cores. v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4) B(4x16);
// with A = a 16x4 slice of the constant matrix circ(2 M4, M4, M4, M4) and B = four state words of sixteen permutations, four
// chained MFMAs apply the whole layer to a tile of 16 permutations. Input and output fragment maps coincide (word w of
// permutation n lives in lane 16 (w % 4) + n, register w / 4 both as B operand of slice w / 4 and as D element), so rounds
// chain with no data movement. All products and sums are exact (integers below 2^36).
// The full-round loop (16 S-boxes + layer per permutation) is timed both ways: layer as 64 v_add_f64 (VALU) vs 16 MFMAs
// per wave and layer running under other waves' S-boxes.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ziren_amd/csrc tools/ubench_mds_mfma.hip -o tools/ubench_mds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "poseidon2_f64.cuh"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double double4_ __attribute__((ext_vector_type(4)));
constexpr int ROUNDS = 256;

__device__ __forceinline__ double rc_of(int w) { return (double)((w * 2654435761u) % 1000003u) - 500000.0; }

// standard layout: one permutation per lane, sixteen words in registers
__global__ __launch_bounds__(256) void rounds_valu(double* out, int rounds) {
  double s[16];
  const int perm = blockIdx.x * blockDim.x + threadIdx.x;
  for (int w = 0; w < 16; w++) s[w] = (double)((perm * 16 + w) * 40503u % 2130706433u) - 1065353216.0;
  for (int r = 0; r < rounds; r++) {
#pragma unroll
    for (int w = 0; w < 16; w++) s[w] = p2f::sbox(s[w] + rc_of(w));
    p2f::external_layer(s);
  }
  for (int w = 0; w < 16; w++) out[(size_t)perm * 16 + w] = p2f::reduce(s[w]);
}

// MFMA layout: a wave holds 4 tiles x 16 permutations; lane l = 16 g + n holds words 4 v + g (v = 0..3) of permutation n of each tile
__global__ __launch_bounds__(256) void rounds_mfma(double* out, int rounds) {
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double t[4][4], a[4], rc[4];
  // circ(2 M4, M4, M4, M4): row i = 4 bi + ri, column k = 4 bk + rk: M4[ri][rk] * (bi == bk ? 2 : 1)
  const int M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
  for (int v = 0; v < 4; v++) {
    const int i = n, k = 4 * v + g;  // A slice v: lane holds A[i = lane % 16][k = lane / 16] = M[i][4 v + g]
    a[v] = (double)(M4[i & 3][k & 3] * ((i >> 2) == (k >> 2) ? 2 : 1));
    rc[v] = rc_of(4 * v + g);
  }
  for (int tile = 0; tile < 4; tile++)
    for (int v = 0; v < 4; v++) {
      const int perm = wave * 64 + tile * 16 + n, w = 4 * v + g;
      t[tile][v] = (double)((perm * 16 + w) * 40503u % 2130706433u) - 1065353216.0;
    }
  for (int r = 0; r < rounds; r++) {
#pragma unroll
    for (int tile = 0; tile < 4; tile++) {
#pragma unroll
      for (int v = 0; v < 4; v++) t[tile][v] = p2f::sbox(t[tile][v] + rc[v]);
    }
#pragma unroll
    for (int tile = 0; tile < 4; tile++) {
      double4_ acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int v = 0; v < 4; v++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[v], t[tile][v], acc, 0, 0, 0);
      // D: col = lane & 15 (permutation), row = (lane >> 4) + 4 reg (word): the same placement as the inputs
      t[tile][0] = acc.x; t[tile][1] = acc.y; t[tile][2] = acc.z; t[tile][3] = acc.w;
    }
  }
  for (int tile = 0; tile < 4; tile++)
    for (int v = 0; v < 4; v++) {
      const int perm = wave * 64 + tile * 16 + n, w = 4 * v + g;
      out[(size_t)perm * 16 + w] = p2f::reduce(t[tile][v]);
    }
}

int main() {
  const int blocks = 256 * 16, threads = 256;
  const size_t n = (size_t)blocks * threads * 16;
  double *da, *db;
  CHECK(hipMalloc(&da, n * 8)); CHECK(hipMalloc(&db, n * 8));
  // correctness: same states, 3 rounds, both ways
  rounds_valu<<<64, threads>>>(da, 3);
  rounds_mfma<<<64, threads>>>(db, 3);
  CHECK(hipDeviceSynchronize());
  std::vector<double> ha(64 * threads * 16), hb(64 * threads * 16);
  CHECK(hipMemcpy(ha.data(), da, ha.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hb.data(), db, hb.size() * 8, hipMemcpyDeviceToHost));
  long bad = 0;
  for (size_t i = 0; i < ha.size(); i++) {
    double d = ha[i] - hb[i];  // both reduced to (-p/2, p/2]: equal, or differ by p at the boundary
    if (d != 0.0 && d != 2130706433.0 && d != -2130706433.0) bad++;
  }
  printf("MFMA layer vs add-chain layer after 3 rounds: %zu words, %ld mismatches\n", ha.size(), bad);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int which = 0; which < 2; which++) {
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(e0));
      if (which == 0) rounds_valu<<<blocks, threads>>>(da, ROUNDS); else rounds_mfma<<<blocks, threads>>>(db, ROUNDS);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) {
        double rounds = (double)blocks * threads * ROUNDS;
        printf("%-34s %8.3f ms  %7.2f G full rounds/s  (%.0f cycles per wave-round @2.4GHz)\n", which == 0 ? "layer on VALU (64 v_add_f64)" : "layer on MFMA (16 x 16x16x4 f64)", ms,
               rounds / (ms * 1e-3) / 1e9, ms * 1e-3 * 2.4e9 * 256 * 4 / (rounds / 64));
      }
    }
  }
  return bad != 0;
}
