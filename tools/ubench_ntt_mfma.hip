// Experiment (round 3, DESIGN.md section 3 "LDE"): a radix-16 NTT step on the matrix cores.
//
// Every kernel of the proof is bound by VALU issue while the MFMA pipe idles. A radix-16 step of an NTT is a 16 x 16 constant matrix
// (w_16^(ij)) times 16 points; on 8-bit limbs that is integer MFMA work: v_mfma_i32_16x16x64_i8 computes D(16x16) += A(16x64) B(64x16),
// and with K = (point j, byte l) the B operand is simply the 16 words of a transform as they lie in registers (lane 16 kb + n holds
// points 4 kb .. 4 kb + 3 of transform n: four VGPRs = sixteen bytes), byte-wise:
//     y_i = sum_j W_ij x_j = sum_s 2^(8 s) * [ sum_{j, l} Wd_{ij, s-l} * xb_{j, l} ]        s = 0 .. 6
// with Wd the balanced base-256 digits of W_ij (as a Montgomery word, so that one Montgomery reduction at the end leaves a Montgomery
// word) and xb the bytes of x_j: seven MFMAs, A^(s)[i][(j, l)] = Wd_{ij, s-l}. i8 is signed: the low three bytes of a word are made
// signed by xor 0x80 (x - 128) and the constant 128 * sum Wd that this leaves out goes into the accumulator input; the top byte of a
// lazy word (|v| < 2^31) is a signed byte as it stands. The seven partial sums (|D_s| <= 2^20) are recombined on the VALU:
//     E0 = D0 + (D1 << 8), E1 = D2 + (D3 << 8), E2 = D4 + (D5 << 8), E3 = D6;   T = E0 + E1 2^16 + E2 (2^32 mod p) + E3 (2^48 mod p)
// and y = monty_reduce_signed(T): ten instructions per point, then the inter-step twiddle (lazy product: 4) and the xor (1): 15 VALU
// instructions per point and radix-16 step against the 20 of four stages of lazy radix-2 butterflies, plus 7 MFMAs per 256 points on
// the other pipe. The output fragment has the input fragment's layout (lane 16 g + n holds rows 4 g .. 4 g + 3 of column n), so steps
// chain without data movement.
//
// This file checks the MFMA step against a scalar DFT and times both formulations in registers.
// (-amdgpu-mfma-vgpr-form: results in VGPRs; by default they land in AGPRs and cost a v_accvgpr_read each)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -I ziren_amd/csrc tools/ubench_ntt_mfma.hip -o tools/ubench_ntt_mfma && tools/ubench_ntt_mfma
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "kb31.cuh"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int REPS = 256;

struct Consts {
  v4i a[7][64];   // A operand of MFMA s for lane l
  v4i c[7][64];   // accumulator input (the 128 * sum Wd correction) of MFMA s for lane l
  uint32_t tw[16];  // inter-step twiddles (any field elements: the timing loop multiplies by them)
};

// one radix-16 step on the wave's 16 transforms: x[q] = point 4 kb + q of transform n (lane = 16 kb + n), lazy signed words in, lazy out
template <int NM = 7>
__device__ __forceinline__ void step_mfma(uint32_t (&x)[4], const v4i (&a)[7], const v4i (&c)[7], uint32_t c32, uint32_t c48) {
  v4i b;
#pragma unroll
  for (int q = 0; q < 4; q++) b[q] = (int)(x[q] ^ 0x00808080u);
  v4i d[7];
#pragma unroll
  for (int s = 0; s < 7; s++) d[s] = s < NM ? __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b, c[s], 0, 0, 0) : (s == NM ? b : c[s]);   // NM < 7: timing probe
#pragma unroll
  for (int r = 0; r < 4; r++) {   // row 4 g + r of column n
    const int32_t e0 = d[0][r] + (d[1][r] << 8), e1 = d[2][r] + (d[3][r] << 8), e2 = d[4][r] + (d[5][r] << 8), e3 = d[6][r];
    int64_t t = kb::mad_i64_i32_uniform(e0, 1u, 0);
    t = kb::mad_i64_i32_uniform(e1, 65536u, t);
    t = kb::mad_i64_i32_uniform(e2, c32, t);
    t = kb::mad_i64_i32_uniform(e3, c48, t);
    x[r] = (uint32_t)kb::monty_reduce_signed(t);
  }
}

__global__ __launch_bounds__(256) void check_mfma(const Consts* k, const uint32_t* in, uint32_t* out, uint32_t c32, uint32_t c48) {
  const int lane = threadIdx.x & 63;
  v4i a[7], c[7];
  for (int s = 0; s < 7; s++) { a[s] = k->a[s][lane]; c[s] = k->c[s][lane]; }
  uint32_t x[4];
  const int n = lane & 15, kb_ = lane >> 4;
  for (int q = 0; q < 4; q++) x[q] = in[n * 16 + 4 * kb_ + q];
  step_mfma(x, a, c, c32, c48);
  for (int r = 0; r < 4; r++) out[n * 16 + 4 * kb_ + r] = x[r];   // row i = 4 g + r of transform n
}

// timing: 4 independent tiles per wave (16 points per lane, as the VALU kernel), REPS x (step + twiddle)
template <int NM>
__global__ __launch_bounds__(256) void time_mfma(const Consts* k, uint32_t* out, uint32_t c32, uint32_t c48) {
  const int lane = threadIdx.x & 63;
  v4i a[7], c[7];
  for (int s = 0; s < 7; s++) { a[s] = k->a[s][lane]; c[s] = k->c[s][lane]; }
  uint32_t x[4][4], tw[4];
  for (int t = 0; t < 4; t++)
    for (int q = 0; q < 4; q++) x[t][q] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + t * 977u + q * 7919u) % kb::P;
  for (int q = 0; q < 4; q++) tw[q] = k->tw[(4 * (lane >> 4) + q) & 15];
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      step_mfma<NM>(x[t], a, c, c32, c48);
#pragma unroll
      for (int q = 0; q < 4; q++)   // inter-step twiddle, lazy
        x[t][q] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32((int32_t)x[t][q], (int32_t)tw[q], 0));
    }
#if defined(PIPELINE)
    // order for the scheduler: tile 0's MFMAs, then every further MFMA between eight VALU instructions of the tile before
    if (NM == 7) {
#pragma unroll
      for (int i = 0; i < 7; i++) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
      for (int i = 0; i < 21; i++) { __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
    }
#endif
  }
  uint32_t acc = 0;
  for (int t = 0; t < 4; t++) for (int q = 0; q < 4; q++) acc ^= x[t][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// the matrix pipe alone: 28 MFMAs per repetition on four independent accumulator chains
__global__ __launch_bounds__(256) void time_mfma_only(const Consts* k, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  v4i a[7], d[4];
  for (int s = 0; s < 7; s++) a[s] = k->a[s][lane];
  for (int t = 0; t < 4; t++) d[t] = k->c[t][lane];
  const v4i b = k->a[3][lane ^ 5];
  for (int r = 0; r < REPS; r++)
#pragma unroll
    for (int s = 0; s < 7; s++)
#pragma unroll
      for (int t = 0; t < 4; t++) d[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[s], b, d[t], 0, 0, 0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = d[0][0] ^ d[1][1] ^ d[2][2] ^ d[3][3];
}
// the same amount of transform work on the VALU: 16 points per lane, four stages of lazy radix-2 butterflies with per-butterfly twiddles
__global__ __launch_bounds__(256) void time_valu(const Consts* k, uint32_t* out) {
  uint32_t x[16], tw[16];
  for (int j = 0; j < 16; j++) { x[j] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + j * 7919u) % kb::P; tw[j] = k->tw[(j + (threadIdx.x & 3)) & 15]; }
  for (int r = 0; r < REPS; r++) {
#pragma unroll
    for (int q = 3; q >= 0; q--) {
      const int half = 1 << (3 - q);
#pragma unroll
      for (int j0 = 0; j0 < 16; j0++) {
        if (j0 & half) continue;
        const int j1 = j0 + half;
        const int32_t a_ = (int32_t)x[j0], b_ = (int32_t)x[j1];
        const uint32_t w = tw[(j0 & (half - 1)) + half - 1];
        const int64_t ar = kb::mad_i64_i32_uniform(a_, kb::ONE, 0);
        x[j0] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(b_, (int32_t)w, ar));
        x[j1] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32(-b_, (int32_t)w, ar));
      }
    }
  }
  uint32_t acc = 0;
  for (int j = 0; j < 16; j++) acc ^= x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static uint64_t powmod(uint64_t b, uint64_t e) { uint64_t r = 1; b %= kb::P; while (e) { if (e & 1) r = r * b % kb::P; b = b * b % kb::P; e >>= 1; } return r; }

int main() {
  const uint64_t P = kb::P, R = (1ull << 32) % P;
  const uint64_t w16 = powmod(3, (P - 1) / 16);
  // balanced base-256 digits of W_ij R mod p
  static int8_t wd[16][16][4];
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      int64_t v = (int64_t)(powmod(w16, (uint64_t)i * j) * R % P);
      for (int m = 0; m < 4; m++) {
        int64_t dgt = ((v + 128) & 255) - 128;
        if (m == 3) dgt = v;            // the top digit takes what is left (|.| <= 127 because v < 2^31)
        wd[i][j][m] = (int8_t)dgt;
        v = (v - dgt) >> 8;
      }
    }
  static Consts hk;
  for (int s = 0; s < 7; s++)
    for (int lane = 0; lane < 64; lane++) {
      // A: lane = i + 16 kb holds K = 16 kb + t, t = 0..15  <->  (j = 4 kb + t / 4, l = t % 4)
      const int i = lane & 15, kbk = lane >> 4;
      int8_t bytes[16];
      for (int t = 0; t < 16; t++) {
        const int j = 4 * kbk + t / 4, l = t % 4, m = s - l;
        bytes[t] = (m >= 0 && m < 4) ? wd[i][j][m] : 0;
      }
      memcpy(&hk.a[s][lane], bytes, 16);
      // C: lane = n + 16 g holds rows 4 g + r: 128 * sum over j and the three low bytes l of Wd_{ij, s-l}
      const int g = lane >> 4;
      for (int r = 0; r < 4; r++) {
        const int row = 4 * g + r;
        int32_t corr = 0;
        for (int j = 0; j < 16; j++)
          for (int l = 0; l < 3; l++) { const int m = s - l; if (m >= 0 && m < 4) corr += wd[row][j][m]; }
        hk.c[s][lane][r] = 128 * corr;
      }
    }
  for (int j = 0; j < 16; j++) hk.tw[j] = (uint32_t)(powmod(7, 1000 + j) * R % P);
  const uint32_t c32 = (uint32_t)((1ull << 32) % P), c48 = (uint32_t)(((1ull << 48) % P));
  Consts* dk;
  CHECK(hipMalloc(&dk, sizeof hk));
  CHECK(hipMemcpy(dk, &hk, sizeof hk, hipMemcpyHostToDevice));
  // ---- correctness: 16 transforms of 16 points (Montgomery words, including negative lazy words) against the scalar DFT
  std::vector<uint32_t> in(256), out(256);
  uint64_t sd = 99;
  for (int i = 0; i < 256; i++) {
    sd = sd * 6364136223846793005ull + 1442695040888963407ull;
    uint32_t v = (uint32_t)((sd >> 33) % P);
    if (i % 5 == 0) v = (uint32_t)(-(int32_t)(v / 2));     // a negative lazy word, congruent to p - v / 2
    if (i % 17 == 0) v = (uint32_t)P - 1;
    in[i] = v;
  }
  uint32_t *din, *dout;
  CHECK(hipMalloc(&din, 1024)); CHECK(hipMalloc(&dout, 1024));
  CHECK(hipMemcpy(din, in.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(check_mfma, dim3(1), dim3(64), 0, 0, dk, din, dout, c32, c48);
  CHECK(hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost));
  const uint64_t Rinv = powmod(R, P - 2);
  long bad = 0;
  for (int n = 0; n < 16; n++)
    for (int i = 0; i < 16; i++) {
      uint64_t acc = 0;
      for (int j = 0; j < 16; j++) {
        const int64_t xs = (int32_t)in[n * 16 + j];
        const uint64_t xc = (uint64_t)(((xs % (int64_t)P) + (int64_t)P) % (int64_t)P) * Rinv % P;   // canonical value of the (lazy) Montgomery word
        acc = (acc + powmod(w16, (uint64_t)i * j) * xc) % P;
      }
      const int64_t ys = (int32_t)out[n * 16 + i];
      const uint64_t yc = (uint64_t)(((ys % (int64_t)P) + (int64_t)P) % (int64_t)P) * Rinv % P;
      bad += yc != acc;
    }
  printf("radix-16 step on v_mfma_i32_16x16x64_i8 vs scalar DFT: %ld mismatches of 256\n", bad);
  // ---- timing
  const int blocks = 256 * 16;
  uint32_t* dres;
  CHECK(hipMalloc(&dres, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms_m = 0, ms_v = 0, ms_1 = 0, ms_0 = 0, ms_o = 0;
  for (int it = 0; it < 3; it++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(time_mfma<7>, dim3(blocks), dim3(256), 0, 0, dk, dres, c32, c48);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_m, e0, e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(time_mfma<1>, dim3(blocks), dim3(256), 0, 0, dk, dres, c32, c48);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_1, e0, e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(time_mfma<0>, dim3(blocks), dim3(256), 0, 0, dk, dres, c32, c48);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_0, e0, e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(time_mfma_only, dim3(blocks), dim3(256), 0, 0, dk, dres);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_o, e0, e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(time_valu, dim3(blocks), dim3(256), 0, 0, dk, dres);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_v, e0, e1));
  }
  const double steps = (double)blocks * 256 * 16 * REPS;   // point-steps (one point through one radix-16 step)
  printf("radix-16 step incl. twiddle, %d reps, %d threads x 16 points:\n", REPS, blocks * 256);
  printf("  lazy radix-2 butterflies (VALU)     %8.3f ms  %7.1f G point-steps/s\n", ms_v, steps / ms_v / 1e6);
  printf("  7 x i8 MFMA + VALU recombination    %8.3f ms  %7.1f G point-steps/s   (%.2fx)\n", ms_m, steps / ms_m / 1e6, ms_v / ms_m);
  printf("  the same with 1 MFMA of the 7 issued     %8.3f ms   (probe: wrong values)\n", ms_1);
  printf("  the same with no MFMA issued            %8.3f ms   (probe: the 15 VALU instructions per point alone)\n", ms_0);
  const double waves_per_simd = (double)blocks * 4 / 1024;
  printf("  28 MFMAs per repetition, no VALU        %8.3f ms   = %.1f cycles per MFMA and SIMD @2.4 GHz\n", ms_o,
         ms_o * 1e-3 * 2.4e9 / (waves_per_simd * REPS * 28));
  return bad != 0;
}
