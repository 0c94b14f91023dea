// Experiment (round 4, EXPERIMENTS.md "LDE: FP64 butterflies"): the radix-16 register pass of the LDE kernels (four radix-2 stages on the
// sixteen points a thread holds) on the FP64 pipe, with the four-instruction modular product of poseidon2_f64.cuh.
//
// The integer pass (lde.cuh, butterflies<false, 4, true>) spends nine instructions per lazy butterfly: a R as a 64-bit word (one
// v_mad_i64_i32), then for each of a + w b and a - w b one multiply-add and a three-instruction signed Montgomery reduction — both
// outputs have to be reduced because an int32 word has no headroom. A double has 22 spare bits: t = w b mod p (mulmod_q: four
// instructions when w / p comes with the twiddle), a + t, a - t (two), and nothing is reduced for the whole transform (|v| grows by
// p (1/2 + 2^-6) per stage: below 2^35 after 23 stages; mulmod_q takes |w v| < 2^77). Data words are Montgomery words in both forms
// (the transform is linear: it does not care), the FP64 twiddles are the canonical values.
//
// Four kernels, the same 8192-point tile per 512-thread block as lde_rows_big, REPS passes each:
//   int        register pass only, twiddles = 15 vector loads of u32 per pass (as the LDS passes of lde.cuh)
//   f64        register pass only, twiddles = 15 vector loads of (w, w / p) as two doubles
//   f64c       register pass only, twiddles = 15 vector loads of u32, converted and multiplied by 1 / p in the pass (two more instructions per twiddle)
//   int_lds / f64_lds   the pass between an LDS read and an LDS write of the sixteen points (b32 / b64, the padded index of lde.cuh), one barrier per pass
// and a check that int and f64 agree modulo p.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ziren_amd/csrc tools/ubench_ntt_f64.hip -o tools/ubench_ntt_f64 && tools/ubench_ntt_f64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include "poseidon2_f64.cuh"
#include "gptr.cuh"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int REPS = 64;
constexpr uint32_t LB = 13, B = 1u << LB, THREADS = 512;
__device__ __forceinline__ uint32_t phys(uint32_t i) { return i + (i >> 5); }

// ---- the integer pass, as lde.cuh has it (DIT, lazy), twiddle index = stage base + lane part ----
__device__ __forceinline__ void pass_int(uint32_t (&x)[16], const uint32_t* __restrict__ tw, uint32_t lo) {
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    const uint32_t q = 3 - qq, half = 1u << (3 - q), tbase = (B - (B >> q)) + lo;
#pragma unroll
    for (uint32_t j0 = 0; j0 < 16; j0++) {
      if (j0 & half) continue;
      const uint32_t j1 = j0 + half;
      const uint32_t w = gp::load(tw + tbase + ((j0 & (half - 1)) << 9));
      const uint32_t a = x[j0], b = x[j1];
      const int64_t ar = kb::mad_i64_i32_uniform((int32_t)a, kb::ONE, 0);
      x[j0] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32((int32_t)b, (int32_t)w, ar));
      x[j1] = (uint32_t)kb::monty_reduce_signed(kb::mad_i64_i32((int32_t)b, -(int32_t)w, ar));
    }
  }
}
// ---- the FP64 pass: t = w b mod p, a + t, a - t ----
template <bool CONVERT>
__device__ __forceinline__ void pass_f64(double (&x)[16], const double2* __restrict__ twd, const uint32_t* __restrict__ twc, uint32_t lo) {
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    const uint32_t q = 3 - qq, half = 1u << (3 - q), tbase = (B - (B >> q)) + lo;
#pragma unroll
    for (uint32_t j0 = 0; j0 < 16; j0++) {
      if (j0 & half) continue;
      const uint32_t j1 = j0 + half;
      const uint32_t ti = tbase + ((j0 & (half - 1)) << 9);
      double w, wp;
      if (CONVERT) {
        w = (double)gp::load(twc + ti);
        wp = w * p2f::PINV;
      } else {
        const double2 t2 = twd[ti];
        w = t2.x; wp = t2.y;
      }
      const double a = x[j0];
      const double t = p2f::mulmod_q(x[j1], w, wp);
      x[j0] = a + t;
      x[j1] = a - t;
    }
  }
}

__global__ __launch_bounds__(THREADS) void bench_int(uint32_t* io, const uint32_t* __restrict__ tw, int reps) {
  const uint32_t tid = threadIdx.x;
  uint32_t* p = io + (size_t)blockIdx.x * B;
  uint32_t x[16];
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) x[j] = p[tid + (j << 9)];
  for (int r = 0; r < reps; r++) pass_int(x, tw, (tid + 37u * r) & 511u);
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) p[tid + (j << 9)] = x[j];
}
template <bool CONVERT>
__global__ __launch_bounds__(THREADS) void bench_f64(uint32_t* io, const double2* __restrict__ twd, const uint32_t* __restrict__ twc, int reps, double* raw) {
  const uint32_t tid = threadIdx.x;
  uint32_t* p = io + (size_t)blockIdx.x * B;
  double x[16];
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) x[j] = (double)p[tid + (j << 9)];
  for (int r = 0; r < reps; r++) {
    pass_f64<CONVERT>(x, twd, twc, (tid + 37u * r) & 511u);
    if ((r & 15) == 15) {   // every 64 stages: the additive growth (p (1/2 + 2^-6) per stage) is taken back, as a kernel boundary would
#pragma unroll
      for (uint32_t j = 0; j < 16; j++) x[j] = p2f::reduce(x[j]);
    }
  }
  if (raw) {
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) raw[(size_t)blockIdx.x * B + tid + (j << 9)] = x[j];
  }
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) {   // canonical word out: floor quotient, then the low word of r + 2^52
    const double q = p2f::rne(p2f::fma_(x[j], p2f::PINV, -0.5 + 0x1p-33));
    const double rr = p2f::fma_(-q, p2f::P, x[j]) + 0x1p52;
    p[tid + (j << 9)] = (uint32_t)__double2loint(rr);
  }
}
__global__ __launch_bounds__(THREADS) void bench_int_lds(uint32_t* io, const uint32_t* __restrict__ tw, int reps) {
  __shared__ uint32_t work[B + (B >> 5)];
  const uint32_t tid = threadIdx.x;
  uint32_t* p = io + (size_t)blockIdx.x * B;
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) work[phys(tid + (j << 9))] = p[tid + (j << 9)];
  __syncthreads();
  for (int r = 0; r < reps; r++) {
    const uint32_t logm2 = (r & 1) ? 5 : 9, lo = tid & ((1u << logm2) - 1), hi = tid >> logm2, base = (hi << (4 + logm2)) + lo;
    uint32_t x[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) x[j] = work[phys(base + (j << logm2))];
    pass_int(x, tw, lo);
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) work[phys(base + (j << logm2))] = x[j];
    __syncthreads();
  }
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) p[tid + (j << 9)] = work[phys(tid + (j << 9))];
}
__global__ __launch_bounds__(THREADS) void bench_f64_lds(uint32_t* io, const double2* __restrict__ twd, int reps) {
  extern __shared__ double workd[];   // B + B / 32 doubles = 66 KiB
  const uint32_t tid = threadIdx.x;
  uint32_t* p = io + (size_t)blockIdx.x * B;
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) workd[phys(tid + (j << 9))] = (double)p[tid + (j << 9)];
  __syncthreads();
  for (int r = 0; r < reps; r++) {
    const uint32_t logm2 = (r & 1) ? 5 : 9, lo = tid & ((1u << logm2) - 1), hi = tid >> logm2, base = (hi << (4 + logm2)) + lo;
    double x[16];
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) x[j] = workd[phys(base + (j << logm2))];
    pass_f64<false>(x, twd, nullptr, lo);
    if ((r & 15) == 15) {
#pragma unroll
      for (uint32_t j = 0; j < 16; j++) x[j] = p2f::reduce(x[j]);
    }
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) workd[phys(base + (j << logm2))] = x[j];
    __syncthreads();
  }
#pragma unroll
  for (uint32_t j = 0; j < 16; j++) {
    const double v = workd[phys(tid + (j << 9))];
    const double q = p2f::rne(p2f::fma_(v, p2f::PINV, -0.5 + 0x1p-33));
    p[tid + (j << 9)] = (uint32_t)__double2loint(p2f::fma_(-q, p2f::P, v) + 0x1p52);
  }
}

static uint64_t sm64(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main() {
  const int blocks = 256 * 8;
  const size_t n = (size_t)blocks * B;
  uint64_t seed = 7;
  std::vector<uint32_t> h(n), tw_m(B), tw_c(B);
  std::vector<double2> tw_d(B);
  for (auto& v : h) v = (uint32_t)(sm64(seed) % kb::P);
  for (uint32_t i = 0; i < B; i++) {
    const uint32_t c = (uint32_t)(sm64(seed) % kb::P);   // any field elements: the passes multiply by them
    tw_c[i] = c;
    tw_m[i] = kb::to_monty(c);
    tw_d[i].x = (double)c;
    tw_d[i].y = (double)c * p2f::PINV;
  }
  uint32_t *d_a, *d_b, *d_twm, *d_twc;
  double2* d_twd;
  double* d_raw;
  CHECK(hipMalloc(&d_a, n * 4)); CHECK(hipMalloc(&d_b, n * 4)); CHECK(hipMalloc(&d_twm, B * 4)); CHECK(hipMalloc(&d_twc, B * 4));
  CHECK(hipMalloc(&d_twd, B * sizeof(double2))); CHECK(hipMalloc(&d_raw, n * 8));
  CHECK(hipMemcpy(d_twm, tw_m.data(), B * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_twc, tw_c.data(), B * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_twd, tw_d.data(), B * sizeof(double2), hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute((const void*)bench_f64_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (B + (B >> 5)) * 8));

  // agreement modulo p: 1, 5 and 64 passes (the last one crosses the reductions of the FP64 loop), both twiddle forms
  long bad = 0;
  double maxabs = 0;
  for (int reps : {1, 5, 64}) {
    std::vector<uint32_t> ra(n), rb(n), rc(n);
    std::vector<double> raw(n);
    CHECK(hipMemcpy(d_a, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bench_int, dim3(blocks), dim3(THREADS), 0, 0, d_a, d_twm, reps);
    CHECK(hipMemcpy(ra.data(), d_a, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(d_b, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bench_f64<false>, dim3(blocks), dim3(THREADS), 0, 0, d_b, d_twd, d_twc, reps, d_raw);
    CHECK(hipMemcpy(rb.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(raw.data(), d_raw, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(d_b, h.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bench_f64<true>, dim3(blocks), dim3(THREADS), 0, 0, d_b, d_twd, d_twc, reps, (double*)nullptr);
    CHECK(hipMemcpy(rc.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
      const int64_t v = (int32_t)ra[i];
      const uint32_t canon = (uint32_t)(((v % (int64_t)kb::P) + kb::P) % kb::P);
      if (canon != rb[i] || canon != rc[i] || rb[i] >= kb::P) bad++;
      const double m = raw[i] < 0 ? -raw[i] : raw[i];
      if (m > maxabs) maxabs = m;
    }
  }
  printf("check: %ld mismatches between the integer and the FP64 passes (modulo p), largest FP64 word %.3g (2^%.1f)\n", bad, maxabs, std::log2(maxabs));

  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch(); launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int it = 0; it < 5; it++) {
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double bf = (double)n / 16 * 32 * REPS;   // butterflies
    printf("%-10s %8.3f ms   %7.2f G butterflies/s\n", name, best, bf / best / 1e6);
  };
  timeit("int", [&] { hipLaunchKernelGGL(bench_int, dim3(blocks), dim3(THREADS), 0, 0, d_a, d_twm, REPS); });
  timeit("f64", [&] { hipLaunchKernelGGL(bench_f64<false>, dim3(blocks), dim3(THREADS), 0, 0, d_b, d_twd, d_twc, REPS, (double*)nullptr); });
  timeit("f64c", [&] { hipLaunchKernelGGL(bench_f64<true>, dim3(blocks), dim3(THREADS), 0, 0, d_b, d_twd, d_twc, REPS, (double*)nullptr); });
  timeit("int_lds", [&] { hipLaunchKernelGGL(bench_int_lds, dim3(blocks), dim3(THREADS), 0, 0, d_a, d_twm, REPS); });
  timeit("f64_lds", [&] { hipLaunchKernelGGL(bench_f64_lds, dim3(blocks), dim3(THREADS), (B + (B >> 5)) * 8, 0, d_b, d_twd, REPS); });
  return bad != 0;
}
