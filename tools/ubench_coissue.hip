// Do the two VALU rates of gfx950 (≈2.5 cycles per wave-instruction for add/and/mov/f32, ≈4.3 for f64 / multiplies / shifts / min)
// belong to pipes that can work at the same time? Waves of one SIMD run either an f64 chain, a simple-integer chain, or both
// (different waves, or interleaved in one wave); if the pipes were independent the mixed runs would take max(), not the sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coissue.hip -o tools/ubench_coissue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096, ILP = 8;

// MODE 0: every wave f64 fma; 1: every wave v_add_u32; 2: even waves f64, odd waves add; 3: every wave both, interleaved;
// 4: every wave v_mul_lo_u32; 5: even waves f64, odd waves mul_lo; 6: every wave f64 + mul_lo interleaved; 7: add + mul_lo interleaved
template <int MODE>
__global__ void bench(uint32_t* out, uint32_t seed) {
  uint32_t a[ILP], b = seed | 1;
  double f[ILP]; double g = (double)(seed | 1) * 1.0000001;
  for (int i = 0; i < ILP; i++) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; f[i] = (double)a[i]; }
  const int wave = threadIdx.x >> 6;
  const bool do_f = MODE == 0 || MODE == 3 || MODE == 6 || ((MODE == 2 || MODE == 5) && (wave & 1) == 0);
  const bool do_i = MODE == 1 || MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7 || ((MODE == 2 || MODE == 5) && (wave & 1) == 1);
  if (MODE == 3 || MODE == 6 || MODE == 7) {
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) {
        if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %2, %0\n v_add_u32 %1, %1, %3" : "+v"(f[i]), "+v"(a[i]) : "v"(g), "v"(b));
        if (MODE == 6) asm volatile("v_fma_f64 %0, %0, %2, %0\n v_mul_lo_u32 %1, %1, %3" : "+v"(f[i]), "+v"(a[i]) : "v"(g), "v"(b));
        if (MODE == 7) { uint32_t& c = a[(i + 4) % ILP]; (void)c; asm volatile("v_add_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(a[i]), "+v"(b) : "v"(a[(i + 1) % ILP])); }
      }
    }
  } else if (do_f) {
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[i]) : "v"(g));
    }
  } else if (do_i) {
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) {
        if (MODE == 4 || MODE == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      }
    }
  }
  uint32_t r = b;
  for (int i = 0; i < ILP; i++) r += a[i] + (uint32_t)(long long)f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char* name, uint32_t* d) {
  const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  bench<MODE><<<blocks, threads>>>(d, 12345);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) bench<MODE><<<blocks, threads>>>(d, 12345 + r);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-58s %8.3f ms\n", name, ms);
  return 0;
}

int main() {
  uint32_t* d;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  run<0>("all 8 waves/SIMD: v_fma_f64", d);
  run<1>("all 8 waves/SIMD: v_add_u32", d);
  run<2>("4 waves v_fma_f64 + 4 waves v_add_u32 (half of each)", d);
  run<3>("all waves: v_fma_f64 and v_add_u32 interleaved (all of both)", d);
  run<4>("all 8 waves/SIMD: v_mul_lo_u32", d);
  run<5>("4 waves v_fma_f64 + 4 waves v_mul_lo_u32 (half of each)", d);
  run<6>("all waves: v_fma_f64 and v_mul_lo_u32 interleaved (all of both)", d);
  run<7>("all waves: v_add_u32 and v_mul_lo_u32 interleaved (all of both)", d);
  return 0;
}
