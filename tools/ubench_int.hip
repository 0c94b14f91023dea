// Integer-pipe micro-benchmark for gfx950: issue rate of the instructions a Montgomery multiply is
// made of. Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o ubench_int ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int OP>
__global__ void bench(uint32_t* out, uint32_t seed) {
  uint32_t a[ILP], b = seed | 1;
  uint64_t w[ILP];
  double f[ILP]; double g = (double)(seed | 1) * 1.0000001;
  for (int i = 0; i < ILP; i++) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; w[i] = a[i]; f[i] = (double)a[i]; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 7, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 7) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 12) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 13) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 15) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
      if (OP == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if (OP == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 18) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 19) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 20) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 21) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 23) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "s"(seed));
      if (OP == 24) asm volatile("v_add_u32 %0, 0x80ffffff, %0" : "+v"(a[i]));
      if (OP == 25) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 26) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 28) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 29) { uint32_t t; asm volatile("v_add_u32 %0, %0, %2\n v_subrev_u32 %1, %3, %0\n v_min_u32 %0, %0, %1" : "+v"(a[i]), "=&v"(t) : "v"(b), "s"(0x7f000001u)); }
      if (OP == 30) { uint32_t t; asm volatile("v_add_u32 %0, %0, %2\n v_subrev_co_u32 %1, vcc, %3, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]), "=&v"(t) : "v"(b), "s"(0x7f000001u) : "vcc"); }
      if (OP == 31) { uint32_t t; uint64_t m; asm volatile("v_add_u32 %0, %0, %3\n v_subrev_co_u32 %1, %2, %4, %0\n v_cndmask_b32 %0, %1, %0, %2" : "+v"(a[i]), "=&v"(t), "=&s"(m) : "v"(b), "s"(0x7f000001u)); }
      if (OP == 32) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 33) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 34) asm volatile("v_mad_i64_i32 %0, vcc, %1, 1, %0" : "+v"(w[i]) : "v"(a[i]) : "vcc");
      if (OP == 35) asm volatile("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(w[i]) : "v"(a[i]) : "vcc");
      if (OP == 36) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i+1)%ILP]));

      if (OP == 40) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g));
      if (OP == 41) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g));
      if (OP == 42) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f[i]) : "v"(g));
      if (OP == 43) asm volatile("v_rndne_f64 %0, %0" : "+v"(f[i]));
      if (OP == 44) { uint32_t t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(f[i])); a[i] ^= t; }
      if (OP == 45) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      if (OP == 46) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      if (OP == 47) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(f[(i+1)%ILP]));
      if (OP == 48) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "s"(g));
      if (OP == 49) asm volatile("v_trunc_f64 %0, %0" : "+v"(f[i]));
      if (OP == 50) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 51) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i+1)%ILP]));
      if (OP == 52) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i+1)%ILP]));
      if (OP == 53) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 54) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
      if (OP == 55) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
      if (OP == 56) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 57) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 58) asm volatile("v_ldexp_f64 %0, %0, 3" : "+v"(f[i]));
      if (OP == 59) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i+1)%ILP]));
      if (OP == 60) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 61) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 62) asm volatile("v_min_f64 %0, %0, %1" : "+v"(f[i]) : "v"(g));
      //if (OP == 63) asm volatile("v_add_nc_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 27) asm volatile("v_mul_lo_u32 %0, %0, %1\n v_add_u32 %2, %2, %1" : "+v"(a[i]), "+v"(b) , "+v"(a[(i+1)%ILP]):);
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < ILP; i++) r += a[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32) + (uint32_t)(long long)f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, uint32_t* d) {
  const int blocks = 256 * 8, threads = 256;  // 8 blocks x 4 waves per CU = 8 waves / SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  bench<OP><<<blocks, threads>>>(d, 12345);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) bench<OP><<<blocks, threads>>>(d, 12345 + r);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double insts = 5.0 * blocks * (threads / 64) * (double)ITERS * ILP;  // wave-instructions
  double per_simd_per_s = insts / (ms * 1e-3) / (256 * 4);
  printf("%-18s %8.3f ms  %7.2f Gwave-inst/s/SIMD-equiv  -> %.2f cycles/wave-inst @2.4GHz\n", name, ms, per_simd_per_s / 1e9,
         2.4e9 / per_simd_per_s);
  return 0;
}

int main() {
  uint32_t* d;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
  run<0>("v_add_u32", d);
  run<7>("v_min_u32", d);
  run<6>("v_lshl_add_u32", d);
  run<9>("v_add3_u32", d);
  run<10>("v_alignbit_b32", d);
  run<1>("v_mul_lo_u32", d);
  run<2>("v_mul_hi_u32", d);
  run<3>("v_mad_u64_u32", d);
  run<4>("v_mul_u32_u24", d);
  run<5>("v_mad_u32_u24", d);
  run<8>("v_mul_hi_u32_u24", d);
  run<11>("v_mad_u32_u16", d);
  run<12>("v_fma_f32", d);
  run<17>("v_add_f32", d);
  run<25>("v_mul_f32", d);
  run<26>("v_min_f32", d);
  run<13>("v_sub_u32", d);
  run<14>("v_and_b32", d);
  run<22>("v_xor_b32", d);
  run<15>("v_lshlrev_b32", d);
  run<16>("v_cndmask_b32", d);
  run<18>("v_min3_u32", d);
  run<19>("v_max_u32", d);
  run<20>("v_min_i32", d);
  run<21>("v_add_co_u32", d);
  run<23>("v_min_u32 sgpr", d);
  run<24>("v_add_u32 literal", d);
  run<28>("v_sub_co_u32 vcc", d);
  run<29>("modadd add/sub/min", d);
  run<30>("modadd add/sub_co/cndmask vcc", d);
  run<31>("modadd add/sub_co/cndmask sgpr", d);
  run<32>("v_mad_i64_i32", d);
  run<33>("v_mul_hi_i32", d);
  run<34>("v_mad_i64_i32 x*1+acc", d);
  run<35>("v_mad_u64_u32 x*1+acc", d);
  run<36>("v_lshl_add_u64", d);
  run<40>("v_add_f64", d);
  run<41>("v_mul_f64", d);
  run<42>("v_fma_f64", d);
  run<47>("v_fma_f64 3src", d);
  run<48>("v_add_f64 sgpr", d);
  run<43>("v_rndne_f64", d);
  run<49>("v_trunc_f64", d);
  run<44>("v_cvt_i32_f64", d);
  run<45>("v_cvt_f64_i32", d);
  run<46>("v_cvt_f64_u32", d);
  run<58>("v_ldexp_f64", d);
  run<62>("v_min_f64", d);
  run<50>("v_pk_add_u16", d);
  run<51>("v_pk_fma_f32", d);
  run<52>("v_pk_add_f32", d);
  run<53>("v_dot4_u32_u8", d);
  run<54>("v_ashrrev_i32", d);
  run<55>("v_bfe_u32", d);
  run<56>("v_perm_b32", d);
  run<57>("v_mul_i32_i24", d);
  run<59>("v_mov_b32", d);
  run<60>("v_or_b32", d);
  run<61>("v_addc_co_u32", d);
  run<0>("v_add_u32 (again)", d);
  run<7>("v_min_u32 (again)", d);
  return 0;
}
