// Poseidon2 throughput, integer pipe vs FP64 pipe (in-register states, no HBM traffic to speak of), and a host-side
// exactness check of the FP64 formulation against the integer one.
//   hipcc --offload-arch=gfx950 -O3 -I ziren_amd/csrc tools/ubench_p2.hip -o tools/ubench_p2
//   tools/ubench_p2 host      (no GPU needed)      tools/ubench_p2 gpu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "poseidon2_f64.cuh"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static uint64_t sm64(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

static int host_check() {
  uint64_t seed = 42;
  const uint32_t ext[] = {0, 1, 2, kb::P - 1, kb::P - 2, kb::P / 2, kb::P / 2 + 1, 0x7effffff, 0x01fffffe, 0x40000000, 0x3fffffff, 0x00ffffff, 0x01000000};
  long bad = 0, n = 0;
  for (int it = 0; it < 400000; it++) {
    uint32_t a[16], b[16];
    for (int i = 0; i < 16; i++) {
      uint64_t r = sm64(seed);
      if (it < 200000) a[i] = (uint32_t)(r % kb::P);
      else a[i] = ext[r % (sizeof ext / 4)];
    }
    if (it == 0) memset(a, 0, sizeof a);
    memcpy(b, a, sizeof a);
    p2::permute_host(a);
    p2f::permute_host_words(b);
    n++;
    if (memcmp(a, b, sizeof a)) { if (bad < 3) { printf("mismatch at %d\n", it); } bad++; }
  }
  // sponge-style chaining without reduction between permutations: words 8..15 carried as unreduced doubles
  for (int it = 0; it < 20000; it++) {
    uint32_t a[16]; double d[16];
    for (int i = 0; i < 16; i++) { a[i] = it & 1 ? ext[sm64(seed) % (sizeof ext / 4)] : (uint32_t)(sm64(seed) % kb::P); d[i] = p2f::load_monty(a[i]); }
    for (int round = 0; round < 12; round++) {
      p2::permute_host(a);
      p2f::permute_host(d);
      for (int i = 0; i < 16; i++) if (p2f::store_monty(d[i]) != a[i]) { bad++; break; }
      for (int i = 0; i < 8; i++) { a[i] = it & 2 ? kb::P - 1 - (i & 1) : (uint32_t)(sm64(seed) % kb::P); d[i] = p2f::load_monty(a[i]); }
      n++;
    }
  }
  printf("host check: %ld permutations, %ld mismatches\n", n, bad);
  return bad != 0;
}

constexpr int REPS = 64;
__global__ __launch_bounds__(256) void bench_int(uint32_t* out, uint32_t seed) {
  uint32_t s[16];
  for (int i = 0; i < 16; i++) s[i] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 7919u + seed) % kb::P;
  for (int r = 0; r < REPS; r++) p2::permute(s);
  uint32_t acc = 0;
  for (int i = 0; i < 16; i++) acc ^= s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void bench_f64(uint32_t* out, uint32_t seed) {
  double s[16];
  for (int i = 0; i < 16; i++) s[i] = p2f::load_monty((threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 7919u + seed) % kb::P);
  for (int r = 0; r < REPS; r++) {
    p2f::permute(s);
    // what a compression costs on top: 16 words in, 8 words out
    uint32_t w[8];
    for (int i = 0; i < 8; i++) w[i] = p2f::store_monty(s[i]);
    for (int i = 0; i < 8; i++) s[i] = p2f::load_monty(w[i]);
    for (int i = 8; i < 16; i++) s[i] = p2f::load_monty(w[i - 8] ^ 1u);
  }
  uint32_t acc = 0;
  for (int i = 0; i < 16; i++) acc ^= p2f::store_monty(s[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void bench_f64_pure(uint32_t* out, uint32_t seed) {
  double s[16];
  for (int i = 0; i < 16; i++) s[i] = p2f::load_monty((threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 7919u + seed) % kb::P);
  for (int r = 0; r < REPS; r++) p2f::permute(s);
  uint32_t acc = 0;
  for (int i = 0; i < 16; i++) acc ^= p2f::store_monty(s[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// parity on the device: the same states through both
__global__ void parity(const uint32_t* in, uint32_t* o_int, uint32_t* o_f64, int n) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t a[16]; double d[16];
  for (int i = 0; i < 16; i++) { a[i] = in[16 * t + i]; d[i] = p2f::load_monty(a[i]); }
  p2::permute(a);
  p2f::permute(d);
  for (int i = 0; i < 16; i++) { o_int[16 * t + i] = a[i]; o_f64[16 * t + i] = p2f::store_monty(d[i]); }
}

template <class K>
static int timeit(const char* name, K kernel, uint32_t* d) {
  const int blocks = 256 * 16, threads = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kernel<<<blocks, threads>>>(d, 1);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) kernel<<<blocks, threads>>>(d, 2 + r);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double perms = 5.0 * blocks * threads * REPS;
  printf("%-28s %8.3f ms  %6.2f G permutations/s\n", name, ms, perms / (ms * 1e-3) / 1e9);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "host")) return host_check();
  CHECK(p2::upload_tables());
  CHECK(p2f::upload_tables());
  const int n = 1 << 16;
  std::vector<uint32_t> h(16 * n);
  uint64_t seed = 7;
  for (auto& x : h) x = (uint32_t)(sm64(seed) % kb::P);
  for (int i = 0; i < 64; i++) h[i] = i & 1 ? kb::P - 1 : 0;
  uint32_t *din, *da, *db, *dout;
  CHECK(hipMalloc(&din, 64 * n)); CHECK(hipMalloc(&da, 64 * n)); CHECK(hipMalloc(&db, 64 * n));
  CHECK(hipMalloc(&dout, 256 * 16 * 256 * 4));
  CHECK(hipMemcpy(din, h.data(), 64 * n, hipMemcpyHostToDevice));
  parity<<<n / 256, 256>>>(din, da, db, n);
  std::vector<uint32_t> a(16 * n), b(16 * n);
  CHECK(hipMemcpy(a.data(), da, 64 * n, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(b.data(), db, 64 * n, hipMemcpyDeviceToHost));
  long bad = 0;
  for (int i = 0; i < n; i++) {
    uint32_t ref[16];
    memcpy(ref, &h[16 * i], 64);
    p2::permute_host(ref);
    if (memcmp(ref, &a[16 * i], 64) || memcmp(ref, &b[16 * i], 64)) bad++;
  }
  printf("device parity (int, f64 vs host int): %d states, %ld mismatches\n", n, bad);
  timeit("integer pipe", bench_int, dout);
  timeit("fp64 pipe (permutation only)", bench_f64_pure, dout);
  timeit("fp64 pipe + 16 in / 8 out", bench_f64, dout);
  return bad != 0;
}
