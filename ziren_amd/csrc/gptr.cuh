// A pointer that was read from memory (a column-pointer array, a batch descriptor) is a generic pointer to the compiler, and a load
// through it is a FLAT instruction: it counts on lgkmcnt as well as vmcnt, so every wait for an LDS read or for a scalar load (the
// Poseidon2 round constants) also waits for it — a prefetch issued "one permutation ahead" stalls the first round-constant wait of
// that permutation instead of running under it. Everything these pointers name is device global memory: the accessors below say so,
// and the accesses become global_load / global_store (vmcnt only).
#pragma once

namespace gp {
#if defined(__HIP_DEVICE_COMPILE__)
#define GP_AS1 __attribute__((address_space(1)))
#else
#define GP_AS1
#endif
template <class T>
__device__ __forceinline__ T load(const T* p) { return *(const T GP_AS1*)p; }
template <class T>
__device__ __forceinline__ void store(T* p, const T& v) { *(T GP_AS1*)p = v; }
// 16-byte accesses: HIP's uint4 is a class type; the access goes through a plain vector type of the same layout
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load(const uint4* p) {
  const u32x4 v = *(const u32x4 GP_AS1*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store(uint4* p, const uint4& v) {
  u32x4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  *(u32x4 GP_AS1*)p = w;
}
// a quartic-extension element (four words, 16-byte aligned) through one 16-byte access; E is kb::E4 (kept a template so that this
// header does not depend on kb31.cuh)
template <class E>
__device__ __forceinline__ E load_e4(const E* p) {
  const u32x4 v = *(const u32x4 GP_AS1*)p;
  E e; e.c[0] = v.x; e.c[1] = v.y; e.c[2] = v.z; e.c[3] = v.w;
  return e;
}
template <class E>
__device__ __forceinline__ void store_e4(E* p, const E& e) {
  u32x4 w; w.x = e.c[0]; w.y = e.c[1]; w.z = e.c[2]; w.w = e.c[3];
  *(u32x4 GP_AS1*)p = w;
}
}  // namespace gp
