// Shared device-side helpers for the ivid_amd HIP kernels (gfx950 / CDNA4 only).
//
// Data layout used by every UNet kernel: activations are NHWC ("pixel-major"):
//   x[n][y][x][c], element type T = float (parity mode) or __bf16 (perf mode).
// All global<->LDS traffic moves in 16-byte pieces (VE elements of T).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VE = 4;  // elements per 16-byte piece
  typedef f32x4 vec;
};
template <> struct Elem<__bf16> {
  static constexpr int VE = 8;
  typedef bf16x8 vec;
};

// 16-byte piece <-> fp32 lanes.
template <typename T> __device__ __forceinline__ void vec_to_f32(const typename Elem<T>::vec& v, float* f);
template <> __device__ __forceinline__ void vec_to_f32<float>(const f32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = v[i];
}
template <> __device__ __forceinline__ void vec_to_f32<__bf16>(const bf16x8& v, float* f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ typename Elem<T>::vec f32_to_vec(const float* f);
template <> __device__ __forceinline__ f32x4 f32_to_vec<float>(const float* f) {
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = f[i];
  return v;
}
template <> __device__ __forceinline__ bf16x8 f32_to_vec<__bf16>(const float* f) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];  // RNE (v_cvt_pk_bf16_f32)
  return v;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence: these kernels are
// HBM-bound only as long as the VALU work per 16-byte piece stays small
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Async 16-byte global -> LDS copy (global_load_lds_dwordx4). LDS destination is
// wave-uniform base + lane*16; the global source address is per lane.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Bijective XCD-aware remap: hardware places block b on XCD b%8; give every XCD a
// contiguous range of logical tile ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

