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
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Storage type of the split-bf16 ("bf16x3") precision mode: activations are plain fp32 in HBM; only the MFMA kernels
// treat it differently (operands split into bf16 hi + lo, three bf16 MFMAs per product, fp32 accumulate).
struct bf16x3_t { float v; };

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
template <> struct Elem<_Float16> {
  static constexpr int VE = 8;
  typedef f16x8 vec;
};
template <> struct Elem<bf16x3_t> {
  static constexpr int VE = 4;
  typedef f32x4 vec;
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
template <> __device__ __forceinline__ void vec_to_f32<_Float16>(const f16x8& v, float* f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <> __device__ __forceinline__ void vec_to_f32<bf16x3_t>(const f32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = v[i];
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

template <> __device__ __forceinline__ f16x8 f32_to_vec<_Float16>(const float* f) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (_Float16)f[i];  // RNE (v_cvt_pk_f16_f32); overflows to inf like the reference's .half()
  return v;
}
template <> __device__ __forceinline__ f32x4 f32_to_vec<bf16x3_t>(const float* f) {
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = f[i];
  return v;
}

// Split of 8 fp32 values into bf16 hi + bf16 lo (x = hi + lo up to 2^-17 |x|): the operand form of the bf16x3 mode.
__device__ __forceinline__ void split_bf16x8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 h0 = (__bf16)x0[i], h1 = (__bf16)x1[i];
    hi[i] = h0;
    hi[4 + i] = h1;
    lo[i] = (__bf16)(x0[i] - (float)h0);
    lo[4 + i] = (__bf16)(x1[i] - (float)h1);
  }
}

// One MFMA "product" in each precision mode.  A and B use the same k permutation, so the sum is unchanged.
template <typename T> struct MmaT;
template <> struct MmaT<__bf16> {
  static __device__ __forceinline__ void run(const bf16x8& a, const bf16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaT<_Float16> {
  static __device__ __forceinline__ void run(const f16x8& a, const f16x8& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MmaT<float> {
  // one 16-byte piece per lane = 4 k-values; MFMA j pairs k=j of the low-lane piece with k=j of the high-lane piece
  static __device__ __forceinline__ void run(const f32x4& a, const f32x4& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
  }
};
template <typename T> struct IsSplit { static constexpr bool value = false; };
template <> struct IsSplit<bf16x3_t> { static constexpr bool value = true; };

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division sequence: these kernels are
// HBM-bound only as long as the VALU work per 16-byte piece stays small
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Async 16-byte global -> LDS copy (global_load_lds_dwordx4): LDS destination = wave-uniform base + lane*16, the
// global source address is per lane.
// Issued through inline assembly ON PURPOSE: hipcc models the builtin as a FLAT operation that touches both memory
// and LDS and then forces every later `s_waitcnt lgkmcnt` to 0 until the next vmcnt(0) — which serialises the
// ds_read -> MFMA software pipeline of the conv kernels.  With the asm form the compiler counts its own LDS reads
// exactly; the DMA itself is ordered by hand (wait_vmcnt0() + barrier before the stage is read, as before).  No other
// code in these kernels uses M0.
// Wave-uniform LDS byte address (SGPR) of a pointer into the dynamic LDS block.  Computed as an offset from the dynamic
// LDS symbol -- a generic->LDS address-space cast would carry a null check (and trips a compiler bug in some
// instantiations).  Every kernel here uses `extern __shared__` memory only.
extern __shared__ __attribute__((aligned(16))) char ivid_dyn_lds[];
__device__ __forceinline__ unsigned lds_addr_of(const void* lds_ptr) {
  const unsigned off = (unsigned)((const char*)lds_ptr - (const char*)ivid_dyn_lds) + __builtin_amdgcn_groupstaticsize();
  return __builtin_amdgcn_readfirstlane(off);
}
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr_of(lds_wave_base)) : "memory");
}
// same with a wave-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: no 64-bit VALU address arithmetic
__device__ __forceinline__ void glds16_s(const void* base, unsigned voff, void* lds_wave_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr_of(lds_wave_base)) : "memory");
}

// Ordering between LDS writes and reads of ONE wave (per-wave transpose slabs): the LDS executes a wave's instructions
// in order, so no workgroup barrier is needed -- only the compiler must not reorder across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Bijective XCD-aware remap: hardware places block b on XCD b%8; give every XCD a
// contiguous range of logical tile ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

