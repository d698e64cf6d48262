// Microbenchmark: sustained MFMA rate of the two bf16 MFMA shapes on random vs zero operands (power / clock ceiling).
// hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// operand order probe: which (a, b) pair each of the 8 MFMAs of an iteration uses
template <int PAT>
__global__ __launch_bounds__(512) void kp(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[(i * 512 + lane) % 4096];
  for (int i = 0; i < 2; ++i) b[i] = in[((i + 4) * 512 + lane) % 4096];
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ai = PAT == 0 ? (i & 3) : (PAT == 1 ? (i >> 1) : 0);     // 0: a cycles, b fixed per 4 (kernel order)
      const int bi = PAT == 0 ? (i >> 2) : (PAT == 1 ? (i & 1) : 0);     // 1: a fixed per 2, b alternates; 2: all equal
      c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ai], b[bi], c[i], 0, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  out[blockIdx.x * 512 + lane] = s;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[(i * 512 + lane) % 4096];
  for (int i = 0; i < 2; ++i) b[i] = in[((i + 4) * 512 + lane) % 4096];
  if (SHAPE == 0) {
    f32x16 c[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 512 + lane] = s;
  } else {
    f32x4 c[32];
    for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 1], c[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
    out[blockIdx.x * 512 + lane] = s;
  }
}

int main() {
  const int nblk = 256 * 1, iters = 20000;
  std::vector<unsigned short> h(4096 * 8);
  bf16x8* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, nblk * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (auto& v : h) {  // random bf16 in about [-2,2] (sign, exponent 126..128, random mantissa) or zeros
      v = zero ? 0 : (unsigned short)(((rand() & 1) << 15) | ((126 + rand() % 3) << 7) | (rand() & 127));
    }
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(512), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per wave per iteration: shape 0: 8 x 32x32x16 ; shape 1: 32 x 16x16x32  -> both 8*32768 = 262144 MAC*2 flop
        const double flop = (double)nblk * 8 * iters * 8.0 * 32 * 32 * 16 * 2;
        if (rep) printf("%s operands, %s: %.2f ms, %.0f TFLOP/s\n", zero ? "zero  " : "random", shape ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9);
      }
    }
  }
  // operand-order probe on random operands
  srand(1);
  for (auto& v : h) v = (unsigned short)(((rand() & 1) << 15) | ((126 + rand() % 3) << 7) | (rand() & 127));
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int pat = 0; pat < 3; ++pat)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (pat == 0) hipLaunchKernelGGL(kp<0>, dim3(nblk), dim3(512), 0, 0, din, dout, iters);
      else if (pat == 1) hipLaunchKernelGGL(kp<1>, dim3(nblk), dim3(512), 0, 0, din, dout, iters);
      else hipLaunchKernelGGL(kp<2>, dim3(nblk), dim3(512), 0, 0, din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)nblk * 8 * iters * 8.0 * 32 * 32 * 16 * 2;
      if (rep) printf("random operands, order pattern %d: %.2f ms, %.0f TFLOP/s\n", pat, ms, flop / ms / 1e9);
    }
  return 0;
}
