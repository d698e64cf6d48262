// Microbenchmark (round 5): what the MATRIX CORE LOOP of a Winograd F(2x2, 3x3) convolution kernel can sustain
// on an MI355X -- the 16 position GEMMs  M[p] = V[p] (tiles x Cin) . U[p]^T (Cin x Cout)  fed from LDS, with every operand of a
// K-stage arriving by LDS-DMA and being read exactly once per wave (the 16 positions are independent GEMMs: a fragment is reused
// across output-channel fragments of ONE position only).  No input / output transform, no GroupNorm: an UPPER bound for a real kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -I ivid_amd/csrc -o scripts/micro/wino_core scripts/micro/wino_core.hip && scripts/micro/wino_core
//
// Workload = the decoder's 128^2 512 -> 256 layer at the benchmark's stacked batch (128 images): 128 x 64 x 64 Winograd tiles.
// Per workgroup (512 threads, 8 waves): MT tiles x NC output channels x 16 positions; wave w owns positions 2w, 2w+1 for the whole
// (MT x NC) block: 2 x (MT/32) x (NC/32) = 8 accumulator fragments = 128 VGPRs, the budget of the product's direct kernel.
// K-stage = 16 input channels (one MFMA k-block): A stage [16][MT][32 B], B stage [16][NC][32 B], double-buffered.
// Reported: executed MFMA TFLOP/s and the DIRECT-EQUIVALENT rate (x 36/16: the multiplications a direct 3x3 convolution needs for the
// same outputs), to be read against the product's direct kernel on this layer (1050 TF/s, profiles/r04_layers_fp16s.json) and the
// power-limited MFMA ceiling on random operands (1606 TF/s, profiles/r01_mfma_power.txt).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"

template <int MT, int NC>
__global__ __launch_bounds__(512) void wino_core(const char* __restrict__ Vg, const char* __restrict__ Ug, float* __restrict__ out,
                                                 int stages) {
  constexpr int MI = MT / 32, NI = NC / 32;
  static_assert(MI * NI * 2 == 8, "8 accumulator fragments per wave");
  constexpr int A_BYTES = 16 * MT * 32, B_BYTES = 16 * NC * 32, ST = A_BYTES + B_BYTES;
  static_assert(2 * ST <= 160 * 1024, "LDS");
  constexpr int APC = A_BYTES / (512 * 16), BPC = B_BYTES / (512 * 16);   // 16-byte pieces per thread and stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's A slabs: [stage][16][MT][32 B] contiguous per (workgroup, stage); B slabs are shared by all workgroups of a
  // cout tile
  // (the cout tiles of one tile block share its A slabs; the A stream wraps over 128 distinct tile blocks = 64-128 MB, i.e. it comes
  // from the memory-side cache: a real kernel loads the UNtransformed halo -- a quarter of these bytes -- from HBM and writes V itself)
  const size_t a_wg = (size_t)((blockIdx.x / (256 / NC)) % 128) * stages * A_BYTES;
  const size_t b_wg = (size_t)(blockIdx.x % (256 / NC)) * stages * B_BYTES;
  // LDS rows are 32 bytes; the two 16-byte pieces of row r are stored at slot p ^ ((r >> 3) & 1): the 16 lanes of a ds_read_b128
  // group (rows r .. r+15, one piece each) then hit 16 distinct 16-byte slots of the 256-byte bank window
  unsigned a_src[APC], b_src[BPC];
#pragma unroll
  for (int i = 0; i < APC; ++i) {
    const int idx = i * 512 + tid, row = idx >> 1, slot = idx & 1;
    a_src[i] = row * 32 + ((slot ^ ((row >> 3) & 1)) << 4);
  }
#pragma unroll
  for (int i = 0; i < BPC; ++i) {
    const int idx = i * 512 + tid, row = idx >> 1, slot = idx & 1;
    b_src[i] = row * 32 + ((slot ^ ((row >> 3) & 1)) << 4);
  }
  auto issue = [&](int st) {
    char* sA = smem + (st & 1) * ST;
    char* sB = sA + A_BYTES;
    const char* ga = Vg + a_wg + (size_t)st * A_BYTES;
    const char* gb = Ug + b_wg + (size_t)st * B_BYTES;
#pragma unroll
    for (int i = 0; i < APC; ++i) glds16_s(ga, a_src[i], sA + (i * 512 + wave * 64) * 16);
#pragma unroll
    for (int i = 0; i < BPC; ++i) glds16_s(gb, b_src[i], sB + (i * 512 + wave * 64) * 16);
  };
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_off[2][MI], b_off[2][NI];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int p = 2 * wave + pp;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row = p * MT + mi * 32 + frow;
      a_off[pp][mi] = row * 32 + ((fhalf ^ ((row >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = p * NC + ni * 32 + frow;
      b_off[pp][ni] = A_BYTES + row * 32 + ((fhalf ^ ((row >> 3) & 1)) << 4);
    }
  }
  f32x16 acc[2][MI][NI];
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pp][mi][ni][r] = 0.f;
  issue(0);
  for (int st = 0; st < stages; ++st) {
    wait_vmcnt0();
    __syncthreads();   // stage st landed for every wave; everyone finished reading stage st-1
    if (st + 1 < stages) issue(st + 1);
    const char* s = smem + (st & 1) * ST;
    f16x8 a[2][MI], b[2][NI];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[pp][mi] = *(const f16x8*)(s + a_off[pp][mi]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[pp][ni] = *(const f16x8*)(s + b_off[pp][ni]);
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[pp][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[pp][mi], b[pp][ni], acc[pp][mi][ni], 0, 0, 0);
  }
  float sum = 0.f;
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[pp][mi][ni][r];
  out[(size_t)blockIdx.x * 512 + tid] = sum;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int MT, int NC> int run(const char* name, int images, int cin) {
  const int stages = cin / 16;
  const long tiles = (long)images * 64 * 64;
  const int nwg = (int)(tiles / MT) * (256 / NC);
  constexpr int A_BYTES = 16 * MT * 32, B_BYTES = 16 * NC * 32;
  const size_t a_bytes = (size_t)128 * stages * A_BYTES;
  const size_t b_bytes = (size_t)(256 / NC) * stages * B_BYTES;
  char *V = nullptr, *U = nullptr; float* out = nullptr;
  CK(hipMalloc(&V, a_bytes)); CK(hipMalloc(&U, b_bytes)); CK(hipMalloc(&out, (size_t)nwg * 512 * 4));
  {   // random fp16 operands (N(0,1)-like magnitudes): the chip's clock under MFMA load depends on the operands' toggling
    std::vector<_Float16> h((size_t)16 << 20);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 3.4f);
    for (size_t o = 0; o < a_bytes; o += h.size() * 2) CK(hipMemcpy(V + o, h.data(), std::min(h.size() * 2, a_bytes - o), hipMemcpyHostToDevice));
    for (size_t o = 0; o < b_bytes; o += h.size() * 2) CK(hipMemcpy(U + o, h.data(), std::min(h.size() * 2, b_bytes - o), hipMemcpyHostToDevice));
  }
  auto kern = wino_core<MT, NC>;
  constexpr int LDS = 2 * (A_BYTES + B_BYTES);
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), LDS, 0, V, U, out, stages);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), LDS, 0, V, U, out, stages);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double flop = 2.0 * tiles * 16 * 256 * cin;                 // executed: 16 position GEMMs
  const double direct = 2.0 * tiles * 4 * 9 * 256 * cin;            // a direct 3x3 convolution of the same 2x2-pixel tiles
  printf("%-26s images %d Cin %d: %6d workgroups, LDS %6d B/wg, %.3f ms  executed %.0f TF/s  direct-equivalent %.0f TF/s  "
         "(LDS traffic %.1f B/clk/CU at 2.4 GHz nominal)\n", name, images, cin, nwg, LDS, ms, flop / ms / 1e9, direct / ms / 1e9,
         2.0 * (double)nwg * stages * (A_BYTES + B_BYTES) / (ms * 1e-3) / 256 / 2.4e9);
  CK(hipFree(V)); CK(hipFree(U)); CK(hipFree(out));
  return 0;
}

int main() {
  int rc = 0;
  rc |= run<32, 128>("32 tiles x 128 couts", 128, 512);
  rc |= run<64, 64>("64 tiles x 64 couts", 128, 512);
  rc |= run<32, 128>("32 tiles x 128 couts", 128, 256);
  return rc;
}
