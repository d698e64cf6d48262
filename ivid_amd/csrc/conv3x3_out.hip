// Output convolution of the UNet, fused:  GroupNorm-apply -> SiLU -> Conv2d 3x3 with a handful of output channels,
// written as fp32 NCHW (adm.py:483-487 `self.out` = GroupNorm32, SiLU, zero_module(conv 3x3, model_channels -> out_channels);
// adm.py:565-566 casts back to the input dtype = fp32).
//
// With Cout = 4 the layer has 0.05 % of the model's FLOPs but read its 1 GiB input three times through the generic path
// (gn_apply: read + write; conv_igemm: nine L2 re-reads of the activated copy).  Here the input is read once:
//   tile      : 8 x 32 output pixels of one image; 8 waves, wave w = image row w (one 32-pixel MFMA fragment)
//   A operand : the halo image of conv3x3_fused.hip (one 144-byte LDS row per halo pixel, transformed y = silu(x*a + b)
//               while it is staged, coefficients parked in the row pads)
//   B operand : ALL 9 taps of a 128-byte channel chunk at once: [tap][Cout][128 B] = 4.5 KB for Cout = 4; the 32-lane
//               fragment reads row min(lane, Cout-1), so the padded accumulator columns hold copies that are never stored
//   schedule  : two barriers per CHUNK (not per tap): 36 MFMAs per wave between them; the 6 halo pieces of the next chunk
//               are requested before the MFMAs and transformed in lockstep after them (16 accumulator VGPRs leave room
//               to keep all six in flight)
//   occupancy : ONE halo image (the transform of chunk c+1 overwrites it after the MFMAs of chunk c) + weight stages sized
//               for the launch's Cout keep the workgroup at ~58 KB of LDS and <= 128 VGPRs, so TWO workgroups share a CU:
//               one's transform (VALU / transcendental pipes) runs under the other's MFMAs and memory waits.  (The
//               double-buffered single-workgroup form measured 1.7 TB/s: nothing overlapped its exp/rcp chains.)
// The kernel is bound by the halo transform and the input stream, not by the matrix pipe.
#include "common.h"
#include "internal.h"

namespace {

struct OutArgs {
  const char* src_lo;  // SPLIT: optional lo plane of src (compensated 16-bit storage, conv_igemm.hip ConvArgs)
  const char* w_lo;    // SPLIT: lo parts of the weights, [Cout][9][C]
  const char* src;
  const float* ab;     // [N][C][2]
  const char* w;       // [Cout][9][C]
  const float* bias;   // [Cout] or null
  float* out;          // fp32 NCHW [N][Cout][H][W]
  int C, N, H, W, Cout;
  int tiles_x, tiles_y, ntiles_total;
};

constexpr int TH = 8, TW = 32, HW_ = TW + 2, HH_ = TH + 2, HROWS = HH_ * HW_;
constexpr int NT = 512, PIECES = (HROWS + 63) / 64;
// halo row: 128 B of channels + a 16-byte pad (odd number of 16-byte slots: conflict-free for any row shift); SPLIT: 128 B
// of hi parts, 128 B of lo parts, pad (17 slots)
__host__ __device__ constexpr int out_arow(bool split) { return split ? 272 : 144; }
constexpr int MAXCO = 16;                              // most output channels a launch may have
constexpr int MAXCO_SPLIT = 8;                         // ... in the split form (two weight stages per chunk, 272-byte halo rows)
constexpr int COEF_BYTES = 32 * 16;                    // GroupNorm coefficients of one chunk: (a0,a1,b0,b1) per channel pair
// One weight stage: 9 * Cout * 128 B, rounded up to whole waves of LDS-DMA (64 lanes x 16 B: the idle lanes of the last
// wave write too)
__host__ __device__ static inline int out_stage_bytes(int Cout) { return (9 * Cout * 8 + 63) / 64 * 1024; }
// LDS of a launch: [halo image][2 coefficient sets][2 weight stages]
static inline int out_lds_bytes(int Cout, bool split = false) {
  return HROWS * out_arow(split) + 2 * COEF_BYTES + 2 * (split ? 2 : 1) * out_stage_bytes(Cout);
}

// SPLIT (16-bit types, precision mode fp16c): the layer is computed to ~2^-21 instead of 2^-11 -- activated values and
// weights are both carried as hi + lo parts and every product is three MFMAs (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi).  The
// head sits right in front of the model output, so its operand roundings are not averaged by anything behind it (they
// were 7 % of the fp16 mode's error budget, tests/tools/error_budget.py) and it has 0.05 % of the FLOPs.
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(SPLIT ? 2 : 4, SPLIT ? 2 : 4))) void conv3x3_out_kernel(const OutArgs p) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  constexpr int BKE = 128 / (int)sizeof(T);
  constexpr int AROW = out_arow(SPLIT), A_BYTES = HROWS * AROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA0 = smem;
  char* const sCf = smem + A_BYTES;
  char* const sB0 = sCf + 2 * COEF_BYTES;
  const int B_HALF = out_stage_bytes(p.Cout);          // one chunk, all taps (SPLIT: the hi parts; the lo parts follow)
  const int B_BYTES = (SPLIT ? 2 : 1) * B_HALF;

  const int tile = xcd_remap(blockIdx.x, p.ntiles_total);
  const int tx = tile % p.tiles_x;
  int rest = tile / p.tiles_x;
  const int ty = rest % p.tiles_y;
  const int img = rest / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int C = p.C, Cout = p.Cout;
  const int chunks = C / BKE;

  // ---- halo pieces of this thread (as conv3x3_fused.hip) ----
  const int cpc = tid & 7, hrow0 = tid >> 3;
  int pix[PIECES];
  unsigned okbits = 0;
#pragma unroll
  for (int j = 0; j < PIECES; ++j) {
    const int hrow = j * 64 + hrow0;
    const int hy = hrow / HW_, hx = hrow - hy * HW_;
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool ok = hrow < HROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    pix[j] = ok ? y * p.W + x : 0;
    okbits |= (ok ? 1u : 0u) << j;
  }
  const int st_lds = hrow0 * AROW + cpc * 16;
  const bool act5 = hrow0 < HROWS - 5 * 64;
  const char* const src_img = p.src + (size_t)img * p.H * p.W * C * sizeof(T);
  const int cb = C * (int)sizeof(T);
  const float* abn = p.ab + (size_t)img * C * 2;

  const bool has_lo = SPLIT && p.src_lo != nullptr;
  const char* const lo_img = has_lo ? p.src_lo + (size_t)img * p.H * p.W * C * sizeof(T) : nullptr;
  auto load_pieces = [&](int ch, vec_t* raw, vec_t* rawl) {
    const char* base = src_img + (size_t)ch * 128;
#pragma unroll
    for (int j = 0; j < PIECES; ++j) raw[j] = *(const vec_t*)(base + (size_t)(__umul24(pix[j], cb) + cpc * 16));
    if constexpr (SPLIT) {
      if (has_lo) {
        const char* bl = lo_img + (size_t)ch * 128;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) rawl[j] = *(const vec_t*)(bl + (size_t)(__umul24(pix[j], cb) + cpc * 16));
      }
    }
  };
  auto ab_load = [&](int ch) -> f32x4 {
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && lane < BKE / 2) q = *(const f32x4*)(abn + (size_t)ch * BKE * 2 + lane * 4);
    return q;
  };
  auto ab_store = [&](const f32x4& q, int set) {
    if (wave == 0 && lane < BKE / 2) *(f32x4*)(sCf + set * COEF_BYTES + lane * 16) = f32x4{q[0], q[2], q[1], q[3]};
  };
  // three pieces at a time in lockstep: shared coefficients (same channel piece), 6 overlapping exp / rcp chains -- with
  // four waves per SIMD resident that is enough to hide the transcendental latency, and it keeps the kernel at 128 VGPRs
  auto xform_all = [&](const vec_t* raw, const vec_t* rawl, char* sAdst, int set) {
    const char* cf = sCf + set * COEF_BYTES + cpc * (VE / 2) * 16;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int GP = VE == 8 ? 2 : 3;
#pragma unroll
    for (int j0 = 0; j0 < PIECES; j0 += GP) {
      float f[GP][VE];
#pragma unroll
      for (int j = 0; j < GP; ++j) vec_to_f32<T>(raw[j0 + j], f[j]);
      if constexpr (SPLIT) {
        if (has_lo) {
#pragma unroll
          for (int j = 0; j < GP; ++j) {
            float l[VE];
            vec_to_f32<T>(rawl[j0 + j], l);
#pragma unroll
            for (int e = 0; e < VE; ++e) f[j][e] += l[e];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < VE / 2; ++k) {
        const f32x4 qk = *(const f32x4*)(cf + k * 16);   // re-read per group: 4 VGPRs live instead of 16
        f32x2 v[GP], d[GP];
#pragma unroll
        for (int j = 0; j < GP; ++j) {
          v[j] = f32x2{f[j][2 * k], f[j][2 * k + 1]} * f32x2{qk[0], qk[1]} + f32x2{qk[2], qk[3]};
          const f32x2 t = v[j] * -1.4426950408889634f;
          d[j] = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        }
#pragma unroll
        for (int j = 0; j < GP; ++j) {
          d[j] = d[j] + 1.0f;
          const f32x2 y = v[j] * f32x2{__builtin_amdgcn_rcpf(d[j][0]), __builtin_amdgcn_rcpf(d[j][1])};
          f[j][2 * k] = y[0];
          f[j][2 * k + 1] = y[1];
        }
      }
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int jj = j0 + j;
        const vec_t hv = f32_to_vec<T>(f[j]);
        u32x4 ob = __builtin_bit_cast(u32x4, hv);
        const unsigned keep = (okbits >> jj) & 1 ? 0xffffffffu : 0u;
        ob &= keep;
        if (jj < PIECES - 1 || act5) *(u32x4*)(sAdst + st_lds + jj * 64 * AROW) = ob;
        if constexpr (SPLIT) {
          float h[VE];
          vec_to_f32<T>(hv, h);
#pragma unroll
          for (int e = 0; e < VE; ++e) h[e] = f[j][e] - h[e];
          u32x4 lb = __builtin_bit_cast(u32x4, f32_to_vec<T>(h));
          lb &= keep;
          if (jj < PIECES - 1 || act5) *(u32x4*)(sAdst + st_lds + jj * 64 * AROW + 128) = lb;
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // groups stay apart: the scheduler would otherwise unpack all six pieces first
    }
  };
  // weights of chunk ch, all taps: piece q = (tap * Cout + row) * 8 + pc  ->  LDS byte 16 q (lane-linear per wave)
  const int npieces_b = 9 * Cout * 8;
  auto issue_b = [&](int stage, int ch) {
    for (int q0 = 0; q0 < npieces_b; q0 += NT) {   // wave-uniform trip count (<= 3)
      const int q = q0 + tid;
      if (q0 + wave * 64 < npieces_b) {             // whole waves past the end skip the DMA
        const int qq = min(q, npieces_b - 1);       // idle lanes of the last wave re-read the last piece
        const int pc = qq & 7, rt = qq >> 3;
        const int tap = rt / Cout, row = rt - tap * Cout;
        const unsigned voff = (unsigned)((((size_t)row * 9 + tap) * C + (size_t)ch * BKE) * sizeof(T)) + pc * 16;
        glds16_s(p.w, voff, sB0 + stage * B_BYTES + (q0 + wave * 64) * 16);
        if constexpr (SPLIT) glds16_s(p.w_lo, voff, sB0 + stage * B_BYTES + B_HALF + (q0 + wave * 64) * 16);
      }
    }
  };

  const int frow = lane & 31, fhalf = lane >> 5;
  const int a_base = (wave * HW_ + frow) * AROW + fhalf * 16;          // top-left tap, k-piece 0
  const int b_base = min(frow, Cout - 1) * 128 + fhalf * 16;

  // ---------------- prologue ----------------
  vec_t raw[PIECES];
  vec_t rawl[SPLIT ? PIECES : 1];
  {
    const f32x4 q0 = ab_load(0);
    issue_b(0, 0);
    load_pieces(0, raw, rawl);
    ab_store(q0, 0);
    wait_vmcnt0();
    __syncthreads();
    xform_all(raw, rawl, sA0, 0);
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int ch = 0; ch < chunks; ++ch) {
    const bool more = ch + 1 < chunks;
    wait_vmcnt0();
    __syncthreads();  // halo(ch) complete, weights(ch) landed, everyone done with the other weight stage
    const char* aptr = sA0 + a_base;
    const char* bptr = sB0 + (ch & 1) * B_BYTES + b_base;
    f32x4 abq = {0.f, 0.f, 0.f, 0.f};
    if (more) {
      issue_b((ch + 1) & 1, ch + 1);
      abq = ab_load(ch + 1);
      load_pieces(ch + 1, raw, rawl);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const vec_t a = *(const vec_t*)(aptr + ((tap / 3) * HW_ + tap % 3) * AROW + kk * 32);
        const vec_t b = *(const vec_t*)(bptr + tap * Cout * 128 + kk * 32);
        if constexpr (SPLIT) {   // small terms first
          const vec_t al = *(const vec_t*)(aptr + ((tap / 3) * HW_ + tap % 3) * AROW + 128 + kk * 32);
          const vec_t bl = *(const vec_t*)(bptr + B_HALF + tap * Cout * 128 + kk * 32);
          MmaT<T>::run(al, b, acc);
          MmaT<T>::run(a, bl, acc);
        }
        MmaT<T>::run(a, b, acc);
      }
    }
    if (more) {
      ab_store(abq, (ch + 1) & 1);
      wait_vmcnt0();
      __syncthreads();  // every wave is done reading the image of chunk ch; coefficients of chunk ch+1 visible
      xform_all(raw, rawl, sA0, (ch + 1) & 1);
    }
  }

  // ---------------- epilogue: accumulator column = output channel, 16 registers = 16 of the row's 32 pixels ----------------
  const int co = frow;
  if (co < Cout) {
    const float bs = p.bias ? p.bias[co] : 0.f;
    float* o = p.out + (((size_t)img * Cout + co) * p.H + (y0 + wave)) * p.W + x0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // registers 4j..4j+3 = pixels 8j + 4*fhalf + 0..3
      const f32x4 v = {acc[4 * j] + bs, acc[4 * j + 1] + bs, acc[4 * j + 2] + bs, acc[4 * j + 3] + bs};
      *(f32x4*)(o + 8 * j + 4 * fhalf) = v;
    }
  }
}

template <typename T, bool SPLIT = false> int launch_out(const OutArgs& a, hipStream_t stream) {
  auto kern = conv3x3_out_kernel<T, SPLIT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       out_lds_bytes(SPLIT ? MAXCO_SPLIT : MAXCO, SPLIT));
    if (e != hipSuccess) return ivid_set_error("conv3x3_gn_out: hipFuncSetAttribute", e);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.ntiles_total), dim3(NT), out_lds_bytes(a.Cout, SPLIT), stream, a);
  return ivid_check_launch("conv3x3_gn_out");
}

}  // namespace

extern "C" int ivid_conv3x3_gn_out(int dtype, const void* src, int C, const float* ab, const void* weight, const float* bias,
                                   float* out, int N, int H, int W, int Cout, void* stream) {
  return ivid_conv3x3_gn_out_c(dtype, src, nullptr, C, ab, weight, nullptr, bias, out, N, H, W, Cout, stream);
}

// weight_lo != NULL (16-bit dtypes): the split form -- weights as hi + lo parts, activated values split in the kernel, three
// MFMAs per product; src_lo: optional lo plane of the input (compensated 16-bit storage, precision mode fp16c)
extern "C" int ivid_conv3x3_gn_out_c(int dtype, const void* src, const void* src_lo, int C, const float* ab, const void* weight,
                                     const void* weight_lo, const float* bias, float* out, int N, int H, int W, int Cout,
                                     void* stream) {
  const int esz = ivid_esz(dtype);
  if (!esz) return ivid_set_error("conv3x3_gn_out: bad dtype", hipSuccess);
  const int bke = 128 / esz;
  if (C <= 0 || C % bke) return ivid_set_error("conv3x3_gn_out: channels must be a multiple of the K-step", hipSuccess);
  if (Cout <= 0 || Cout > MAXCO) return ivid_set_error("conv3x3_gn_out: 1..16 output channels (use ivid_conv3x3_gn otherwise)", hipSuccess);
  if (W % TW || H % TH) return ivid_set_error("conv3x3_gn_out: needs W % 32 == 0 and H % 8 == 0", hipSuccess);
  if (!src || !ab || !weight || !out) return ivid_set_error("conv3x3_gn_out: null argument", hipSuccess);
  if ((size_t)H * W * C * esz >= ((size_t)1 << 31) || (size_t)Cout * 9 * C * esz >= ((size_t)1 << 32))
    return ivid_set_error("conv3x3_gn_out: image or weight matrix too large for 32-bit offsets", hipSuccess);
  if ((src_lo || weight_lo) && esz != 2) return ivid_set_error("conv3x3_gn_out: the split form needs a 16-bit dtype", hipSuccess);
  if (src_lo && !weight_lo) return ivid_set_error("conv3x3_gn_out: src_lo needs the split form (weight_lo)", hipSuccess);
  if (weight_lo && Cout > MAXCO_SPLIT) return ivid_set_error("conv3x3_gn_out: the split form takes 1..8 output channels", hipSuccess);
  OutArgs a;
  a.src_lo = (const char*)src_lo; a.w_lo = (const char*)weight_lo;
  a.src = (const char*)src; a.ab = ab; a.w = (const char*)weight; a.bias = bias; a.out = out;
  a.C = C; a.N = N; a.H = H; a.W = W; a.Cout = Cout;
  a.tiles_x = W / TW; a.tiles_y = H / TH; a.ntiles_total = N * a.tiles_x * a.tiles_y;
  if (weight_lo) {
    if (dtype == IVID_BF16) return launch_out<__bf16, true>(a, (hipStream_t)stream);
    return launch_out<_Float16, true>(a, (hipStream_t)stream);
  }
  if (dtype == IVID_BF16) return launch_out<__bf16>(a, (hipStream_t)stream);
  if (dtype == IVID_F16) return launch_out<_Float16>(a, (hipStream_t)stream);
  // IVID_BF16X3: fp32 storage and PLAIN fp32 weights here (0.05 % of the FLOPs; the kernel is bound by its input stream)
  return launch_out<float>(a, (hipStream_t)stream);
}
