// GroupNorm32 (+SiLU, +FiLM scale/shift, +nearest-up / avg-pool-down, + skip concat) for NHWC tensors.
//
// Reference: GroupNorm32 (adm.py:36-41: nn.GroupNorm(32, C, eps=1e-5, affine) evaluated in fp32),
// SiLU (adm.py:159,180,485), FiLM `out_norm(h)*(1+scale)+shift` (adm.py:214-218), the resampling
// between activation and conv in up/down ResBlocks (adm.py:203-208) and torch.cat([h, hs.pop()])
// (adm.py:563), which is never materialised on its own: the three kernels read two base pointers.
//
// All three are HBM-bound streaming kernels (16-byte pieces, consecutive lanes = consecutive
// channel pieces of one pixel = fully coalesced NHWC rows):
//   gn_partial  : per (n, pixel-chunk) block -> per-channel sum / sum of squares (fp32)
//   gn_finalize : per (n, group) block -> mean/rstd in fp64 from the partials, folded with gamma/beta
//                 and the FiLM scale/shift into one (a,b) pair per (n, channel)
//   gn_apply    : out = act(x*a + b), with resampling, written once as the conv's input tensor
// Group statistics that straddle the concat seam (e.g. 1792 = 1024+768 channels, 56-channel groups)
// need no special casing because partials are per channel.
#include "common.h"
#include "internal.h"

namespace {

constexpr int GN_NT = 256;

__host__ __device__ inline int gn_ppc(int HW) { return HW >= 4096 ? 256 : 64; }

template <typename T, bool LO = false>
__global__ __launch_bounds__(GN_NT) void gn_partial_kernel(const char* __restrict__ src0, int C0,
                                                           const char* __restrict__ src1, int C1, int HW, int ppc,
                                                           int nchunks, float* __restrict__ partial,
                                                           const char* __restrict__ lo0 = nullptr,
                                                           const char* __restrict__ lo1 = nullptr) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  __shared__ float red[GN_NT * VE * 2];
  const int C = C0 + C1, CV = C / VE, CV0 = C0 / VE;
  const int chunk = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int CVs = CV < GN_NT ? CV : GN_NT;
  const int PIF = CV < GN_NT ? GN_NT / CV : 1;  // pixels in flight per block iteration
  const int p0 = t / CVs;
  const int pbeg = chunk * ppc, pend = min(pbeg + ppc, HW);
  float* outp = partial + ((size_t)n * nchunks + chunk) * C * 2;
  for (int cv = t - p0 * CVs; cv < CV; cv += CVs) {
    float s[VE], ss[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s[e] = ss[e] = 0.f;
    if (p0 < PIF) {
      const bool second = cv >= CV0;
      const char* base = second ? src1 : src0;
      const char* lob = LO ? (second ? lo1 : lo0) : nullptr;   // statistics of a tensor with a lo plane describe hi + lo
      const int Cs = second ? C1 : C0;
      const int cc = (second ? cv - CV0 : cv) * VE;
      for (int p = pbeg + p0; p < pend; p += PIF) {
        const vec_t v = *(const vec_t*)(base + (((size_t)n * HW + p) * Cs + cc) * sizeof(T));
        float f[VE];
        vec_to_f32<T>(v, f);
        if constexpr (LO) {
          if (lob) {
            float l[VE];
            vec_to_f32<T>(*(const vec_t*)(lob + (((size_t)n * HW + p) * Cs + cc) * sizeof(T)), l);
#pragma unroll
            for (int e = 0; e < VE; ++e) f[e] += l[e];
          }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          s[e] += f[e];
          ss[e] += f[e] * f[e];
        }
      }
    }
    if (PIF == 1) {
      if (p0 == 0) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          outp[(cv * VE + e) * 2] = s[e];
          outp[(cv * VE + e) * 2 + 1] = ss[e];
        }
      }
    } else {
      // reduce over the PIF threads that share this channel piece (loop body runs once here: CV <= NT)
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        red[(t * VE + e) * 2] = s[e];
        red[(t * VE + e) * 2 + 1] = ss[e];
      }
      __syncthreads();
      if (p0 == 0) {
        for (int j = 1; j < PIF; ++j) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            s[e] += red[((t + j * CVs) * VE + e) * 2];
            ss[e] += red[((t + j * CVs) * VE + e) * 2 + 1];
          }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          outp[(cv * VE + e) * 2] = s[e];
          outp[(cv * VE + e) * 2 + 1] = ss[e];
        }
      }
    }
  }
}

constexpr int FIN_NT = 128;   // round 6: 16 blocks per CU resident = the 4096 blocks of a batch-128 launch in ONE round (256 threads: two
                              // rounds): 6.8 -> 4.7 us on the small tensors, 16.6 -> 14.9 on the 128^2 concat; 64 threads: slower again

// One block per (image, group).  The partial sums of a group are a [nchunks][cpg] matrix per source tensor (row = 8*cpg
// contiguous bytes inside a [nchunks][C][2] array): thread t takes channel t % cpg of chunks t / cpg, t / cpg + 256/cpg, ...
// so that a wave reads whole rows and a thread walks at most 4 loads back to back (the 64-thread version of round 1 walked up
// to 16: 9.7 us per launch by rocprofv3, 87 launches per forward; this one 6.2 us; ONE 1024-thread block per image, 32 threads
// per group, measured 7.7 us -- the time is the serial load -> fp64-add chain, not the dispatch of the 4096 small blocks).  fp64 accumulation;
// the cross-thread reduction is a fixed tree (shuffles, then one LDS pass over the 4 waves): deterministic.
__global__ __launch_bounds__(FIN_NT) void gn_finalize_kernel(const float* __restrict__ partial, int C0,
                                                             const float* __restrict__ partial1, int nchunks,
                                                             int nchunks1, int C,
                                                             int HW, int groups, float eps,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ film, int film_stride,
                                                             int film_off, float* __restrict__ ab) {
  __shared__ double red[2][FIN_NT / 64];
  const int g = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int cpg = C / groups;
  const int C1 = C - C0;
  // the two concat sources may come with different block counts (their producers used different tiles)
  const float* pp0 = partial + (size_t)n * nchunks * C0 * 2;
  const float* pp1 = partial1 ? partial1 + (size_t)n * nchunks1 * C1 * 2 : nullptr;
  const int nmax = nchunks > nchunks1 ? nchunks : nchunks1;
  double s = 0.0, ss = 0.0;
  for (int idx = t; idx < nmax * cpg; idx += FIN_NT) {
    const int ch = idx / cpg, c = g * cpg + (idx - ch * cpg);
    const float* q;
    if (c < C0) {
      if (ch >= nchunks) continue;
      q = pp0 + ((size_t)ch * C0 + c) * 2;
    } else {
      if (ch >= nchunks1) continue;
      q = pp1 + ((size_t)ch * C1 + (c - C0)) * 2;
    }
    const f32x2 v = *(const f32x2*)q;
    s += (double)v[0];
    ss += (double)v[1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    ss += __shfl_xor(ss, off);
  }
  if ((t & 63) == 0) { red[0][t >> 6] = s; red[1][t >> 6] = ss; }
  __syncthreads();
  s = red[0][0]; ss = red[1][0];
#pragma unroll
  for (int w = 1; w < FIN_NT / 64; ++w) { s += red[0][w]; ss += red[1][w]; }
  const double cnt = (double)cpg * (double)HW;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  for (int j = t; j < cpg; j += FIN_NT) {
    const int c = g * cpg + j;
    float a = rstd * gamma[c];
    float b = beta[c] - meanf * a;
    if (film) {
      const float sc = 1.0f + film[(size_t)n * film_stride + film_off + c];
      const float sh = film[(size_t)n * film_stride + film_off + C + c];
      a *= sc;
      b = b * sc + sh;
    }
    ab[((size_t)n * C + c) * 2] = a;
    ab[((size_t)n * C + c) * 2 + 1] = b;
  }
}

// resample: 0 same, 1 nearest x2 up (output 2H x 2W), 2 avg-pool x2 of activated values (output H/2 x W/2)
// LO: the sources may carry lo planes (compensated 16-bit storage, see conv_igemm.hip ConvArgs): x = hi + lo
// POOL (resample 2 only): the 2x2 average of the RAW inputs is written as well (pool_hi, and pool_lo = what the 16-bit rounding of
// pool_hi dropped, if given) -- the residual `x_upd(x)` of a `down` ResBlock (adm.py:205-208), so that the block's second
// convolution adds a same-size residual instead of averaging four pixels of the full-size tensor per output in its epilogue.
template <typename T, int ACT, bool LO = false, bool POOL = false>
__global__ __launch_bounds__(GN_NT) void gn_apply_kernel(const char* __restrict__ src0, int C0,
                                                         const char* __restrict__ src1, int C1,
                                                         const float* __restrict__ ab, char* __restrict__ out, int H,
                                                         int W, int resample, int ppc, const char* __restrict__ lo0 = nullptr,
                                                         const char* __restrict__ lo1 = nullptr, char* __restrict__ pool_hi = nullptr,
                                                         char* __restrict__ pool_lo = nullptr) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  const int C = C0 + C1, CV = C / VE, CV0 = C0 / VE;
  const int Ho = resample == 1 ? H * 2 : (resample == 2 ? H / 2 : H);
  const int Wo = resample == 1 ? W * 2 : (resample == 2 ? W / 2 : W);
  const int HWo = Ho * Wo;
  const int chunk = blockIdx.x, n = blockIdx.y, t = threadIdx.x;
  const int CVs = CV < GN_NT ? CV : GN_NT;
  const int PIF = CV < GN_NT ? GN_NT / CV : 1;
  const int p0 = t / CVs;
  if (p0 >= PIF) return;
  const int pbeg = chunk * ppc, pend = min(pbeg + ppc, HWo);
  for (int cv = t - p0 * CVs; cv < CV; cv += CVs) {
    float a[VE], b[VE];
    {
      const float* abp = ab + ((size_t)n * C + (size_t)cv * VE) * 2;
#pragma unroll
      for (int e = 0; e < VE; e += 2) {
        const f32x4 q = *(const f32x4*)(abp + e * 2);
        a[e] = q[0]; b[e] = q[1]; a[e + 1] = q[2]; b[e + 1] = q[3];
      }
    }
    const bool second = cv >= CV0;
    const char* base = second ? src1 : src0;
    const char* lob = LO ? (second ? lo1 : lo0) : nullptr;
    const int Cs = second ? C1 : C0;
    const int cc = (second ? cv - CV0 : cv) * VE;
    auto load = [&](size_t sp, float* f) {
      vec_to_f32<T>(*(const vec_t*)(base + (sp * Cs + cc) * sizeof(T)), f);
      if constexpr (LO) {
        if (lob) {
          float l[VE];
          vec_to_f32<T>(*(const vec_t*)(lob + (sp * Cs + cc) * sizeof(T)), l);
#pragma unroll
          for (int e = 0; e < VE; ++e) f[e] += l[e];
        }
      }
    };
    for (int p = pbeg + p0; p < pend; p += PIF) {
      float r[VE];
      if (resample == 2) {
        const int yo = p / Wo, xo = p - yo * Wo;
        float raw[POOL ? VE : 1];
#pragma unroll
        for (int e = 0; e < VE; ++e) r[e] = 0.f;
        if constexpr (POOL) {
#pragma unroll
          for (int e = 0; e < VE; ++e) raw[e] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const size_t sp = ((size_t)n * H + 2 * yo + (d >> 1)) * W + 2 * xo + (d & 1);
          float f[VE];
          load(sp, f);
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const float y = f[e] * a[e] + b[e];
            r[e] += ACT ? silu_f(y) : y;
            if constexpr (POOL) raw[e] += f[e];      // same order as the convolution epilogue's res_mode 3
          }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) r[e] *= 0.25f;
        if constexpr (POOL) {
          const size_t po = (((size_t)n * HWo + p) * C + (size_t)cv * VE) * sizeof(T);
#pragma unroll
          for (int e = 0; e < VE; ++e) raw[e] *= 0.25f;
          const vec_t ph = f32_to_vec<T>(raw);
          *(vec_t*)(pool_hi + po) = ph;
          if (pool_lo) {
            float g[VE];
            vec_to_f32<T>(ph, g);
#pragma unroll
            for (int e = 0; e < VE; ++e) g[e] = raw[e] - g[e];
            *(vec_t*)(pool_lo + po) = f32_to_vec<T>(g);
          }
        }
      } else {
        size_t sp;
        if (resample == 1) {
          const int yo = p / Wo, xo = p - yo * Wo;
          sp = ((size_t)n * H + (yo >> 1)) * W + (xo >> 1);
        } else {
          sp = (size_t)n * HWo + p;
        }
        float f[VE];
        load(sp, f);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float y = f[e] * a[e] + b[e];
          r[e] = ACT ? silu_f(y) : y;
        }
      }
      *(vec_t*)(out + (((size_t)n * HWo + p) * C + (size_t)cv * VE) * sizeof(T)) = f32_to_vec<T>(r);
    }
  }
}

int check_channels(int dtype, int C0, int C1, const void* src1) {
  if (!ivid_esz(dtype)) return ivid_set_error("gn: bad dtype", hipSuccess);
  const int ve = 16 / ivid_esz(dtype);
  if (C0 <= 0 || C0 % ve || C1 < 0 || C1 % ve) return ivid_set_error("gn: channels must be multiples of 16 bytes", hipSuccess);
  if (C1 > 0 && !src1) return ivid_set_error("gn: src1 missing", hipSuccess);
  return 0;
}

}  // namespace

extern "C" int ivid_gn_num_chunks(int HW) {
  const int ppc = gn_ppc(HW);
  return (HW + ppc - 1) / ppc;
}

extern "C" int ivid_gn_partial(int dtype, const void* src0, int C0, const void* src1, int C1, int N, int HW,
                               float* partial, void* stream) {
  return ivid_gn_partial_c(dtype, src0, nullptr, C0, src1, nullptr, C1, N, HW, partial, stream);
}

// with the lo planes of the sources (compensated 16-bit storage): the statistics describe hi + lo, like the ones the
// convolution epilogues write
extern "C" int ivid_gn_partial_c(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                                 int C1, int N, int HW, float* partial, void* stream) {
  if (int e = check_channels(dtype, C0, C1, src1)) return e;
  const int ppc = gn_ppc(HW), nchunks = (HW + ppc - 1) / ppc;
  dim3 grid(nchunks, N);
  if (src0_lo || src1_lo) {
    if (ivid_esz(dtype) != 2) return ivid_set_error("gn_partial: lo planes need a 16-bit dtype", hipSuccess);
    if (dtype == IVID_F16)
      hipLaunchKernelGGL((gn_partial_kernel<_Float16, true>), grid, dim3(GN_NT), 0, (hipStream_t)stream, (const char*)src0, C0,
                         (const char*)src1, C1, HW, ppc, nchunks, partial, (const char*)src0_lo, (const char*)src1_lo);
    else
      hipLaunchKernelGGL((gn_partial_kernel<__bf16, true>), grid, dim3(GN_NT), 0, (hipStream_t)stream, (const char*)src0, C0,
                         (const char*)src1, C1, HW, ppc, nchunks, partial, (const char*)src0_lo, (const char*)src1_lo);
    return ivid_check_launch("gn_partial");
  }
  if (dtype == IVID_F32 || dtype == IVID_BF16X3)
    hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(GN_NT), 0, (hipStream_t)stream, (const char*)src0, C0,
                       (const char*)src1, C1, HW, ppc, nchunks, partial);
  else if (dtype == IVID_F16)
    hipLaunchKernelGGL(gn_partial_kernel<_Float16>, grid, dim3(GN_NT), 0, (hipStream_t)stream, (const char*)src0, C0,
                       (const char*)src1, C1, HW, ppc, nchunks, partial);
  else
    hipLaunchKernelGGL(gn_partial_kernel<__bf16>, grid, dim3(GN_NT), 0, (hipStream_t)stream, (const char*)src0, C0,
                       (const char*)src1, C1, HW, ppc, nchunks, partial);
  return ivid_check_launch("gn_partial");
}

extern "C" int ivid_gn_finalize(const float* partial, int nchunks, int N, int C, int HW, int groups, float eps,
                                const float* gamma, const float* beta, const float* film, int film_stride,
                                int film_off, float* ab, void* stream) {
  if (groups <= 0 || C % groups) return ivid_set_error("gn_finalize: C must be divisible by groups", hipSuccess);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(FIN_NT), 0, (hipStream_t)stream, partial, C, nullptr, nchunks,
                       0, C, HW, groups, eps, gamma, beta, film, film_stride, film_off, ab);
  return ivid_check_launch("gn_finalize");
}

extern "C" int ivid_gn_finalize2(const float* partial0, int C0, int nchunks0, const float* partial1, int C1, int nchunks1,
                                 int N, int HW,
                                 int groups, float eps, const float* gamma, const float* beta, const float* film,
                                 int film_stride, int film_off, float* ab, void* stream) {
  const int C = C0 + C1;
  if (groups <= 0 || C % groups) return ivid_set_error("gn_finalize2: C must be divisible by groups", hipSuccess);
  if (C1 > 0 && !partial1) return ivid_set_error("gn_finalize2: partial1 missing", hipSuccess);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(FIN_NT), 0, (hipStream_t)stream, partial0, C0, partial1, nchunks0,
                       nchunks1, C, HW, groups, eps, gamma, beta, film, film_stride, film_off, ab);
  return ivid_check_launch("gn_finalize2");
}

extern "C" int ivid_gn_apply(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, void* out,
                             int N, int H, int W, int resample, int act, void* stream) {
  return ivid_gn_apply_c(dtype, src0, nullptr, C0, src1, nullptr, C1, ab, out, N, H, W, resample, act, stream);
}

// with optional lo planes of the two sources (compensated 16-bit storage, precision mode fp16c); the output is a plain tensor
extern "C" int ivid_gn_apply_c(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                               int C1, const float* ab, void* out, int N, int H, int W, int resample, int act, void* stream) {
  return ivid_gn_apply_p(dtype, src0, src0_lo, C0, src1, src1_lo, C1, ab, out, nullptr, nullptr, N, H, W, resample, act, stream);
}

// ... and, for resample 2 (2x2 average pool), the pooled RAW input as a second output (pool_hi [+ pool_lo]): the residual of a
// `down` ResBlock; 16-bit dtypes only
extern "C" int ivid_gn_apply_p(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                               int C1, const float* ab, void* out, void* pool_hi, void* pool_lo, int N, int H, int W, int resample,
                               int act, void* stream) {
  if (int e = check_channels(dtype, C0, C1, src1)) return e;
  if (pool_lo && !pool_hi) return ivid_set_error("gn_apply: pool_lo without pool_hi", hipSuccess);
  if (pool_hi && (resample != 2 || ivid_esz(dtype) != 2))
    return ivid_set_error("gn_apply: the pooled raw output exists for resample 2 in the 16-bit dtypes", hipSuccess);
  const bool lo = src0_lo || src1_lo;
  if (lo && ivid_esz(dtype) != 2) return ivid_set_error("gn_apply: lo planes need a 16-bit dtype", hipSuccess);
  if (resample < 0 || resample > 2) return ivid_set_error("gn_apply: bad resample", hipSuccess);
  if (resample == 2 && ((H | W) & 1)) return ivid_set_error("gn_apply: avg-pool needs even H,W", hipSuccess);
  const int Ho = resample == 1 ? H * 2 : (resample == 2 ? H / 2 : H);
  const int Wo = resample == 1 ? W * 2 : (resample == 2 ? W / 2 : W);
  const int HWo = Ho * Wo;
  // pixels per block: an elementwise pass has no layout tied to its chunks, so small tensors (16^2, 8^2: 2 - 50 MB, one or four
  // chunks of 64 pixels per image = 128 - 512 blocks whose threads walk 32 dependent load -> store iterations) are cut finer, down
  // to 8 pixels per block, until the grid holds >= 8 blocks per CU (round 6: 8^2 x 1024 29 -> ~13 us, 16^2 x 768 50 -> ~25 us)
  int ppc = gn_ppc(HWo);
  while (ppc > 8 && (long long)((HWo + ppc - 1) / ppc) * N < 2048) ppc >>= 1;
  const int nchunks = (HWo + ppc - 1) / ppc;
  dim3 grid(nchunks, N);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T, A)                                                                                              \
  hipLaunchKernelGGL((gn_apply_kernel<T, A>), grid, dim3(GN_NT), 0, s, (const char*)src0, C0, (const char*)src1, \
                     C1, ab, (char*)out, H, W, resample, ppc)
#define LAUNCH_LO(T, A)                                                                                                 \
  hipLaunchKernelGGL((gn_apply_kernel<T, A, true>), grid, dim3(GN_NT), 0, s, (const char*)src0, C0, (const char*)src1, \
                     C1, ab, (char*)out, H, W, resample, ppc, (const char*)src0_lo, (const char*)src1_lo)
#define LAUNCH_POOL(T, A)                                                                                                     \
  hipLaunchKernelGGL((gn_apply_kernel<T, A, true, true>), grid, dim3(GN_NT), 0, s, (const char*)src0, C0, (const char*)src1, \
                     C1, ab, (char*)out, H, W, resample, ppc, (const char*)src0_lo, (const char*)src1_lo, (char*)pool_hi,    \
                     (char*)pool_lo)
  if (pool_hi) {
    if (dtype == IVID_F16) { if (act) LAUNCH_POOL(_Float16, 1); else LAUNCH_POOL(_Float16, 0); }
    else { if (act) LAUNCH_POOL(__bf16, 1); else LAUNCH_POOL(__bf16, 0); }
  } else if (lo) {
    if (dtype == IVID_F16) { if (act) LAUNCH_LO(_Float16, 1); else LAUNCH_LO(_Float16, 0); }
    else { if (act) LAUNCH_LO(__bf16, 1); else LAUNCH_LO(__bf16, 0); }
  } else if (dtype == IVID_F32 || dtype == IVID_BF16X3) {
    if (act) LAUNCH(float, 1); else LAUNCH(float, 0);
  } else if (dtype == IVID_F16) {
    if (act) LAUNCH(_Float16, 1); else LAUNCH(_Float16, 0);
  } else {
    if (act) LAUNCH(__bf16, 1); else LAUNCH(__bf16, 0);
  }
#undef LAUNCH
#undef LAUNCH_LO
#undef LAUNCH_POOL
  return ivid_check_launch("gn_apply");
}
