// Small streaming kernels around the UNet and the DDPM/DDIM update (all HBM-bound, fp32 math).
//   - PosEncoding + label embedding gather with null-class mask   (adm.py:30-33, 545-555)
//   - SiLU on the fp32 embedding vectors                           (adm.py:175, 360)
//   - fp32 NCHW -> NHWC (dtype, zero-padded channels, CFG batch replication) at the model boundary
//   - one fused DDIM step: CFG combine, x0 prediction, clamp, replace_rgb / replace_depth /
//     convex-hull depth constraint, eps re-derivation, x_{t-1}       (ddim.py:81-102,
//     classifier_free_guidance.py:39-42)
//   - one fused DDPM ancestral step                                 (ddpm.py:85-101, 127-131)
//   - InpaintCFG.make_cond_inputs                                   (inpaint_cfg.py:24-49)
#include "common.h"
#include "internal.h"

namespace {

__global__ void embed_inputs_kernel(const int64_t* __restrict__ times, const int64_t* __restrict__ classes, int Bsrc,
                                    int null_from, const float* __restrict__ freqs, int half,
                                    const float* __restrict__ label_emb, int emb_dim, float* __restrict__ pos,
                                    float* __restrict__ cls) {
  const int n = blockIdx.x, src = n % Bsrc;
  const float t = (float)times[src];
  for (int k = threadIdx.x; k < half; k += blockDim.x) {
    const float a = t * freqs[k];
    pos[(size_t)n * 2 * half + k] = cosf(a);
    pos[(size_t)n * 2 * half + half + k] = sinf(a);
  }
  if (cls) {
    long long c = -1;
    if (classes && n < null_from) c = classes[src];
    for (int k = threadIdx.x; k < emb_dim; k += blockDim.x)
      cls[(size_t)n * emb_dim + k] = c >= 0 ? label_emb[(size_t)c * emb_dim + k] : 0.f;
  }
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = x[i];
    y[i] = v / (1.0f + expf(-v));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int Bsrc, int Cin, int HW,
                                                           int Cpad, char* __restrict__ out) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  const int p = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (p >= HW) return;
  const float* xs = x + (size_t)(n % Bsrc) * Cin * HW + p;
  char* o = out + ((size_t)n * HW + p) * Cpad * sizeof(T);
  for (int c0 = 0; c0 < Cpad; c0 += VE) {
    float f[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) f[e] = (c0 + e < Cin) ? xs[(size_t)(c0 + e) * HW] : 0.f;
    *(vec_t*)(o + c0 * sizeof(T)) = f32_to_vec<T>(f);
  }
}

// Stem im2col: the model input has only 4..10 channels, so a 3x3 implicit GEMM over a channel-padded NHWC copy spends 9
// K-steps on 94 % zeros.  Instead the 3x3 patch of every pixel is laid out as ONE K row  k = tap*Cin + c  (zero for padded
// taps and for k >= 9 Cin), and the stem becomes a 1x1 GEMM with K = Kpad (one K-step for Cin <= 7).
// SPLIT (16-bit types): the K row holds three segments of 9*Cin values  [x_hi | x_lo | x_hi]  (x_hi = T(x), x_lo = T(x - x_hi));
// against weight rows  [w_hi | w_hi | w_lo]  the GEMM then accumulates x_hi*w_hi + x_lo*w_hi + x_hi*w_lo, i.e. the stem
// convolution to ~2^-21 instead of the 2^-11 of a single 16-bit product -- for two more K-steps of the cheapest layer.
// One thread = one 16-byte piece of one pixel's K row, the pieces of a row on consecutive lanes: a wave writes whole rows
// (round 6; one thread per PIXEL wrote 16 bytes at a stride of a row per lane and iteration: 1.1 TB/s on a 537 MB output).
// CIN: the model's input channels as a compile-time constant (4 = RGBD, 9 / 10 = the inpainting models), 0 = any.
template <typename T, bool SPLIT = false, int CIN = 0>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, int Bsrc, int CinRt, int H, int W,
                                                          int Kpad, char* __restrict__ out) {
  typedef typename Elem<T>::vec vec_t;
  constexpr int VE = Elem<T>::VE;
  const int Cin = CIN ? CIN : CinRt;
  const int HW = H * W, ppr = Kpad / VE;
  const int t = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  const int p = t / ppr, piece = t - p * ppr;
  if (p >= HW) return;
  const int y = p / W, xx = p - y * W;
  const float* xs = x + (size_t)(n % Bsrc) * Cin * HW;
  char* o = out + ((size_t)n * HW + p) * Kpad * sizeof(T);
  const int K = 9 * Cin;
  {
    const int k0 = piece * VE;
    float f[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      int k = k0 + e, seg = 0;
      if constexpr (SPLIT) {
        seg = k / K;
        k -= seg * K;
        if (seg > 2) k = K;   // padding behind the third segment
      }
      float v = 0.f;
      if (k < K) {
        const int tap = k / Cin, c = k - tap * Cin;
        const int yy = y + tap / 3 - 1, xc = xx + tap % 3 - 1;
        if ((unsigned)yy < (unsigned)H && (unsigned)xc < (unsigned)W) v = xs[(size_t)c * HW + yy * W + xc];
      }
      if constexpr (SPLIT) {
        if (seg == 1) v -= (float)(T)v;   // lo part (exact difference; rounded once by the store below)
      }
      f[e] = v;
    }
    *(vec_t*)(o + k0 * sizeof(T)) = f32_to_vec<T>(f);
  }
}

struct StepPtrs {
  const float *x_t, *eps_c, *eps_u, *rgb, *rgb_mask, *depth, *depth_mask, *convex, *noise;
  float *x_prev, *x0;
};

// One thread = one pixel of one sample, all 4 RGBD channels (the replace / constrain terms differ per channel).
__global__ __launch_bounds__(256) void ddim_step_kernel(StepPtrs q, ivid_ddim_coef k, int HW) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (p >= HW) return;
  const size_t b4 = (size_t)n * 4 * HW + p, b3 = (size_t)n * 3 * HW + p, b1 = (size_t)n * HW + p;
  float xt[4], x0[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    xt[c] = q.x_t[b4 + (size_t)c * HW];
    float e = q.eps_c[b4 + (size_t)c * HW];
    if (q.eps_u) e = (1.0f + k.cfg_strength) * e - k.cfg_strength * q.eps_u[b4 + (size_t)c * HW];
    float v = k.sqrt_recip_ac * xt[c] - k.sqrt_recipm1_ac * e;  // ddim.py:36-37
    if (k.clip_denoised) v = fminf(fmaxf(v, -1.0f), 1.0f);
    x0[c] = v;
  }
  if (k.replace_rgb_w >= 0.f) {  // ddim.py:86-89 (active only while t_prev != 0)
    const float m = q.rgb_mask[b1], w = k.replace_rgb_w, nz = k.nonzero;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = q.rgb[b3 + (size_t)c * HW];
      x0[c] = (1.0f - nz) * x0[c] + nz * ((w * r + (1.0f - w) * x0[c]) * m + x0[c] * (1.0f - m));
    }
  }
  if (k.replace_depth_w >= 0.f) {  // ddim.py:90-95
    const float m = q.depth_mask[b1], w = k.replace_depth_w;
    x0[3] = (w * q.depth[b1] + (1.0f - w) * x0[3]) * m + x0[3] * (1.0f - m);
    if (k.constrain_w >= 0.f) {
      const float cw = k.constrain_w;
      x0[3] = x0[3] * m + (cw * fmaxf(x0[3], q.convex[b1]) + (1.0f - cw) * x0[3]) * (1.0f - m);
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float e2 = (k.sqrt_recip_ac * xt[c] - x0[c]) / k.sqrt_recipm1_ac;  // ddim.py:43-44
    float v = k.sqrt_ac_prev * x0[c] + k.dir_coef * e2;                      // ddim.py:99
    if (q.noise) v += k.nonzero * k.sigma * q.noise[b4 + (size_t)c * HW];    // ddim.py:101-102
    q.x_prev[b4 + (size_t)c * HW] = v;
    q.x0[b4 + (size_t)c * HW] = x0[c];
  }
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(StepPtrs q, ivid_ddpm_coef k, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float xt = q.x_t[i];
  float e = q.eps_c[i];
  if (q.eps_u) e = (1.0f + k.cfg_strength) * e - k.cfg_strength * q.eps_u[i];
  float x0 = k.sqrt_recip_ac * xt - k.sqrt_recipm1_ac * e;  // ddpm.py:105-108
  if (k.clip_denoised) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
  float v = k.coef1 * x0 + k.coef2 * xt;  // ddpm.py:57-60
  if (q.noise) v += k.std * q.noise[i];   // ddpm.py:130
  q.x_prev[i] = v;
  q.x0[i] = x0;
}

__global__ __launch_bounds__(256) void inpaint_cond_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ mask,
                                                           const float* __restrict__ mask_rgb,
                                                           const float* __restrict__ nrgb,
                                                           const float* __restrict__ nd, float* __restrict__ out,
                                                           int HW) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (p >= HW) return;
  const int Co = mask_rgb ? 10 : 9;
  const size_t b4 = (size_t)n * 4 * HW + p, b3 = (size_t)n * 3 * HW + p, b1 = (size_t)n * HW + p;
  float* o = out + (size_t)n * Co * HW + p;
  int c = 0;
  for (int j = 0; j < 4; ++j) o[(size_t)(c++) * HW] = x[b4 + (size_t)j * HW];
  const float m = mask[b1];
  const float mr = mask_rgb ? mask_rgb[b1] : m;
  if (mask_rgb) o[(size_t)(c++) * HW] = mr;
  for (int j = 0; j < 3; ++j)
    o[(size_t)(c++) * HW] = y[b4 + (size_t)j * HW] * mr + nrgb[b3 + (size_t)j * HW] * (1.0f - mr);
  o[(size_t)(c++) * HW] = y[b4 + (size_t)3 * HW] * m + nd[b1] * (1.0f - m);
  o[(size_t)(c++) * HW] = m;
}

}  // namespace

extern "C" int ivid_embed_inputs(const int64_t* times, const int64_t* classes, int Bsrc, int N, int null_from,
                                 const float* freqs, int half, const float* label_emb, int emb_dim, float* pos,
                                 float* cls, void* stream) {
  if (Bsrc <= 0 || N <= 0) return ivid_set_error("embed_inputs: bad batch", hipSuccess);
  if (cls && classes && !label_emb) return ivid_set_error("embed_inputs: label_emb missing", hipSuccess);
  hipLaunchKernelGGL(embed_inputs_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, times, classes, Bsrc, null_from,
                     freqs, half, label_emb, emb_dim, pos, cls);
  return ivid_check_launch("embed_inputs");
}

// 16 bytes per lane, grid-stride: a plain device copy that stays a kernel node of the forward's hipGraph
__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long n16) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

// fp32 NHWC tensor -> two 16-bit planes hi = T(x), lo = T(x - hi) (the compensated storage of the fp16c / fp16s modes).  Used where
// an island of split-precision layers (fp32 storage, three MFMA passes per product) hands its tensors to the 16-bit part
// of the network (precision mode fp16s: the first encoder level).
template <typename T>
__global__ __launch_bounds__(256) void f32_to_hilo_kernel(const f32x4* __restrict__ src, char* __restrict__ hi,
                                                          char* __restrict__ lo, long long n8) {
  typedef typename Elem<T>::vec vec_t;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = src[2 * i], b = src[2 * i + 1];
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}, g[8];
    const vec_t h = f32_to_vec<T>(f);
    vec_to_f32<T>(h, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = f[e] - g[e];
    *(vec_t*)(hi + i * 16) = h;
    *(vec_t*)(lo + i * 16) = f32_to_vec<T>(g);
  }
}

extern "C" int ivid_f32_to_hilo(int dtype, const float* src, void* hi, void* lo, long long n, void* stream) {
  if (dtype != IVID_F16 && dtype != IVID_BF16) return ivid_set_error("f32_to_hilo: needs a 16-bit dtype", hipSuccess);
  if (n < 0 || n % 8 || !src || !hi || !lo || ((uintptr_t)src | (uintptr_t)hi | (uintptr_t)lo) % 16)
    return ivid_set_error("f32_to_hilo: n % 8 == 0 and 16-byte aligned pointers", hipSuccess);
  if (!n) return 0;
  const long long n8 = n / 8;
  long long blocks = (n8 + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (dtype == IVID_F16)
    hipLaunchKernelGGL(f32_to_hilo_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src,
                       (char*)hi, (char*)lo, n8);
  else
    hipLaunchKernelGGL(f32_to_hilo_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src,
                       (char*)hi, (char*)lo, n8);
  return ivid_check_launch("f32_to_hilo");
}

// Device copy of `bytes` (a multiple of 16, both pointers 16-byte aligned).  Used by the stacked classifier-free-guidance
// forward: up to the first FiLM (adm.py:214-218) the conditional and the unconditional half of the batch see identical inputs
// (same x, same t, no class dependence before the embedding is applied), so the first ResBlock's in_layers convolution runs on
// one half and its output is duplicated (classifier_free_guidance.py:39-42 runs the backbone twice instead).
extern "C" int ivid_copy(void* dst, const void* src, long long bytes, void* stream) {
  if (bytes < 0 || bytes % 16 || ((uintptr_t)dst | (uintptr_t)src) % 16) return ivid_set_error("copy: 16-byte granularity", hipSuccess);
  if (!bytes) return 0;
  const long long n16 = bytes / 16;
  long long blocks = (n16 + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, n16);
  return ivid_check_launch("copy");
}

// ---- counter-based Gaussian noise (ivid_randn): Philox4x32-10 (Salmon et al., SC'11) + Box-Muller ----
// Output block j (values 4j .. 4j+3) of stream `sid` under key `seed` = Philox(counter = (j_lo, j_hi, sid_lo, sid_hi), key = (seed_lo,
// seed_hi)); a 32-bit word r becomes the uniform u = ((r >> 8) + 0.5) * 2^-24 in (0, 1) -- exact in fp32 --, pairs (u0, u1) become
// sqrt(-2 ln u0) * (cos, sin)(2 pi u1).  A value depends on (seed, sid, index) alone: any launch shape, any rank, any order.
namespace {
__device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__global__ __launch_bounds__(256) void randn_kernel(unsigned long long seed, unsigned long long sid, float* __restrict__ out, long long n) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (4 * j >= n) return;
  uint32_t c[4] = {(uint32_t)j, (uint32_t)((unsigned long long)j >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u0 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u0));
    float sn, cs;
    sincospif(2.0f * u1, &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
  if (4 * j + 4 <= n) {
    *(f32x4*)(out + 4 * j) = f32x4{z[0], z[1], z[2], z[3]};
  } else {
    for (int e = 0; 4 * j + e < n; ++e) out[4 * j + e] = z[e];
  }
}
}  // namespace

extern "C" int ivid_randn(unsigned long long seed, unsigned long long stream_id, float* out, long long n, void* stream) {
  if (!out || n < 0 || ((uintptr_t)out & 15)) return ivid_set_error("randn: out must be a 16-byte aligned device pointer", hipSuccess);
  if (n == 0) return 0;
  const long long blocks = (n + 1023) / 1024;
  if (blocks >= (1ll << 31)) return ivid_set_error("randn: n too large", hipSuccess);
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seed, stream_id, out, n);
  return ivid_check_launch("randn");
}

extern "C" int ivid_silu_f32(const float* x, float* y, long long n, void* stream) {
  hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return ivid_check_launch("silu");
}

extern "C" int ivid_nchw_to_nhwc(int dtype, const float* x, int Bsrc, int N, int Cin, int H, int W, int Cpad,
                                 void* out, void* stream) {
  if (!ivid_esz(dtype)) return ivid_set_error("nchw_to_nhwc: bad dtype", hipSuccess);
  const int ve = 16 / ivid_esz(dtype);
  if (Cpad % ve || Cpad < Cin) return ivid_set_error("nchw_to_nhwc: bad Cpad", hipSuccess);
  const int HW = H * W;
  dim3 grid((HW + 255) / 256, N);
  if (dtype == IVID_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, x, Bsrc, Cin, HW, Cpad,
                       (char*)out);
  else if (dtype == IVID_F32 || dtype == IVID_BF16X3)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, Bsrc, Cin, HW, Cpad,
                       (char*)out);
  else if (dtype == IVID_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, Bsrc, Cin, HW, Cpad,
                       (char*)out);
  else
    return ivid_set_error("nchw_to_nhwc: bad dtype", hipSuccess);
  return ivid_check_launch("nchw_to_nhwc");
}

template <typename T, bool SPLIT>
static void launch_stem(const float* x, int Bsrc, int N, int Cin, int H, int W, int Kpad, void* out, void* stream) {
  const int ppr = Kpad / Elem<T>::VE;
  dim3 grid((unsigned)(((long long)H * W * ppr + 255) / 256), N);
  hipStream_t s = (hipStream_t)stream;
  if (Cin == 4) hipLaunchKernelGGL((stem_im2col_kernel<T, SPLIT, 4>), grid, dim3(256), 0, s, x, Bsrc, Cin, H, W, Kpad, (char*)out);
  else if (Cin == 10) hipLaunchKernelGGL((stem_im2col_kernel<T, SPLIT, 10>), grid, dim3(256), 0, s, x, Bsrc, Cin, H, W, Kpad, (char*)out);
  else hipLaunchKernelGGL((stem_im2col_kernel<T, SPLIT, 0>), grid, dim3(256), 0, s, x, Bsrc, Cin, H, W, Kpad, (char*)out);
}

extern "C" int ivid_stem_im2col(int dtype, const float* x, int Bsrc, int N, int Cin, int H, int W, int Kpad, void* out,
                                void* stream) {
  if (!ivid_esz(dtype)) return ivid_set_error("stem_im2col: bad dtype", hipSuccess);
  const int ve = 16 / ivid_esz(dtype);
  if (Kpad % ve || Kpad < 9 * Cin || Cin <= 0) return ivid_set_error("stem_im2col: bad Kpad", hipSuccess);
  if ((long long)H * W * (Kpad / ve) >= (1ll << 31)) return ivid_set_error("stem_im2col: image too large", hipSuccess);
  if (dtype == IVID_F16) launch_stem<_Float16, false>(x, Bsrc, N, Cin, H, W, Kpad, out, stream);
  else if (dtype == IVID_F32 || dtype == IVID_BF16X3) launch_stem<float, false>(x, Bsrc, N, Cin, H, W, Kpad, out, stream);
  else if (dtype == IVID_BF16) launch_stem<__bf16, false>(x, Bsrc, N, Cin, H, W, Kpad, out, stream);
  else return ivid_set_error("stem_im2col: bad dtype", hipSuccess);
  return ivid_check_launch("stem_im2col");
}

extern "C" int ivid_stem_im2col_split(int dtype, const float* x, int Bsrc, int N, int Cin, int H, int W, int Kpad, void* out,
                                      void* stream) {
  if (ivid_esz(dtype) != 2) return ivid_set_error("stem_im2col_split: 16-bit dtypes only", hipSuccess);
  if (Kpad % 8 || Kpad < 27 * Cin || Cin <= 0) return ivid_set_error("stem_im2col_split: bad Kpad", hipSuccess);
  if ((long long)H * W * (Kpad / 8) >= (1ll << 31)) return ivid_set_error("stem_im2col_split: image too large", hipSuccess);
  if (dtype == IVID_F16) launch_stem<_Float16, true>(x, Bsrc, N, Cin, H, W, Kpad, out, stream);
  else launch_stem<__bf16, true>(x, Bsrc, N, Cin, H, W, Kpad, out, stream);
  return ivid_check_launch("stem_im2col_split");
}

extern "C" int ivid_ddim_step(const float* x_t, const float* eps_c, const float* eps_u, const ivid_ddim_coef* host_coef,
                              const float* rgb, const float* rgb_mask, const float* depth, const float* depth_mask,
                              const float* convex, const float* noise, float* x_prev, float* x0, int B, int HW,
                              void* stream) {
  ivid_ddim_coef k = *host_coef;
  if (k.replace_rgb_w >= 0.f && (!rgb || !rgb_mask)) return ivid_set_error("ddim_step: replace_rgb tensors missing", hipSuccess);
  if (k.replace_depth_w >= 0.f && (!depth || !depth_mask)) return ivid_set_error("ddim_step: replace_depth tensors missing", hipSuccess);
  if (k.replace_depth_w >= 0.f && k.constrain_w >= 0.f && !convex) return ivid_set_error("ddim_step: convex missing", hipSuccess);
  if (k.sigma != 0.f && !noise) return ivid_set_error("ddim_step: noise missing for eta > 0", hipSuccess);
  StepPtrs q{x_t, eps_c, eps_u, rgb, rgb_mask, depth, depth_mask, convex, k.sigma != 0.f ? noise : nullptr, x_prev, x0};
  hipLaunchKernelGGL(ddim_step_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, q, k, HW);
  return ivid_check_launch("ddim_step");
}

extern "C" int ivid_ddpm_step(const float* x_t, const float* eps_c, const float* eps_u, const ivid_ddpm_coef* host_coef,
                              const float* noise, float* x_prev, float* x0, int B, int HW, void* stream) {
  ivid_ddpm_coef k = *host_coef;
  if (k.std != 0.f && !noise) return ivid_set_error("ddpm_step: noise missing", hipSuccess);
  StepPtrs q{x_t, eps_c, eps_u, nullptr, nullptr, nullptr, nullptr, nullptr, k.std != 0.f ? noise : nullptr, x_prev, x0};
  const long long total = (long long)B * 4 * HW;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, k,
                     total);
  return ivid_check_launch("ddpm_step");
}

// SuperResCFG.make_cond_inputs (sr_cfg.py:23-36): out = cat[x, bilinear_up(y, scale, align_corners=False)] along channels.
// torch's upsample_bilinear2d arithmetic: src = max((dst + 0.5) / scale - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1),
// lambda = src - i0, value = (1-ly) * ((1-lx) v00 + lx v01) + ly * ((1-lx) v10 + lx v11), all in fp32.
__global__ __launch_bounds__(256) void sr_cond_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ out, int Cx, int Cy, int S, int s) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (p >= S * S) return;
  const int r = p / S, c = p - r * S;
  const float rs = (float)s / (float)S;   // 1 / scale
  const float fy = fmaxf(((float)r + 0.5f) * rs - 0.5f, 0.f), fx = fmaxf(((float)c + 0.5f) * rs - 0.5f, 0.f);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, s - 1), x1 = min(x0 + 1, s - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const size_t HW = (size_t)S * S, hw = (size_t)s * s;
  float* o = out + (size_t)n * (Cx + Cy) * HW + p;
  for (int ch = 0; ch < Cx; ++ch) o[(size_t)ch * HW] = x[((size_t)n * Cx + ch) * HW + p];
  for (int ch = 0; ch < Cy; ++ch) {
    const float* q = y + ((size_t)n * Cy + ch) * hw;
    const float v = hy * (hx * q[(size_t)y0 * s + x0] + lx * q[(size_t)y0 * s + x1]) +
                    ly * (hx * q[(size_t)y1 * s + x0] + lx * q[(size_t)y1 * s + x1]);
    o[(size_t)(Cx + ch) * HW] = v;
  }
}

extern "C" int ivid_sr_cond(const float* x, const float* y, float* out, int B, int Cx, int Cy, int S, int s, void* stream) {
  if (B <= 0 || Cx <= 0 || Cy <= 0 || S <= 0 || s <= 0 || S % s) return ivid_set_error("sr_cond: S must be a multiple of s", hipSuccess);
  hipLaunchKernelGGL(sr_cond_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, y, out, Cx, Cy, S, s);
  return ivid_check_launch("sr_cond");
}

extern "C" int ivid_inpaint_cond(const float* x, const float* y, const float* mask, const float* mask_rgb,
                                 const float* noise_rgb, const float* noise_depth, float* out, int B, int HW,
                                 void* stream) {
  hipLaunchKernelGGL(inpaint_cond_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, y, mask,
                     mask_rgb, noise_rgb, noise_depth, out, HW);
  return ivid_check_launch("inpaint_cond");
}
