// Fused  GroupNorm-apply (+FiLM) + SiLU [+ nearest x2 upsample] + 3x3 convolution  with an LDS-staged halo tile.
//
// Replaces, for the 3x3 convolutions of ResBlock2d at spatial widths >= 32 (87 % of the model's FLOPs):
//   in_layers  = GroupNorm32 -> SiLU -> [h_upd] -> Conv2d 3x3        (adm.py:157-161, 203-208)
//   out_layers = GroupNorm32*(1+scale)+shift -> SiLU -> Conv2d 3x3   (adm.py:177-183, 214-219)
// including the skip concat of the decoder (two source tensors, adm.py:563) and the residual epilogue.
//
// Compared with gn_apply + conv_igemm this removes the activated tensor's HBM round trip (one write + one read of
// every conv input) and loads each activation slab from L2 ONCE per channel chunk instead of once per tap:
//
//   tile      : 8 x 32 output pixels of ONE image (M = 256) x 256 output channels, 8 waves (2 x 4), wave = 4 image rows
//               x 64 channels (the 32-lane MFMA fragment is one 32-pixel image-row segment)
//   A operand : per 128-byte channel chunk, the (8+2) x (32+2) pixel HALO of the tile is read raw from HBM/L2 into
//               registers (one 16-byte piece = 8 bf16 channels of one pixel per lane), transformed ONCE
//               y = silu(x*a + b) with the per-(image, channel) GroupNorm/FiLM coefficients (gn_finalize's output),
//               zeroed outside the image (the conv pads the ACTIVATED tensor), and written to a double-buffered LDS
//               image with one 144-byte row per halo pixel (odd 16-byte stride: conflict-free for any row shift, so a
//               fragment address is one per-lane base + a compile-time offset).  The 9 taps read their fragments from
//               that image at shifted rows.
//   B operand : the [256 cout x 128 B] weight slab of (chunk, tap) streams through a 2-stage LDS ring with
//               global_load_lds (XOR-swizzled as in conv_igemm.hip).
//   schedule  : K-step = (chunk, tap), taps unrolled.  The two wave groups (waves w and w+4 share a SIMD) run ping-pong,
//               one barrier apart: while one executes its 32 MFMAs the other fetches its fragments, transforms one halo
//               piece of the NEXT chunk (requested two taps earlier, 2 x 4 VGPRs in flight) and -- the lagging group
//               only -- issues the whole weight DMA of the next step.  Details and the hazard analysis sit next to the
//               loop; measurements (ablations, phase cycle counters, power ceiling) in profiles/ and DESIGN.md.
//   optional  : the ResBlock's 1x1 skip_connection on the raw block input, accumulated into the same tile after the
//               3x3 K-steps (ivid_conv3x3_gn_skip).
#include "conv3x3_fused_body.h"



// with optional lo planes of the output and the residual source (compensated 16-bit storage, precision mode fp16c) and,
// optionally, the 1x1 skip phase in split precision (skip_weight_lo != NULL: lo planes of the skip sources + the lo part of
// the skip weights, precision mode fp16s)
static int fused_any(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                     int C1, const float* ab, int up,
                     const void* weight, const float* bias, void* out, void* out_lo, const void* res,
                     const void* res_lo, int res_mode, int N, int H, int W, int Cout, float* stats,
                     const void* skip0, int skipC0, const void* skip1, int skipC1, const void* skip_weight,
                     const void* skip0_lo, const void* skip1_lo, const void* skip_weight_lo, void* out16_hi, void* out16_lo,
                     void* stream);

extern "C" int ivid_conv3x3_gn_skip_s(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                                      int C1, const float* ab, int up,
                                      const void* weight, const float* bias, void* out, void* out_lo, const void* res,
                                      const void* res_lo, int res_mode, int N, int H, int W, int Cout, float* stats,
                                      const void* skip0, int skipC0, const void* skip1, int skipC1, const void* skip_weight,
                                      const void* skip0_lo, const void* skip1_lo, const void* skip_weight_lo, void* stream) {
  return fused_any(dtype, src0, src0_lo, C0, src1, src1_lo, C1, ab, up, weight, bias, out, out_lo, res, res_lo, res_mode, N, H, W, Cout,
                   stats, skip0, skipC0, skip1, skipC1, skip_weight, skip0_lo, skip1_lo, skip_weight_lo, nullptr, nullptr, stream);
}

// IVID_BF16X3 only (fp32 storage in, split-bf16 MFMA): the result leaves as two fp16 planes hi + lo (out16_hi / out16_lo: the
// compensated storage form of the 16-bit modes) in addition to (out != NULL) or instead of (out == NULL) the fp32 tensor --
// the hand-over from the split-precision island of the fp16s mode to the 16-bit part of the network without a conversion pass.
extern "C" int ivid_conv3x3_gn_o16(const void* src0, int C0, const void* src1, int C1, const float* ab, const void* weight,
                                   const float* bias, void* out, void* out16_hi, void* out16_lo, const void* res, int res_mode, int N,
                                   int H, int W, int Cout, float* stats, void* stream) {
  if (!out16_hi || !out16_lo) return ivid_set_error("conv3x3_gn_o16: both fp16 planes are required", hipSuccess);
  return fused_any(IVID_BF16X3, src0, nullptr, C0, src1, nullptr, C1, ab, 0, weight, bias, out, nullptr, res, nullptr, res_mode, N, H, W,
                   Cout, stats, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, out16_hi, out16_lo, stream);
}

static int fused_any(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                     int C1, const float* ab, int up,
                     const void* weight, const float* bias, void* out, void* out_lo, const void* res,
                     const void* res_lo, int res_mode, int N, int H, int W, int Cout, float* stats,
                     const void* skip0, int skipC0, const void* skip1, int skipC1, const void* skip_weight,
                     const void* skip0_lo, const void* skip1_lo, const void* skip_weight_lo, void* out16_hi, void* out16_lo,
                     void* stream) {
  const int esz = ivid_esz(dtype);
  if (!esz) return ivid_set_error("conv3x3_gn: bad dtype", hipSuccess);
  const int bke = 128 / esz, ve = 16 / esz;
  // Cout <= 128 (the small / SR models' first levels): the 16x32x128 variant with 64-byte chunks (csrc/conv3x3_fused128.hip)
  static const bool no128 = getenv("IVID_NO_FUSED128") && atoi(getenv("IVID_NO_FUSED128"));
  const bool narrow = !no128 && ivid_fused128_supports(dtype, C0, C1, H, W, Cout, skipC0, skipC1);
  if (!narrow && (C0 <= 0 || C0 % bke || C1 < 0 || C1 % bke))
    return ivid_set_error("conv3x3_gn: channels must be multiples of the K-step", hipSuccess);
  if (C1 > 0 && !src1) return ivid_set_error("conv3x3_gn: src1 missing", hipSuccess);
  if (W % FusedShape<true>::TW || H % FusedShape<true>::TH) return ivid_set_error("conv3x3_gn: needs W % 32 == 0 and H % 8 == 0 (use ivid_gn_apply + ivid_conv2d otherwise)", hipSuccess);
  if (Cout % ve) return ivid_set_error("conv3x3_gn: Cout must be a multiple of 16 bytes", hipSuccess);
  if (up && ((H | W) & 1)) return ivid_set_error("conv3x3_gn: upsample needs even H,W", hipSuccess);
  if (res_mode < 0 || res_mode > 3 || (res_mode && !res)) return ivid_set_error("conv3x3_gn: bad residual", hipSuccess);
  if (!ab) return ivid_set_error("conv3x3_gn: ab missing", hipSuccess);
  const int sbke = narrow ? bke / 2 : bke;   // the 128-wide variant works on 64-byte chunks
  if (skipC0 < 0 || skipC1 < 0 || skipC0 % sbke || skipC1 % sbke || (skipC0 == 0 && skipC1 > 0))
    return ivid_set_error("conv3x3_gn: skip channels must be multiples of the K-step", hipSuccess);
  if (skipC0 > 0 && (!skip0 || !skip_weight || (skipC1 > 0 && !skip1)))
    return ivid_set_error("conv3x3_gn: skip source / weight missing", hipSuccess);
  {  // the kernel addresses one image / the weight matrix with 32-bit byte offsets from a 64-bit wave-uniform base
    const size_t hs = up ? H / 2 : H, ws = up ? W / 2 : W, cmax = C0 > C1 ? C0 : C1, smax = skipC0 > skipC1 ? skipC0 : skipC1;
    if ((size_t)9 * (C0 + C1) * esz >= ((size_t)1 << 24) || hs * ws * cmax * esz >= ((size_t)1 << 31) || (size_t)Cout * 9 * (C0 + C1) * esz >= ((size_t)1 << 32) ||
        (size_t)H * W * smax * esz >= ((size_t)1 << 31) || (size_t)Cout * (skipC0 + skipC1) * esz >= ((size_t)1 << 32))
      return ivid_set_error("conv3x3_gn: image or weight matrix too large for 32-bit offsets", hipSuccess);
  }
  const bool any_lo = out_lo || res_lo || src0_lo || src1_lo;
  if (any_lo && esz != 2) return ivid_set_error("conv3x3_gn: lo planes need a 16-bit dtype", hipSuccess);
  if (src1_lo && C1 <= 0) return ivid_set_error("conv3x3_gn: src1_lo without src1", hipSuccess);
  if (res_lo && !res_mode) return ivid_set_error("conv3x3_gn: res_lo without a residual", hipSuccess);
  if (skip_weight_lo) {
    if (esz != 2 || dtype != IVID_F16) return ivid_set_error("conv3x3_gn: the split skip phase is an fp16 instantiation", hipSuccess);
    if (skipC0 <= 0 || !skip0_lo || (skipC1 > 0 && !skip1_lo))
      return ivid_set_error("conv3x3_gn: the split skip phase needs a lo plane of every skip source", hipSuccess);
  }
  if (out16_hi && dtype != IVID_BF16X3) return ivid_set_error("conv3x3_gn_o16: IVID_BF16X3 only", hipSuccess);
  if (narrow)
    return ivid_fused128_launch(dtype, src0, C0, src1, C1, ab, up, weight, bias, out, res, res_mode, N, H, W, Cout, stats, skip0,
                                skipC0, skip1, skipC1, skip_weight, stream, out_lo, res_lo, src0_lo, src1_lo, skip0_lo, skip1_lo,
                                skip_weight_lo, out16_hi, out16_lo);
  FusedArgs a;
  a.src0 = (const char*)src0; a.src1 = (const char*)src1; a.ab = ab; a.w = (const char*)weight; a.bias = bias;
  a.out = (char*)out; a.res = (const char*)res; a.zero = (const char*)ivid_zero_page(); a.stats = stats;
  if (!a.zero) return -1;
  a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.up = up ? 1 : 0; a.res_mode = res_mode;
  fused_geometry<true>(a);
  a.sk0 = (const char*)skip0; a.sk1 = (const char*)skip1; a.skw = (const char*)skip_weight; a.skC0 = skipC0; a.skC1 = skipC1;
  a.out_lo = (char*)out_lo; a.res_lo = (const char*)res_lo; a.src0_lo = (const char*)src0_lo; a.src1_lo = (const char*)src1_lo;
  a.sk0_lo = (const char*)skip0_lo; a.sk1_lo = (const char*)skip1_lo; a.skw_lo = (const char*)skip_weight_lo;
  a.out16_hi = (char*)out16_hi; a.out16_lo = (char*)out16_lo;
#ifdef IVID_DEV_TIMELINE
  a.dbg = g_timeline;
#endif
  if (out16_hi) {
    return launch_fused<bf16x3_t, true, false, false, false, true>(a, (hipStream_t)stream);
  }
  if (skip_weight_lo) {
    if (src0_lo || src1_lo) return launch_fused<_Float16, true, true, true, true>(a, (hipStream_t)stream);
    return launch_fused<_Float16, true, true, false, true>(a, (hipStream_t)stream);
  }
  if (src0_lo || src1_lo) {
    if (dtype == IVID_BF16) return launch_fused<__bf16, true, true, true>(a, (hipStream_t)stream);
    return launch_fused<_Float16, true, true, true>(a, (hipStream_t)stream);
  }
  if (any_lo) {
    if (dtype == IVID_BF16) return launch_fused<__bf16, true, true>(a, (hipStream_t)stream);
    return launch_fused<_Float16, true, true>(a, (hipStream_t)stream);
  }
  if (dtype == IVID_BF16) return launch_fused<__bf16, true>(a, (hipStream_t)stream);
  if (dtype == IVID_F16) return launch_fused<_Float16, true>(a, (hipStream_t)stream);
  if (dtype == IVID_BF16X3) return launch_fused<bf16x3_t, true>(a, (hipStream_t)stream);
  return launch_fused<float, true>(a, (hipStream_t)stream);
}

extern "C" int ivid_conv3x3_gn_skip_c(int dtype, const void* src0, const void* src0_lo, int C0, const void* src1, const void* src1_lo,
                                      int C1, const float* ab, int up,
                                      const void* weight, const float* bias, void* out, void* out_lo, const void* res,
                                      const void* res_lo, int res_mode, int N, int H, int W, int Cout, float* stats,
                                      const void* skip0, int skipC0, const void* skip1, int skipC1, const void* skip_weight,
                                      void* stream) {
  return ivid_conv3x3_gn_skip_s(dtype, src0, src0_lo, C0, src1, src1_lo, C1, ab, up, weight, bias, out, out_lo, res, res_lo, res_mode,
                                N, H, W, Cout, stats, skip0, skipC0, skip1, skipC1, skip_weight, nullptr, nullptr, nullptr, stream);
}

extern "C" int ivid_conv3x3_gn_skip(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                                    const void* weight, const float* bias, void* out, const void* res, int res_mode, int N,
                                    int H, int W, int Cout, float* stats, const void* skip0, int skipC0, const void* skip1,
                                    int skipC1, const void* skip_weight, void* stream) {
  return ivid_conv3x3_gn_skip_c(dtype, src0, nullptr, C0, src1, nullptr, C1, ab, up, weight, bias, out, nullptr, res, nullptr, res_mode,
                                N, H, W, Cout, stats, skip0, skipC0, skip1, skipC1, skip_weight, stream);
}

#ifdef IVID_DEV_TIMELINE
extern "C" void ivid_dev_timeline(void* buf) { g_timeline = (unsigned long long*)buf; }
#endif

extern "C" int ivid_conv3x3_gn(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                               const void* weight, const float* bias, void* out, const void* res, int res_mode, int N,
                               int H, int W, int Cout, float* stats, void* stream) {
  return ivid_conv3x3_gn_skip(dtype, src0, C0, src1, C1, ab, up, weight, bias, out, res, res_mode, N, H, W, Cout, stats,
                              nullptr, 0, nullptr, 0, nullptr, stream);
}
