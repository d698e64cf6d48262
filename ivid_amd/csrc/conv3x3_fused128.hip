// The NARROW shape of the fused  GroupNorm-apply (+FiLM) + SiLU [+ nearest x2 upsample] + 3x3 convolution  kernel, for layers
// with Cout <= 128 (the first two levels of the small / SR models: 49 % + 19.5 % of their FLOPs): 16 x 32 output pixels x 128
// output channels, 64-byte channel chunks, 8 waves as 4 (m) x 2 (n), LDS 2 x 48,960 + 2 x 8,192 = 114,304 B.
//
// The kernel body is conv3x3_fused_body.h -- the same source as the wide kernel (FusedShape<false>): halo image with an odd
// 16-byte row stride, ping-pong wave groups, asm LDS-DMA weights, the 1x1 skip phase (plain or in split precision), wave-local
// epilogue, lo planes, the split-bf16 (bf16x3) instantiation.  This translation unit holds its instantiations and the
// dispatch conv3x3_fused.hip calls (ivid_conv3x3_gn_skip_s routes here when ivid_fused128_supports says so).
#include "conv3x3_fused_body.h"

// Can the narrow shape take the layer?  (Cout <= 128, 16 x 32 pixel tiles, 64-byte channel chunks)
bool ivid_fused128_supports(int dtype, int C0, int C1, int H, int W, int Cout, int skipC0, int skipC1) {
  typedef FusedShape<false> S;
  const int esz = ivid_esz(dtype);
  if (!esz) return false;
  const int bke = S::CHB / esz;
  return Cout <= S::BN && H % S::TH == 0 && W % S::TW == 0 && C0 > 0 && C0 % bke == 0 && C1 >= 0 && C1 % bke == 0 &&
         skipC0 >= 0 && skipC0 % bke == 0 && skipC1 >= 0 && skipC1 % bke == 0;
}

// Arguments already validated by ivid_conv3x3_gn_skip_s / ivid_conv3x3_gn_o16 (csrc/conv3x3_fused.hip), which dispatch here.
int ivid_fused128_launch(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                         const void* weight, const float* bias, void* out, const void* res, int res_mode, int N, int H, int W,
                         int Cout, float* stats, const void* skip0, int skipC0, const void* skip1, int skipC1,
                         const void* skip_weight, void* stream, void* out_lo, const void* res_lo, const void* src0_lo,
                         const void* src1_lo, const void* skip0_lo, const void* skip1_lo, const void* skip_weight_lo,
                         void* out16_hi, void* out16_lo) {
  FusedArgs a;
  a.src0 = (const char*)src0; a.src1 = (const char*)src1; a.ab = ab; a.w = (const char*)weight; a.bias = bias;
  a.out = (char*)out; a.res = (const char*)res; a.zero = (const char*)ivid_zero_page(); a.stats = stats;
  if (!a.zero) return -1;
  a.C0 = C0; a.C1 = C1; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.up = up ? 1 : 0; a.res_mode = res_mode;
  fused_geometry<false>(a);
  a.sk0 = (const char*)skip0; a.sk1 = (const char*)skip1; a.skw = (const char*)skip_weight; a.skC0 = skipC0; a.skC1 = skipC1;
  a.out_lo = (char*)out_lo; a.res_lo = (const char*)res_lo; a.src0_lo = (const char*)src0_lo; a.src1_lo = (const char*)src1_lo;
  a.sk0_lo = (const char*)skip0_lo; a.sk1_lo = (const char*)skip1_lo; a.skw_lo = (const char*)skip_weight_lo;
  a.out16_hi = (char*)out16_hi; a.out16_lo = (char*)out16_lo;
#ifdef IVID_DEV_TIMELINE
  a.dbg = nullptr;
#endif
  hipStream_t s = (hipStream_t)stream;
  if (out16_hi) return launch_fused<bf16x3_t, false, false, false, false, true>(a, s);
  if (skip_weight_lo) {
    if (src0_lo || src1_lo) return launch_fused<_Float16, false, true, true, true>(a, s);
    return launch_fused<_Float16, false, true, false, true>(a, s);
  }
  if (src0_lo || src1_lo) {
    if (dtype == IVID_BF16) return launch_fused<__bf16, false, true, true>(a, s);
    return launch_fused<_Float16, false, true, true>(a, s);
  }
  if (out_lo || res_lo) {
    if (dtype == IVID_BF16) return launch_fused<__bf16, false, true>(a, s);
    return launch_fused<_Float16, false, true>(a, s);
  }
  if (dtype == IVID_BF16) return launch_fused<__bf16, false>(a, s);
  if (dtype == IVID_F16) return launch_fused<_Float16, false>(a, s);
  if (dtype == IVID_BF16X3) return launch_fused<bf16x3_t, false>(a, s);
  return launch_fused<float, false>(a, s);
}
